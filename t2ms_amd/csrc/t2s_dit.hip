// DiT denoiser forward for MI355X: handle, weight packing, small VALU kernels and the
// per-forward launch sequence.  Mirrors Transformer.forward
// (reference model/denoiser/transformer.py:158-193); heavy lifting is in t2s_gemm.h
// (fused linears) and t2s_attn.hip (fused attention).
#include <mutex>
#include <string>
#include <vector>

#include "t2s_gemm.h"

namespace t2s {

// ------------------------------------------------------------------ error string
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int launch_attn(const float* q, const float* k, const float* v, float* o, int BH, int o_row_stride,
                int o_head_stride, int o_seq_stride, hipStream_t st);
int attn_init();

// ------------------------------------------------------------------ small kernels
__global__ void pack_weight_kernel(const float* __restrict__ W, float* __restrict__ P, int N, int K,
                                   int n_offset, int K_total_rows) {
    // W is (N,K); packed destination covers rows [n_offset, n_offset+N) of a (K_total_rows,K) matrix
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * K) return;
    const int n = idx / K, k = idx - n * K;
    P[packed_index(n + n_offset, k, K)] = W[idx];
    (void)K_total_rows;
}

// TimeEmbedding.forward (transformer.py:30-40): out[b] = [sin(100 t / f) | cos(100 t / f)]
__global__ void time_embedding_kernel(const float* __restrict__ t, const float* __restrict__ freqs,
                                      float* __restrict__ out, int B) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * 64) return;
    const int b = idx >> 6, i = idx & 63;
    const float arg = (t[b] * 100.0f) / freqs[i];
    out[b * D + i] = sinf(arg);
    out[b * D + 64 + i] = cosf(arg);
}

// c = t_emb (+ text) (transformer.py:176-178).  Rows [0,uncond_rows) get no text.
// With step_ptr != NULL, temb is a (steps,128) table and row *step_ptr is used for every sequence
// (the sampling loop: t is shared by the batch, infer.py:78,84).
__global__ void cond_kernel(float* __restrict__ c, const float* __restrict__ temb, int temb_rows,
                            const int* __restrict__ step_ptr, const float* __restrict__ text,
                            int uncond_rows, int S) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * D) return;
    const int s = idx >> 7, d = idx & 127;
    const int trow = step_ptr ? *step_ptr : (temb_rows == 1 ? 0 : s);
    float val = temb[(size_t)trow * D + d];
    if (s >= uncond_rows) val += text[(size_t)(s - uncond_rows) * D + d];
    c[idx] = val;
}

// patchify (transformer.py:166-172): token n = hh*32 + ww reads the 2x2 patch
// in[b][2ww+j][2hh+i]; conv 1->4 (2x2, stride 2), Linear 4->128, + pos_embed.
// Sequence s reads latent row s % B (both CFG branches share x).
__global__ __launch_bounds__(256) void patchify_kernel(
    const float* __restrict__ x, int B, float* __restrict__ h, int S, const float* __restrict__ cw,
    const float* __restrict__ cb, const float* __restrict__ pw, const float* __restrict__ pb,
    const float* __restrict__ pos) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (token, float4 column)
    const int c4 = gid & 31;
    const int tokg = gid >> 5;
    if (tokg >= S * NTOK) return;
    const int s = tokg / NTOK, n = tokg - s * NTOK;
    const int hh = n >> 5, ww = n & 31;
    const float* xin = x + (size_t)(s % B) * LAT;
    float px[2][2];  // [i][j]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) px[i][j] = xin[(2 * ww + j) * LATW + 2 * hh + i];
    float cv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float acc = cw[c * 4 + 0] * px[0][0];
        acc += cw[c * 4 + 1] * px[0][1];
        acc += cw[c * 4 + 2] * px[1][0];
        acc += cw[c * 4 + 3] * px[1][1];
        cv[c] = acc + cb[c];
    }
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int d = c4 * 4 + e;
        const f32x4 w = *reinterpret_cast<const f32x4*>(pw + d * 4);
        float acc = w.x * cv[0];
        acc += w.y * cv[1];
        acc += w.z * cv[2];
        acc += w.w * cv[3];
        o[e] = acc + pb[d] + pos[n * D + d];
    }
    *reinterpret_cast<f32x4*>(h + (size_t)tokg * D + c4 * 4) = o;
}

// final layer (transformer.py:182-191): affine LayerNorm (eps 1e-5), Linear 128->4, unpatchify:
// out[s][(2ww+pw)*30 + 2hh+ph] = y[ph*2+pw].  Sequences [0,split) go to out0, the rest to out1.
__global__ __launch_bounds__(256) void final_kernel(const float* __restrict__ h, int S,
                                                    const float* __restrict__ lnw,
                                                    const float* __restrict__ lnb,
                                                    const float* __restrict__ ow,
                                                    const float* __restrict__ ob,
                                                    float* __restrict__ out0,
                                                    float* __restrict__ out1, int split) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = gid & 31;
    int tokg = gid >> 5;
    const bool valid = tokg < S * NTOK;
    if (!valid) tokg = S * NTOK - 1;
    const f32x4 v = *reinterpret_cast<const f32x4*>(h + (size_t)tokg * D + c4 * 4);
    float s1 = (v.x + v.y) + (v.z + v.w);
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) s1 += __shfl_xor(s1, o, 64);
    const float mean = s1 * (1.0f / 128.0f);
    const f32x4 d = v - mean;
    float s2 = (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o, 64);
    const float rstd = rsqrtf(s2 * (1.0f / 128.0f) + 1e-5f);
    const f32x4 g = *reinterpret_cast<const f32x4*>(lnw + c4 * 4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(lnb + c4 * 4);
    const f32x4 y = (d * rstd) * g + b;
    float acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(ow + p * D + c4 * 4);
        float a = (y.x * w.x + y.y * w.y) + (y.z * w.z + y.w * w.w);
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) a += __shfl_xor(a, o, 64);
        acc[p] = a + ob[p];
    }
    if (valid && c4 < 4) {
        const int s = tokg / NTOK, n = tokg - s * NTOK;
        const int hh = n >> 5, ww = n & 31;
        const int ph = c4 >> 1, pw = c4 & 1;
        float* dst = (s < split) ? out0 + (size_t)s * LAT : out1 + (size_t)(s - split) * LAT;
        dst[(2 * ww + pw) * LATW + 2 * hh + ph] = acc[c4];
    }
}

}  // namespace t2s

using namespace t2s;

// ------------------------------------------------------------------ handle
struct t2s_dit {
    int max_seqs = 0;
    // parameters (device)
    float* arena = nullptr;  // all small fp32 params, offsets below
    float *conv_w, *conv_b, *patch_w, *patch_b, *pos, *ln_w, *ln_b, *out_w, *out_b, *freqs;
    float *qkv_b[NBLK], *proj_b[NBLK], *fc1_b[NBLK], *fc2_b[NBLK], *ada_b;
    f32x4 *qkv_p[NBLK], *proj_p[NBLK], *fc1_p[NBLK], *fc2_p[NBLK], *ada_p;
    // workspace (device)
    float *h = nullptr, *q = nullptr, *k = nullptr, *v = nullptr, *ao = nullptr, *mid = nullptr;
    float *mod = nullptr, *c = nullptr;
};

namespace {

struct ArenaPlan {
    size_t off = 0;
    size_t take(size_t n) {
        size_t o = off;
        off += (n + 63) & ~size_t(63);
        return o;
    }
};

int copy_param(float* dst, const float* src, size_t n, hipStream_t st) {
    T2S_HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    return T2S_OK;
}

int pack(const float* W, f32x4* P, int N, int K, int n_offset, hipStream_t st) {
    const int total = N * K;
    pack_weight_kernel<<<(total + 255) / 256, 256, 0, st>>>(W, reinterpret_cast<float*>(P), N, K,
                                                            n_offset, 0);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

int upload_weights(t2s_dit* h, const t2s_dit_weights* w, hipStream_t st) {
    T2S_REQUIRE(w->conv_w && w->conv_b && w->patch_w && w->patch_b && w->pos_embed && w->ln_w &&
                    w->ln_b && w->out_w && w->out_b && w->time_freqs,
                "t2s_dit weights: NULL top-level pointer");
    int rc;
#define CP(dst, src, n) \
    if ((rc = copy_param(dst, src, n, st)) != T2S_OK) return rc
    CP(h->conv_w, w->conv_w, 16);
    CP(h->conv_b, w->conv_b, 4);
    CP(h->patch_w, w->patch_w, 128 * 4);
    CP(h->patch_b, w->patch_b, 128);
    CP(h->pos, w->pos_embed, NTOK * D);
    CP(h->ln_w, w->ln_w, D);
    CP(h->ln_b, w->ln_b, D);
    CP(h->out_w, w->out_w, 4 * D);
    CP(h->out_b, w->out_b, 4);
    CP(h->freqs, w->time_freqs, 64);
    for (int i = 0; i < NBLK; ++i) {
        const t2s_dit_block_weights& b = w->blk[i];
        T2S_REQUIRE(b.qkv_w && b.qkv_b && b.proj_w && b.proj_b && b.fc1_w && b.fc1_b && b.fc2_w &&
                        b.fc2_b && b.ada_w && b.ada_b,
                    "t2s_dit weights: NULL pointer in block %d", i);
        CP(h->qkv_b[i], b.qkv_b, 3 * D);
        CP(h->proj_b[i], b.proj_b, D);
        CP(h->fc1_b[i], b.fc1_b, 2 * D);
        CP(h->fc2_b[i], b.fc2_b, D);
        CP(h->ada_b + i * MODW, b.ada_b, MODW);
        if ((rc = pack(b.qkv_w, h->qkv_p[i], 3 * D, D, 0, st)) != T2S_OK) return rc;
        if ((rc = pack(b.proj_w, h->proj_p[i], D, D, 0, st)) != T2S_OK) return rc;
        if ((rc = pack(b.fc1_w, h->fc1_p[i], 2 * D, D, 0, st)) != T2S_OK) return rc;
        if ((rc = pack(b.fc2_w, h->fc2_p[i], D, 2 * D, 0, st)) != T2S_OK) return rc;
        if ((rc = pack(b.ada_w, h->ada_p, MODW, D, i * MODW, st)) != T2S_OK) return rc;
    }
#undef CP
    return T2S_OK;
}

// One DiT forward over S sequences (sequence s reads latent row s % B).
int run_forward(t2s_dit* h, const float* x, int B, int S, int uncond_rows, const float* temb,
                int temb_rows, const int* step_ptr, const float* text, float* out0, float* out1,
                int split, hipStream_t st) {
    int rc;
    cond_kernel<<<(S * D + 255) / 256, 256, 0, st>>>(h->c, temb, temb_rows, step_ptr, text,
                                                     uncond_rows, S);
    T2S_LAUNCH_CHECK();
    {   // adaLN for all 4 blocks at once: mod = silu(c) @ W_ada^T + b   (transformer.py:106-109,115)
        GemmArgs a{};
        a.A = h->c; a.Wp = h->ada_p; a.bias = h->ada_b; a.out = h->mod; a.M = S; a.N = MODROW;
        if ((rc = launch_gemm_rows<128, 3, PRO_SILU, EPI_BIAS>(a, st)) != T2S_OK) return rc;
    }
    {
        const int threads = S * NTOK * 32;
        patchify_kernel<<<(threads + 255) / 256, 256, 0, st>>>(x, B, h->h, S, h->conv_w, h->conv_b,
                                                               h->patch_w, h->patch_b, h->pos);
        T2S_LAUNCH_CHECK();
    }
    const int M = S * NTOK;
    for (int i = 0; i < NBLK; ++i) {
        const int base = i * MODW;
        {   // LN1 + modulate + qkv linear, scattered to (S,4,480,32) q/k/v
            GemmArgs a{};
            a.A = h->h; a.Wp = h->qkv_p[i]; a.bias = h->qkv_b[i]; a.M = M; a.N = 3 * D;
            a.mod = h->mod; a.shift_off = base + 0 * D; a.scale_off = base + 1 * D;
            a.q = h->q; a.k = h->k; a.v = h->v;
            if ((rc = launch_gemm_rows<128, 3, PRO_LNMOD, EPI_QKV>(a, st)) != T2S_OK) return rc;
        }
        if ((rc = launch_attn(h->q, h->k, h->v, h->ao, S * NH, D, DH, NTOK * D, st)) != T2S_OK)
            return rc;
        {   // x += gate_msa * proj(attn)
            GemmArgs a{};
            a.A = h->ao; a.Wp = h->proj_p[i]; a.bias = h->proj_b[i]; a.out = h->h; a.M = M; a.N = D;
            a.mod = h->mod; a.gate_off = base + 2 * D;
            if ((rc = launch_gemm_rows<128, 1, PRO_PLAIN, EPI_GATERES>(a, st)) != T2S_OK) return rc;
        }
        {   // LN2 + modulate + fc1 + GELU(tanh)
            GemmArgs a{};
            a.A = h->h; a.Wp = h->fc1_p[i]; a.bias = h->fc1_b[i]; a.out = h->mid; a.M = M; a.N = 2 * D;
            a.mod = h->mod; a.shift_off = base + 3 * D; a.scale_off = base + 4 * D;
            if ((rc = launch_gemm_rows<128, 2, PRO_LNMOD, EPI_GELU>(a, st)) != T2S_OK) return rc;
        }
        {   // x += gate_mlp * fc2(.)
            GemmArgs a{};
            a.A = h->mid; a.Wp = h->fc2_p[i]; a.bias = h->fc2_b[i]; a.out = h->h; a.M = M; a.N = D;
            a.mod = h->mod; a.gate_off = base + 5 * D;
            if ((rc = launch_gemm_rows<256, 1, PRO_PLAIN, EPI_GATERES>(a, st)) != T2S_OK) return rc;
        }
    }
    {
        const int threads = S * NTOK * 32;
        final_kernel<<<(threads + 255) / 256, 256, 0, st>>>(h->h, S, h->ln_w, h->ln_b, h->out_w,
                                                            h->out_b, out0, out1, split);
        T2S_LAUNCH_CHECK();
    }
    return T2S_OK;
}

}  // namespace

// exported to the sampler TU
namespace t2s {
int dit_forward_cfg_step(t2s_dit* h, const float* x, const float* temb_table, const int* step_ptr,
                         const float* text, float* out_u, float* out_c, int B, hipStream_t st) {
    return run_forward(h, x, B, 2 * B, B, temb_table, 1, step_ptr, text, out_u, out_c, B, st);
}
}  // namespace t2s

// ------------------------------------------------------------------ C ABI
extern "C" {

const char* t2s_last_error(void) { return t2s::g_err; }
const char* t2s_version(void) { return "t2s 0.1 gfx950 fp32-mfma"; }

int t2s_dit_create(const t2s_dit_weights* w, int max_seqs, t2s_dit** out) {
    T2S_REQUIRE(w && out, "t2s_dit_create: NULL argument");
    T2S_REQUIRE(max_seqs > 0 && max_seqs <= 65536, "t2s_dit_create: max_seqs=%d out of range", max_seqs);
    t2s_dit* h = new t2s_dit();
    h->max_seqs = max_seqs;
    ArenaPlan p;
    const size_t o_conv_w = p.take(16), o_conv_b = p.take(4), o_patch_w = p.take(512),
                 o_patch_b = p.take(128), o_pos = p.take(NTOK * D), o_ln_w = p.take(D),
                 o_ln_b = p.take(D), o_out_w = p.take(4 * D), o_out_b = p.take(4),
                 o_freqs = p.take(64), o_ada_b = p.take(MODROW), o_ada_p = p.take((size_t)MODROW * D);
    size_t o_qkv_b[NBLK], o_proj_b[NBLK], o_fc1_b[NBLK], o_fc2_b[NBLK];
    size_t o_qkv_p[NBLK], o_proj_p[NBLK], o_fc1_p[NBLK], o_fc2_p[NBLK];
    for (int i = 0; i < NBLK; ++i) {
        o_qkv_b[i] = p.take(3 * D); o_proj_b[i] = p.take(D); o_fc1_b[i] = p.take(2 * D);
        o_fc2_b[i] = p.take(D);
        o_qkv_p[i] = p.take(3 * D * D); o_proj_p[i] = p.take(D * D);
        o_fc1_p[i] = p.take(2 * D * D); o_fc2_p[i] = p.take(2 * D * D);
    }
    hipError_t e = hipMalloc(&h->arena, p.off * sizeof(float));
    if (e != hipSuccess) {
        set_error("t2s_dit_create: hipMalloc(params) failed: %s", hipGetErrorString(e));
        delete h;
        return T2S_E_HIP;
    }
    float* A = h->arena;
    h->conv_w = A + o_conv_w; h->conv_b = A + o_conv_b; h->patch_w = A + o_patch_w;
    h->patch_b = A + o_patch_b; h->pos = A + o_pos; h->ln_w = A + o_ln_w; h->ln_b = A + o_ln_b;
    h->out_w = A + o_out_w; h->out_b = A + o_out_b; h->freqs = A + o_freqs; h->ada_b = A + o_ada_b;
    h->ada_p = reinterpret_cast<f32x4*>(A + o_ada_p);
    for (int i = 0; i < NBLK; ++i) {
        h->qkv_b[i] = A + o_qkv_b[i]; h->proj_b[i] = A + o_proj_b[i];
        h->fc1_b[i] = A + o_fc1_b[i]; h->fc2_b[i] = A + o_fc2_b[i];
        h->qkv_p[i] = reinterpret_cast<f32x4*>(A + o_qkv_p[i]);
        h->proj_p[i] = reinterpret_cast<f32x4*>(A + o_proj_p[i]);
        h->fc1_p[i] = reinterpret_cast<f32x4*>(A + o_fc1_p[i]);
        h->fc2_p[i] = reinterpret_cast<f32x4*>(A + o_fc2_p[i]);
    }
    const size_t S = (size_t)max_seqs, tokD = S * NTOK * D;
    float** bufs[] = {&h->h, &h->q, &h->k, &h->v, &h->ao, &h->mid, &h->mod, &h->c};
    const size_t sizes[] = {tokD, tokD, tokD, tokD, tokD, 2 * tokD, S * MODROW, S * D};
    for (int i = 0; i < 8; ++i) {
        e = hipMalloc(bufs[i], sizes[i] * sizeof(float));
        if (e != hipSuccess) {
            set_error("t2s_dit_create: hipMalloc(workspace %d, %zu B) failed: %s", i,
                      sizes[i] * sizeof(float), hipGetErrorString(e));
            t2s_dit_destroy(h);
            return T2S_E_HIP;
        }
    }
    int rc = attn_init();
    if (rc == T2S_OK) rc = gemm_rows_init<256, 1, PRO_PLAIN, EPI_GATERES>();
    if (rc == T2S_OK) rc = upload_weights(h, w, nullptr);
    if (rc == T2S_OK && hipStreamSynchronize(nullptr) != hipSuccess) {
        set_error("t2s_dit_create: weight upload failed");
        rc = T2S_E_HIP;
    }
    if (rc != T2S_OK) {
        t2s_dit_destroy(h);
        return rc;
    }
    *out = h;
    return T2S_OK;
}

int t2s_dit_update_weights(t2s_dit* h, const t2s_dit_weights* w, void* stream) {
    T2S_REQUIRE(h && w, "t2s_dit_update_weights: NULL argument");
    return upload_weights(h, w, (hipStream_t)stream);
}

void t2s_dit_destroy(t2s_dit* h) {
    if (!h) return;
    float* bufs[] = {h->arena, h->h, h->q, h->k, h->v, h->ao, h->mid, h->mod, h->c};
    for (float* b : bufs)
        if (b) (void)hipFree(b);
    delete h;
}

int t2s_dit_max_seqs(const t2s_dit* h) { return h ? h->max_seqs : 0; }

int t2s_time_embedding(const t2s_dit* h, const float* t, float* out, int B, void* stream) {
    T2S_REQUIRE(h && t && out && B > 0, "t2s_time_embedding: bad argument");
    time_embedding_kernel<<<(B * 64 + 255) / 256, 256, 0, (hipStream_t)stream>>>(t, h->freqs, out, B);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

int t2s_dit_forward(t2s_dit* h, const float* x, const float* temb, int temb_rows, const float* text,
                    float* out, int B, void* stream) {
    T2S_REQUIRE(h && x && temb && out, "t2s_dit_forward: NULL argument");
    T2S_REQUIRE(B > 0 && B <= h->max_seqs, "t2s_dit_forward: B=%d exceeds max_seqs=%d", B, h->max_seqs);
    T2S_REQUIRE(temb_rows == 1 || temb_rows == B, "t2s_dit_forward: temb_rows=%d must be 1 or B=%d",
                temb_rows, B);
    return run_forward(h, x, B, B, text ? 0 : B, temb, temb_rows, nullptr, text, out, out, B,
                       (hipStream_t)stream);
}

int t2s_dit_forward_cfg(t2s_dit* h, const float* x, const float* temb, const float* text,
                        float* out_uncond, float* out_cond, int B, void* stream) {
    T2S_REQUIRE(h && x && temb && text && out_uncond && out_cond, "t2s_dit_forward_cfg: NULL argument");
    T2S_REQUIRE(B > 0 && 2 * B <= h->max_seqs, "t2s_dit_forward_cfg: 2*B=%d exceeds max_seqs=%d", 2 * B,
                h->max_seqs);
    return t2s::dit_forward_cfg_step(h, x, temb, nullptr, text, out_uncond, out_cond, B,
                                     (hipStream_t)stream);
}

int t2s_dit_read_stream(const t2s_dit* h, float* out, int S, void* stream) {
    T2S_REQUIRE(h && out && S > 0 && S <= h->max_seqs, "t2s_dit_read_stream: bad argument");
    T2S_HIP_CHECK(hipMemcpyAsync(out, h->h, (size_t)S * NTOK * D * sizeof(float),
                                 hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return T2S_OK;
}

}  // extern "C"
