// DiT denoiser forward for MI355X: handle, weight packing, small VALU kernels and the
// per-forward launch sequence.  Mirrors Transformer.forward
// (reference model/denoiser/transformer.py:158-193).  The heavy lifting is in
//   t2s_rows.h  : the register-resident row-local chain (proj, MLP, next block's qkv)
//   t2s_attn.hip: fused attention
//   t2s_gemm.h  : the adaLN modulation GEMM (SiLU prologue)
// Between kernels the residual stream, the attention output and q/k/v stay in the
// fragment-major layout (t2s_common.h: frag_index).
#include <stdlib.h>
#include <vector>

#include "t2s_gemm.h"
#include "t2s_rows_x3.h"
#include "t2s_rows16.h"

namespace t2s {

// ------------------------------------------------------------------ error string
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int launch_attn_packed(const float* q, const float* k, const float* vT, float* o, int BH, hipStream_t st);
int attn_init();
int launch_attn_x3(const float* q, const __bf16* k3, const __bf16* vT3, float* o, int BH, hipStream_t st);
int attn_x3_init();
void train_free(t2s_dit* h);

// ------------------------------------------------------------------ small kernels
// W (N,K) row-major -> MFMA-fragment order.  mode 0: packed_index (tile-major: [nt][G][lane][e]);
// mode 1 (fc2, K=256): chunk order [c = G/4][nt][g = G%4][lane][e] so the 16 fragments one fc1
// chunk feeds into fc2 are one contiguous 16 KiB (t2s_rows.h); modes 2 / 3: the same two orders for the
// 16-token kernel's 16x16x4 fragments (t2s_rows16.h: packed16_index, packed16_fc2_index).
__device__ __forceinline__ void pack_weight_elem(const float* __restrict__ W, float* __restrict__ P, int N, int K,
                                                 int n_offset, int mode, int idx) {
    if (idx >= N * K) return;
    const int n = idx / K + n_offset, k = idx - (idx / K) * K;
    if (mode == 0) {
        P[packed_index(n, k, K)] = W[idx];
    } else if (mode == 2) {
        P[packed16_index(n, k)] = W[idx];
    } else if (mode == 3) {
        P[packed16_fc2_index(n, k)] = W[idx];
    } else {
        const int nt = n >> 5, j = n & 31, G = k >> 3, h = (k >> 2) & 1, e = k & 3;
        const int c = G >> 2, g = G & 3;
        P[((((size_t)(c * (N >> 5) + nt) * 4 + g) * 64) + (h * 32 + j)) * 4 + e] = W[idx];
    }
}

__global__ void pack_weight_kernel(const float* __restrict__ W, float* __restrict__ P, int N, int K,
                                   int n_offset, int mode) {
    pack_weight_elem(W, P, N, K, n_offset, mode, blockIdx.x * blockDim.x + threadIdx.x);
}

// Every parameter refresh of a handle in TWO launches instead of 50 (30 device-to-device copies + 20 packs, 3-5 us
// each: 0.25 ms of a 14.5 ms bf16 training step): the job tables travel as kernel arguments, blockIdx.y = job.
constexpr int MULTI_JOBS = 48;
struct PackJob {
    const float* W;
    float* P;
    int N, K, n_offset, mode;
};
struct PackTable {
    PackJob j[MULTI_JOBS];
};
__global__ void pack_weight_multi_kernel(const PackTable t) {
    const PackJob& j = t.j[blockIdx.y];
    pack_weight_elem(j.W, j.P, j.N, j.K, j.n_offset, j.mode, blockIdx.x * blockDim.x + threadIdx.x);
}
struct CopyJob {
    const float* src;
    float* dst;
    int n;
};
struct CopyTable {
    CopyJob j[MULTI_JOBS];
};
__global__ void copy_multi_kernel(const CopyTable t) {
    const CopyJob& j = t.j[blockIdx.y];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < j.n; i += gridDim.x * blockDim.x) j.dst[i] = j.src[i];
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

// adaLN modulation of all 4 blocks (transformer.py:106-109,115,176-178) in ONE launch:
//   mod[s][:] = silu(c[s]) @ W_ada^T + b,   c[s] = t_emb (+ text[s - uncond_rows] for s >= uncond_rows).
// With step_ptr != NULL, temb is a (steps,128) table and row *step_ptr is used for every sequence (the sampling
// loop: t is shared by the batch, infer.py:78,84).
// This is a latency problem, not a throughput one (S x 3072 x 128: 0.05-0.4 GFLOP): one wave per (32 sequences,
// 32 outputs) tile = 64 MFMAs, the A operand (silu(c)) built straight in registers, the packed weights read from
// L2 -- 384 waves at S = 64, 1536 at S = 512.  (The generic gemm_rows_kernel ran it as 8 workgroups of 384 MFMAs per
// wave at S = 64: 29 us of a 690 us sampling step.)
// table_rows > 0 (the sampler's whole-run table): row = j * table_rows + r is step j; r = 0 is the text-free branch, r > 0
// adds text row r - 1 -- S = steps * table_rows rows, step_ptr unused.
__global__ __launch_bounds__(256) void adaln_kernel(float* __restrict__ mod, const float* __restrict__ temb,
                                                    int temb_rows, const int* __restrict__ step_ptr,
                                                    const float* __restrict__ text, int uncond_rows, int S,
                                                    const f32x4* __restrict__ Wp, const float* __restrict__ bias,
                                                    int table_rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int nt = blockIdx.y * 4 + wave;                 // 32-output tile, 0..95
    const int row = blockIdx.x * 32 + j;
    const int rl = row < S ? row : S - 1;                 // clamped for the loads
    int trow, xrow;                                       // time-embedding row, text row (-1: none)
    if (table_rows > 0) {
        trow = rl / table_rows;
        xrow = rl - trow * table_rows - 1;
    } else {
        trow = step_ptr ? *step_ptr : (temb_rows == 1 ? 0 : rl % temb_rows);   // a CFG pass (S = 2 temb_rows): both branches read row b
        xrow = rl >= uncond_rows ? rl - uncond_rows : -1;
    }
    const f32x4* tp = reinterpret_cast<const f32x4*>(temb + (size_t)trow * D) + half;
    const f32x4* xp = xrow >= 0 ? reinterpret_cast<const f32x4*>(text + (size_t)xrow * D) + half : nullptr;
    const f32x4* wp = Wp + (size_t)nt * 16 * 64 + lane;
    f32x4 a[16], b[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) b[g] = wp[g * 64];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        f32x4 v = tp[2 * g];
        if (xp) v += xp[2 * g];
        v.x = v.x / (1.0f + __expf(-v.x));
        v.y = v.y / (1.0f + __expf(-v.y));
        v.z = v.z / (1.0f + __expf(-v.z));
        v.w = v.w / (1.0f + __expf(-v.w));
        a[g] = v;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mfma32(a[g][e], b[g][e], acc);
    const int col = nt * 32 + j;
    const float bv = bias[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int grow = blockIdx.x * 32 + acc_row(r, half);
        if (grow < S) mod[(size_t)grow * MODROW + col] = acc[r] + bv;
    }
}

// patchify (transformer.py:166-172): token n = hh*32 + ww reads the 2x2 patch
// in[b][2ww+j][2hh+i]; conv 1->4 (2x2, stride 2), Linear 4->128, + pos_embed.
// Sequence s reads latent row s % B (both CFG branches share x).  Output fragment-major.
__global__ __launch_bounds__(256) void patchify_kernel(
    const float* __restrict__ x, int B, float* __restrict__ h, int S, const float* __restrict__ cw,
    const float* __restrict__ cb, const float* __restrict__ pw, const float* __restrict__ pb,
    const float* __restrict__ pos) {
    // one thread per output float4; consecutive threads walk the fragment-major order
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= S * NTOK * 32) return;
    const int l = gid & 63, G = (gid >> 6) & 15, tile = gid >> 10;
    const int tokg = tile * 32 + (l & 31);
    const int c4 = 2 * G + (l >> 5);  // float4 column index: cols 4*c4 .. 4*c4+3
    const int s = tokg / NTOK, n = tokg - s * NTOK;
    // the SAME helpers as the row kernels' patchify prologue (t2s_rows.h): explicit fmaf chains, so the two forms agree bit for
    // bit whatever the compiler would contract on its own (tests/test_hip_contracts.py runs both)
    struct { const float* p_lat; int p_B; const float *p_cw, *p_cb; } pa{x, B, cw, cb};
    float cv[4];
    patch_conv(pa, s, n, cv);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int d = c4 * 4 + e;
        o[e] = patch_feature(cv, *reinterpret_cast<const f32x4*>(pw + d * 4), pb[d], pos[n * D + d]);
    }
    reinterpret_cast<f32x4*>(h)[gid] = o;
}

// fragment-major (rows,128) -> row-major, for the test tap
__global__ void unfrag128_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * D) return;
    const int row = idx >> 7, col = idx & 127;
    dst[idx] = src[frag_index(row, col, D)];
}

}  // namespace t2s

using namespace t2s;

#include "t2s_dit_internal.h"

namespace {

struct ArenaPlan {
    size_t off = 0;
    size_t take(size_t n) {
        size_t o = off;
        off += (n + 63) & ~size_t(63);
        return o;
    }
};

// split planes of the row-chain weights from the handle's packed fp32 copies (T2S_MATH_BF16X3)
int pack_x3_weights(t2s_dit* h, hipStream_t st) {
    int rc;
    for (int i = 0; i < NBLK; ++i) {
        if ((rc = pack_rows_x3(reinterpret_cast<const float*>(h->qkv_p[i]), reinterpret_cast<bf16x8*>(h->qkv3[i]), 3 * D, D, 0, st)) ||
            (rc = pack_rows_x3(reinterpret_cast<const float*>(h->proj_p[i]), reinterpret_cast<bf16x8*>(h->proj3[i]), D, D, 0, st)) ||
            (rc = pack_rows_x3(reinterpret_cast<const float*>(h->fc1_p[i]), reinterpret_cast<bf16x8*>(h->fc13[i]), 2 * D, D, 0, st)) ||
            (rc = pack_rows_x3(reinterpret_cast<const float*>(h->fc2_c[i]), reinterpret_cast<bf16x8*>(h->fc2c3[i]), D, 2 * D, 1, st)))
            return rc;
    }
    return T2S_OK;
}

// Floats the kernels read behind every pointer of t2s_dit_weights, in declaration order (10 top-level, then 10 per block)
struct WeightNeed {
    const char* name;
    size_t floats;
};
constexpr WeightNeed TOP_NEED[10] = {{"conv.weight", 16}, {"conv.bias", 4}, {"patch_emb.weight", 512}, {"patch_emb.bias", 128},
                                     {"pos_embed", (size_t)NTOK * D}, {"ln.weight", D}, {"ln.bias", D},
                                     {"linear_emb_to_patch.weight", 4 * D}, {"linear_emb_to_patch.bias", 4}, {"time_freqs", 64}};
constexpr WeightNeed BLK_NEED[10] = {{"attn.qkv.weight", 3 * D * D}, {"attn.qkv.bias", 3 * D}, {"attn.proj.weight", D * D},
                                     {"attn.proj.bias", D}, {"mlp.fc1.weight", 2 * D * D}, {"mlp.fc1.bias", 2 * D},
                                     {"mlp.fc2.weight", 2 * D * D}, {"mlp.fc2.bias", D}, {"adaLN_modulation.1.weight", (size_t)MODW * D},
                                     {"adaLN_modulation.1.bias", MODW}};
static_assert(sizeof(t2s_dit_weights) == sizeof(void*) * T2S_DIT_N_TENSORS, "t2s_dit_weights is T2S_DIT_N_TENSORS pointers");

// Every tensor must be at least as large as what t2s_dit_create / _update_weights read from it: by the caller's own count
// (n_floats, may be NULL) and -- what no caller can get wrong -- by the extent of the device allocation the pointer lies in
// (hipMemGetAddressRange; a pointer HIP cannot place is passed through: nothing known, nothing refused).  An undersized
// tensor is T2S_E_INVALID here, not a memory fault in a pack kernel (round 4: a 480 x 128 dummy behind the (768,128) adaLN
// matrix read 148 KB past its buffer and faulted only when that happened to end an allocator segment).
int check_weights(const t2s_dit_weights* w, const uint64_t* n_floats, bool ranges) {
    const float* const* ptrs = reinterpret_cast<const float* const*>(w);
    for (int i = 0; i < T2S_DIT_N_TENSORS; ++i) {
        const WeightNeed& need = i < 10 ? TOP_NEED[i] : BLK_NEED[(i - 10) % 10];
        const int blk = i < 10 ? -1 : (i - 10) / 10;
        char name[96];
        if (blk < 0) snprintf(name, sizeof(name), "%s", need.name);
        else snprintf(name, sizeof(name), "layers.%d.%s", blk, need.name);
        T2S_REQUIRE(ptrs[i], "t2s_dit weights: %s is NULL", name);
        T2S_REQUIRE(!n_floats || n_floats[i] >= need.floats, "t2s_dit weights: %s holds %llu floats, the kernels read %zu", name,
                    (unsigned long long)n_floats[i], need.floats);
        if (ranges) {
            char what[128];
            snprintf(what, sizeof(what), "t2s_dit weights: %s", name);
            const int rc = check_device_extent(ptrs[i], need.floats * sizeof(float), what);
            if (rc != T2S_OK) return rc;
        }
    }
    return T2S_OK;
}

int upload_weights(t2s_dit* h, const t2s_dit_weights* w, hipStream_t st) {
    T2S_REQUIRE(w->conv_w && w->conv_b && w->patch_w && w->patch_b && w->pos_embed && w->ln_w &&
                    w->ln_b && w->out_w && w->out_b && w->time_freqs,
                "t2s_dit weights: NULL top-level pointer");
    CopyTable ct{};
    PackTable pt{};
    int nc = 0, np = 0, max_pack = 0;
    auto cp = [&](float* dst, const float* src, int n) { ct.j[nc++] = CopyJob{src, dst, n}; };
    auto pk = [&](const float* W, f32x4* P, int N, int K, int n_offset, int mode) {
        pt.j[np++] = PackJob{W, reinterpret_cast<float*>(P), N, K, n_offset, mode};
        max_pack = N * K > max_pack ? N * K : max_pack;
    };
    cp(h->conv_w, w->conv_w, 16);
    cp(h->conv_b, w->conv_b, 4);
    cp(h->patch_w, w->patch_w, 128 * 4);
    cp(h->patch_b, w->patch_b, 128);
    cp(h->pos, w->pos_embed, NTOK * D);
    cp(h->ln_w, w->ln_w, D);
    cp(h->ln_b, w->ln_b, D);
    cp(h->out_w, w->out_w, 4 * D);
    cp(h->out_b, w->out_b, 4);
    cp(h->freqs, w->time_freqs, 64);
    for (int i = 0; i < NBLK; ++i) {
        const t2s_dit_block_weights& b = w->blk[i];
        T2S_REQUIRE(b.qkv_w && b.qkv_b && b.proj_w && b.proj_b && b.fc1_w && b.fc1_b && b.fc2_w &&
                        b.fc2_b && b.ada_w && b.ada_b,
                    "t2s_dit weights: NULL pointer in block %d", i);
        cp(h->qkv_b[i], b.qkv_b, 3 * D);
        cp(h->proj_b[i], b.proj_b, D);
        cp(h->fc1_b[i], b.fc1_b, 2 * D);
        cp(h->fc2_b[i], b.fc2_b, D);
        cp(h->ada_b + i * MODW, b.ada_b, MODW);
        pk(b.qkv_w, h->qkv_p[i], 3 * D, D, 0, 0);
        pk(b.proj_w, h->proj_p[i], D, D, 0, 0);
        pk(b.fc1_w, h->fc1_p[i], 2 * D, D, 0, 0);
        pk(b.fc2_w, h->fc2_c[i], D, 2 * D, 0, 1);
        pk(b.ada_w, h->ada_p, MODW, D, i * MODW, 0);
        pk(b.qkv_w, h->qkv_p16[i], 3 * D, D, 0, 2);
        pk(b.proj_w, h->proj_p16[i], D, D, 0, 2);
        pk(b.fc1_w, h->fc1_p16[i], 2 * D, D, 0, 2);
        pk(b.fc2_w, h->fc2_c16[i], D, 2 * D, 0, 3);
    }
    static_assert(10 + 5 * NBLK <= MULTI_JOBS && 9 * NBLK <= MULTI_JOBS, "job tables too small");
    copy_multi_kernel<<<dim3(16, nc), 256, 0, st>>>(ct);
    T2S_LAUNCH_CHECK();
    pack_weight_multi_kernel<<<dim3((max_pack + 255) / 256, np), 256, 0, st>>>(pt);
    T2S_LAUNCH_CHECK();
    if (h->w3 != nullptr) return pack_x3_weights(h, st);
    return T2S_OK;
}

// One DiT forward over S sequences (sequence s reads latent row s % B).
// ws_seq0: first workspace slot (in sequences) this pass may use -- the sampler runs two half batches as independent
// lanes on two streams, each in its own slice of the workspace (slots [ws_seq0, ws_seq0 + S))
struct ModTable {          // the sampler's precomputed adaLN table (NULL base: compute this pass's rows here)
    const float* base = nullptr;
    int rows = 0, row0 = 0;
};

int run_forward(t2s_dit* h, const float* x, int B, int S, int uncond_rows, const float* temb,
                int temb_rows, const int* step_ptr, const float* text, float* out0, float* out1,
                int split, hipStream_t st, bool keep_stream = true, int ws_seq0 = 0, ModTable mt = ModTable()) {
    int rc;
    const size_t tok0 = (size_t)ws_seq0 * NTOK * D;
    float* const w_h = h->h + tok0;
    float* const w_q = h->q + tok0;
    float* const w_k = h->k + tok0;
    float* const w_v = h->v + tok0;
    float* const w_ao = h->ao + tok0;
    float* const w_h0 = h->h0 + tok0 / 2;                      // one slot per PAIR of sequences (CFG pass)
    float* const w_mod = h->mod + (size_t)ws_seq0 * MODROW;
    __bf16* const w_k3 = h->k3 ? h->k3 + tok0 * 3 : nullptr;
    __bf16* const w_v3 = h->v3 ? h->v3 + tok0 * 3 : nullptr;
    const bool x3 = h->math == T2S_MATH_BF16X3;
    const bool use_table = mt.base != nullptr && step_ptr != nullptr;
    if (!use_table) {   // adaLN for all 4 blocks at once: mod = silu(c) @ W_ada^T + b, c = t_emb (+ text)
        TimeScope ts(h, TC_OTHER, st);
        adaln_kernel<<<dim3((S + 31) / 32, MODROW / 128), 256, 0, st>>>(w_mod, temb, temb_rows, step_ptr, text,
                                                                       uncond_rows, S, h->ada_p, h->ada_b, 0);
        T2S_LAUNCH_CHECK();
    }
    // patchify: the two branches of a CFG pass (S == 2B) see the same tokens, so only B sequences are
    // computed, into their own buffer; block 0's kernels read sequence s % B from it (in place would race:
    // the workgroup of sequence s overwrites slot s while that of s + B still reads it)
    const bool shared_in = S == 2 * B;
    float* tokens = shared_in ? w_h0 : w_h;
    const int in_seqs = shared_in ? B : S;
    // The <qkv only> row kernel of block 0 patchifies in its prologue (t2s_rows.h, t2s_rows16.h, t2s_rows_x3.h: the same
    // helpers, the same bits as patchify_kernel) and writes the tokens for block 0's <proj + MLP> kernel.
    // T2S_PATCHIFY_KERNEL=1 restores the stand-alone launch (A/B).
    static const bool patch_launch = getenv("T2S_PATCHIFY_KERNEL") && atoi(getenv("T2S_PATCHIFY_KERNEL")) != 0;
    const bool patch_fused = !patch_launch;
    if (!patch_fused) {
        const int threads = in_seqs * NTOK * 32;
        TimeScope ts(h, TC_OTHER, st);
        patchify_kernel<<<(threads + 255) / 256, 256, 0, st>>>(x, B, tokens, in_seqs, h->conv_w, h->conv_b,
                                                               h->patch_w, h->patch_b, h->pos);
        T2S_LAUNCH_CHECK();
    }
    const int M = S * NTOK;
    // Row-local chain as one register-resident kernel per block (t2s_rows.h):
    //   rows<qkv only>(block 0) ; { attention(i) ; rows<proj+MLP of i, qkv of i+1> } x 4
    auto rows_args = [&](int blk, int qkv_blk) {
        RowArgs a{};
        a.x = w_h; a.ao = w_ao; a.mod = w_mod; a.M = M; a.blk = blk; a.qkv_blk = qkv_blk;
        if (use_table) { a.mod = mt.base; a.mod_step = step_ptr; a.mod_rows = mt.rows; a.mod_uncond = uncond_rows; a.mod_row0 = mt.row0; }
        if (blk < 0 && patch_fused) {
            a.p_lat = x; a.p_B = B; a.p_cw = h->conv_w; a.p_cb = h->conv_b; a.p_pw = h->patch_w; a.p_pb = h->patch_b; a.p_pos = h->pos;
        }
        const bool first = blk <= 0 && qkv_blk <= 1;     // rows<qkv 0> and rows<block 0, qkv 1> read the patchified tokens
        a.x_in = first ? tokens : w_h; a.in_seqs = first ? in_seqs : S;
        if (blk == NBLK - 1) {   // the last kernel also runs the final layer
            a.f_lnw = h->ln_w; a.f_lnb = h->ln_b; a.f_ow = h->out_w; a.f_ob = h->out_b;
            a.out0 = out0; a.out1 = out1; a.split = split; a.keep_x = keep_stream;
        }
        if (blk >= 0) {
            a.Wp = h->proj_p[blk]; a.W1 = h->fc1_p[blk]; a.W2c = h->fc2_c[blk];
            a.bp = h->proj_b[blk]; a.b1 = h->fc1_b[blk]; a.b2 = h->fc2_b[blk];
        }
        if (qkv_blk >= 0) { a.Wq = h->qkv_p[qkv_blk]; a.bq = h->qkv_b[qkv_blk]; }
        a.q = w_q; a.k = w_k; a.v = w_v;
        return a;
    };
    // T2S_MATH_BF16X3: every product of the row chain and of the attention is evaluated as six bf16 MFMAs
    // (fp32-accurate, t2s_x3.h); k / V^T travel as split bf16 planes
    auto rows_args_x3 = [&](int blk, int qkv_blk) {
        RowArgsX3 a{};
        a.x = w_h; a.ao = w_ao; a.mod = w_mod; a.M = M; a.blk = blk; a.qkv_blk = qkv_blk;
        if (use_table) { a.mod = mt.base; a.mod_step = step_ptr; a.mod_rows = mt.rows; a.mod_uncond = uncond_rows; a.mod_row0 = mt.row0; }
        if (blk < 0 && patch_fused) {
            a.p_lat = x; a.p_B = B; a.p_cw = h->conv_w; a.p_cb = h->conv_b; a.p_pw = h->patch_w; a.p_pb = h->patch_b; a.p_pos = h->pos;
        }
        const bool first = blk <= 0 && qkv_blk <= 1;
        a.x_in = first ? tokens : w_h; a.in_seqs = first ? in_seqs : S;
        if (blk == NBLK - 1) {
            a.f_lnw = h->ln_w; a.f_lnb = h->ln_b; a.f_ow = h->out_w; a.f_ob = h->out_b;
            a.out0 = out0; a.out1 = out1; a.split = split; a.keep_x = keep_stream;
        }
        if (blk >= 0) {
            a.Wp = reinterpret_cast<const bf16x8*>(h->proj3[blk]); a.W1 = reinterpret_cast<const bf16x8*>(h->fc13[blk]);
            a.W2c = reinterpret_cast<const bf16x8*>(h->fc2c3[blk]);
            a.bp = h->proj_b[blk]; a.b1 = h->fc1_b[blk]; a.b2 = h->fc2_b[blk];
        }
        if (qkv_blk >= 0) { a.Wq = reinterpret_cast<const bf16x8*>(h->qkv3[qkv_blk]); a.bq = h->qkv_b[qkv_blk]; }
        a.q = w_q; a.k3 = w_k3; a.v3 = w_v3;
        return a;
    };
    // Small launches run the row chain on 16-token tiles (t2s_rows16.h; bit-identical results): up to 100 sequences, i.e.
    // where the 32-token kernel has fewer than ~1.5 waves per SIMD.  Same-box series/s, 16- vs 32-token tiles: 8 series
    // (16 sequences) +27 %, 16 +25 %, 24 +1 %, 32 (the 8-GPU strong-scaling shard) +1.2 %, 40 +7.5 %, 48 +7.7 %, 64 -2 %
    // (profiles/r03_rows16_ab.txt; measured with the last block's kernel still on 32-token tiles).
    // T2S_ROWS16_MAX_SEQS moves the switch point (A/B runs; 0 = never).
    static const int rows16_max = getenv("T2S_ROWS16_MAX_SEQS") ? atoi(getenv("T2S_ROWS16_MAX_SEQS")) : 100;
    const bool use16 = !x3 && S <= rows16_max;
    auto rows_args16 = [&](int blk, int qkv_blk) {
        RowArgs a = rows_args(blk, qkv_blk);
        if (blk >= 0) { a.Wp = h->proj_p16[blk]; a.W1 = h->fc1_p16[blk]; a.W2c = h->fc2_c16[blk]; }
        if (qkv_blk >= 0) a.Wq = h->qkv_p16[qkv_blk];
        return a;
    };
    {
        TimeScope ts(h, TC_ROWS_FIRST, st);
        rc = x3 ? launch_dit_rows_x3<false, true>(rows_args_x3(-1, 0), st)
                : (use16 ? launch_dit_rows16<false, true>(rows_args16(-1, 0), st) : launch_dit_rows<false, true>(rows_args(-1, 0), st));
        if (rc != T2S_OK) return rc;
    }
    for (int i = 0; i < NBLK; ++i) {
        {
            TimeScope ts(h, TC_ATTN, st);
            rc = x3 ? launch_attn_x3(w_q, w_k3, w_v3, w_ao, S * NH, st)
                    : launch_attn_packed(w_q, w_k, w_v, w_ao, S * NH, st);
            if (rc != T2S_OK) return rc;
        }
        TimeScope ts(h, i + 1 < NBLK ? TC_ROWS : TC_ROWS_LAST, st);
        if (i + 1 < NBLK)
            rc = x3 ? launch_dit_rows_x3<true, true>(rows_args_x3(i, i + 1), st)
                    : (use16 ? launch_dit_rows16<true, true>(rows_args16(i, i + 1), st) : launch_dit_rows<true, true>(rows_args(i, i + 1), st));
        else
            rc = x3 ? launch_dit_rows_x3<true, false>(rows_args_x3(i, -1), st)
                    : (use16 ? launch_dit_rows16<true, false>(rows_args16(i, -1), st) : launch_dit_rows<true, false>(rows_args(i, -1), st));
        if (rc != T2S_OK) return rc;
    }
    // (the final layer -- LayerNorm, Linear 128 -> 4, unpatchify -- ran inside the last row kernel)
    return T2S_OK;
}

}  // namespace

// exported to the sampler TU
namespace t2s {
int dit_forward_cfg_step(t2s_dit* h, const float* x, const float* temb_table, const int* step_ptr,
                         const float* text, float* out_u, float* out_c, int B, hipStream_t st, int ws_seq0,
                         const float* mod_table, int mod_rows, int mod_row0) {
    ModTable mt;
    mt.base = mod_table; mt.rows = mod_rows; mt.row0 = mod_row0;
    return run_forward(h, x, B, 2 * B, B, temb_table, 1, step_ptr, text, out_u, out_c, B, st, /*keep_stream=*/false,
                       ws_seq0, mt);
}

// The adaLN modulation of EVERY step of a sampling run in one launch: table (steps, B + 1, MODROW), row 0 of a step = the
// text-free branch (identical for the whole batch), row 1 + b = batch row b.  It depends on the step's time embedding and
// the text only, never on the state, so it need not sit on the loop's critical path (10 us of a 640 us step at 32 series):
// 3.2 GB of the 288 GB at B = 256 / 1000 steps, computed in a few ms per run.  f32 arithmetic: the same MFMA order per
// row as the per-step launch, i.e. the same bits.
int dit_adaln_table(t2s_dit* h, const float* temb_table, int steps, const float* text, int B, float* table, hipStream_t st) {
    const long long rows = (long long)steps * (B + 1);
    if (rows <= 0 || rows > 0x7fffffffLL / 2) {
        set_error("dit_adaln_table: %lld rows out of range", rows);
        return T2S_E_INVALID;
    }
    adaln_kernel<<<dim3((unsigned)((rows + 31) / 32), MODROW / 128), 256, 0, st>>>(table, temb_table, steps, nullptr, text, 0,
                                                                                  (int)rows, h->ada_p, h->ada_b, B + 1);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}
}  // namespace t2s

// ------------------------------------------------------------------ C ABI
extern "C" {

const char* t2s_last_error(void) { return t2s::g_err; }
// 0.3: + t2s_philox_uniform, t2s_dit_forward_cfg_rows, t2s_sampler_set_loop_graph (additions only)
// 0.4: + t2s_time_embedding_freqs, t2s_dit_weights_check, t2s_mse_ws, t2s_vae_update_weights, t2s_vae_encode_backward (additions only)
// 0.5: + t2s_mlp_pack, t2s_mlp_forward, t2s_mlp_backward (additions only)
const char* t2s_version(void) { return "t2s 0.5 gfx950 fp32-mfma"; }

int t2s_dit_create(const t2s_dit_weights* w, int max_seqs, t2s_dit** out) {
    T2S_REQUIRE(w && out, "t2s_dit_create: NULL argument");
    T2S_REQUIRE(max_seqs > 0 && max_seqs <= 65536, "t2s_dit_create: max_seqs=%d out of range", max_seqs);
    {
        const int rc_w = check_weights(w, nullptr, /*ranges=*/true);
        if (rc_w != T2S_OK) return rc_w;
    }
    t2s_dit* h = new t2s_dit();
    h->max_seqs = max_seqs;
    ArenaPlan p;
    const size_t o_conv_w = p.take(16), o_conv_b = p.take(4), o_patch_w = p.take(512),
                 o_patch_b = p.take(128), o_pos = p.take(NTOK * D), o_ln_w = p.take(D),
                 o_ln_b = p.take(D), o_out_w = p.take(4 * D), o_out_b = p.take(4),
                 o_freqs = p.take(64), o_ada_b = p.take(MODROW), o_ada_p = p.take((size_t)MODROW * D);
    size_t o_qkv_b[NBLK], o_proj_b[NBLK], o_fc1_b[NBLK], o_fc2_b[NBLK];
    size_t o_qkv_p[NBLK], o_proj_p[NBLK], o_fc1_p[NBLK], o_fc2_c[NBLK];
    size_t o_qkv_p16[NBLK], o_proj_p16[NBLK], o_fc1_p16[NBLK], o_fc2_c16[NBLK];
    for (int i = 0; i < NBLK; ++i) {
        o_qkv_b[i] = p.take(3 * D); o_proj_b[i] = p.take(D); o_fc1_b[i] = p.take(2 * D);
        o_fc2_b[i] = p.take(D);
        o_qkv_p[i] = p.take(3 * D * D); o_proj_p[i] = p.take(D * D);
        o_fc1_p[i] = p.take(2 * D * D); o_fc2_c[i] = p.take(2 * D * D);
        o_qkv_p16[i] = p.take(3 * D * D); o_proj_p16[i] = p.take(D * D);
        o_fc1_p16[i] = p.take(2 * D * D); o_fc2_c16[i] = p.take(2 * D * D);
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
        h->fc2_c[i] = reinterpret_cast<f32x4*>(A + o_fc2_c[i]);
        h->qkv_p16[i] = reinterpret_cast<f32x4*>(A + o_qkv_p16[i]);
        h->proj_p16[i] = reinterpret_cast<f32x4*>(A + o_proj_p16[i]);
        h->fc1_p16[i] = reinterpret_cast<f32x4*>(A + o_fc1_p16[i]);
        h->fc2_c16[i] = reinterpret_cast<f32x4*>(A + o_fc2_c16[i]);
    }
    const size_t S = (size_t)max_seqs, tokD = S * NTOK * D;
    float** bufs[] = {&h->h, &h->q, &h->k, &h->v, &h->ao, &h->mod, &h->h0};
    const size_t sizes[] = {tokD, tokD, tokD, tokD, tokD, S * MODROW, (S / 2 + 1) * NTOK * D};
    for (int i = 0; i < 7; ++i) {
        e = hipMalloc(bufs[i], sizes[i] * sizeof(float));
        if (e != hipSuccess) {
            set_error("t2s_dit_create: hipMalloc(workspace %d, %zu B) failed: %s", i,
                      sizes[i] * sizeof(float), hipGetErrorString(e));
            t2s_dit_destroy(h);
            return T2S_E_HIP;
        }
    }
    int rc = attn_init();
    if (rc == T2S_OK) rc = dit_rows_init();
    if (rc == T2S_OK) rc = dit_rows16_init();
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
    const int rc_w = check_weights(w, nullptr, /*ranges=*/true);
    if (rc_w != T2S_OK) return rc_w;
    return upload_weights(h, w, (hipStream_t)stream);
}

int t2s_dit_set_math(t2s_dit* h, int math) {
    T2S_REQUIRE(h, "t2s_dit_set_math: NULL handle");
    T2S_REQUIRE(math == T2S_MATH_F32 || math == T2S_MATH_BF16X3, "t2s_dit_set_math: unknown mode %d", math);
    if (math == T2S_MATH_BF16X3 && h->k3 == nullptr) {
        int rc;
        if ((rc = attn_x3_init()) || (rc = dit_rows_x3_init<false, true>()) || (rc = dit_rows_x3_init<true, true>()) ||
            (rc = dit_rows_x3_init<true, false>()))
            return rc;
        const size_t bytes = (size_t)h->max_seqs * NTOK * D * 3 * sizeof(__bf16);   // three bf16 planes
        const size_t wvals = (size_t)NBLK * (3 + 1 + 2 + 2) * D * D * 3;              // qkv, proj, fc1, fc2: 3 planes
        if (hipMalloc(&h->k3, bytes) != hipSuccess || hipMalloc(&h->v3, bytes) != hipSuccess ||
            hipMalloc(&h->w3, wvals * sizeof(__bf16)) != hipSuccess) {
            if (h->k3) (void)hipFree(h->k3);
            if (h->v3) (void)hipFree(h->v3);
            h->k3 = h->v3 = h->w3 = nullptr;
            set_error("t2s_dit_set_math: hipMalloc(2 x %zu B + weights) failed", bytes);
            return T2S_E_HIP;
        }
        __bf16* p = h->w3;
        for (int i = 0; i < NBLK; ++i) {
            h->qkv3[i] = p; p += (size_t)3 * D * D * 3;
            h->proj3[i] = p; p += (size_t)D * D * 3;
            h->fc13[i] = p; p += (size_t)2 * D * D * 3;
            h->fc2c3[i] = p; p += (size_t)2 * D * D * 3;
        }
        if ((rc = pack_x3_weights(h, nullptr))) return rc;
        T2S_HIP_CHECK(hipStreamSynchronize(nullptr));
    }
    h->math = math;
    return T2S_OK;
}

void t2s_dit_destroy(t2s_dit* h) {
    if (!h) return;
    t2s::train_free(h);
    if (h->k3) (void)hipFree(h->k3);
    if (h->v3) (void)hipFree(h->v3);
    if (h->w3) (void)hipFree(h->w3);
    float* bufs[] = {h->arena, h->h, h->q, h->k, h->v, h->ao, h->mod, h->h0};
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

int t2s_time_embedding_freqs(const float* freqs, const float* t, float* out, int B, void* stream) {
    T2S_REQUIRE(freqs && t && out && B > 0, "t2s_time_embedding_freqs: bad argument");
    time_embedding_kernel<<<(B * 64 + 255) / 256, 256, 0, (hipStream_t)stream>>>(t, freqs, out, B);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

int t2s_dit_weights_check(const t2s_dit_weights* w, const uint64_t* n_floats, int n_entries) {
    T2S_REQUIRE(w, "t2s_dit_weights_check: NULL weights");
    T2S_REQUIRE(!n_floats || n_entries == T2S_DIT_N_TENSORS, "t2s_dit_weights_check: n_entries=%d, t2s_dit_weights has %d tensors", n_entries,
                T2S_DIT_N_TENSORS);
    return check_weights(w, n_floats, /*ranges=*/true);
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
                                     (hipStream_t)stream, 0, nullptr, 0, 0);
}

int t2s_dit_forward_cfg_rows(t2s_dit* h, const float* x, const float* temb, int temb_rows, const float* text,
                             float* out_uncond, float* out_cond, int B, void* stream) {
    T2S_REQUIRE(h && x && temb && text && out_uncond && out_cond, "t2s_dit_forward_cfg_rows: NULL argument");
    T2S_REQUIRE(B > 0 && 2 * B <= h->max_seqs, "t2s_dit_forward_cfg_rows: 2*B=%d exceeds max_seqs=%d", 2 * B, h->max_seqs);
    T2S_REQUIRE(temb_rows == 1 || temb_rows == B, "t2s_dit_forward_cfg_rows: temb_rows=%d must be 1 or B=%d", temb_rows, B);
    return run_forward(h, x, B, 2 * B, B, temb, temb_rows, nullptr, text, out_uncond, out_cond, B, (hipStream_t)stream,
                            /*keep_stream=*/false);
}

int t2s_dit_timing_begin(t2s_dit* h) {
    T2S_REQUIRE(h, "t2s_dit_timing_begin: NULL handle");
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    h->ev_pool.clear();
    h->ev_class.clear();
    h->timing = true;
    return T2S_OK;
}

int t2s_dit_timing_end(t2s_dit* h, double* out6) { return t2s_dit_timing_end_ex(h, out6, 3); }

int t2s_dit_timing_end_ex(t2s_dit* h, double* out, int n_classes) {
    T2S_REQUIRE(h && out && n_classes > 0 && n_classes <= t2s::TC_COUNT, "t2s_dit_timing_end_ex: bad argument (n_classes=%d)", n_classes);
    h->timing = false;
    for (int i = 0; i < 2 * n_classes; ++i) out[i] = 0.0;
    for (size_t i = 0; i < h->ev_class.size(); ++i) {
        T2S_HIP_CHECK(hipEventSynchronize(h->ev_pool[2 * i + 1]));
        float ms = 0.f;
        T2S_HIP_CHECK(hipEventElapsedTime(&ms, h->ev_pool[2 * i], h->ev_pool[2 * i + 1]));
        const int cls = h->ev_class[i];
        if (cls == TC_ROWS_FIRST || cls == TC_ROWS_LAST) {      // class 1 = every row-chain launch
            if (n_classes > TC_ROWS) out[2 * TC_ROWS] += ms, out[2 * TC_ROWS + 1] += 1.0;
        }
        if (cls >= n_classes) continue;
        out[2 * cls] += ms;
        out[2 * cls + 1] += 1.0;
    }
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    h->ev_pool.clear();
    h->ev_class.clear();
    return T2S_OK;
}


int t2s_dit_read_stream(const t2s_dit* h, float* out, int S, void* stream) {
    T2S_REQUIRE(h && out && S > 0 && S <= h->max_seqs, "t2s_dit_read_stream: bad argument");
    const int total = S * NTOK * D;
    unfrag128_kernel<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(h->h, out, S * NTOK);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

}  // extern "C"
