// Training step of the DiT for MI355X (reference train.py:101-127): forward that keeps the
// activations, hand-written backward for every trainable tensor, fused AdamW.
//
//   forward  (per block)  x_in -LN1,mod-> a1 -qkv-> q,k,v -attn-> o -proj-> p ; x_mid = x_in + g1 p
//                         x_mid -LN2,mod-> a2 -fc1-> u -gelu,fc2-> f ; x_out = x_mid + g2 f
//   backward (reverse)    gate/residual, fc2 (dgrad+wgrad), gelu', fc1, LN2/mod, proj, attention
//                         (dQ kernel + dK/dV kernel, P recomputed from the saved log-sum-exp), qkv,
//                         LN1/mod; then patchify, final layer and the adaLN linear.
//
// Round-1 design: correctness first, fp32 end to end, activations row-major.  Forward linears and
// every dgrad reuse gemm_rows_kernel (dgrad = the same kernel on transposed-packed weights);
// weight gradients are one MFMA kernel (A = dY^T, B = X straight from row-major global memory,
// contraction over token rows, fp32 atomics into the gradient); per-sequence reductions (adaLN
// shift / scale / gate gradients, LayerNorm backward) are deterministic one-workgroup-per-sequence
// kernels.  Parity: tests/test_hip_train.py against autograd through the CPU oracle and the
// reference-generated fixture tests/golden/train_step.npz.
#include <vector>

#include "t2s_wgrad.h"
#include "t2s_dit_internal.h"

using namespace t2s;

namespace t2s {
int attn_plain_train_fwd(const float* q, const float* k, const float* v, float* o_rows, float* lse, int BH,
                         hipStream_t st);
int attn_bwd(const float* q, const float* k, const float* v, const float* o_rows, const float* do_rows,
             const float* lse, float* dsum, float* dqkv_rows, int BH, hipStream_t st);
void train_free(t2s_dit* h);
int attn16_train_fwd(const __bf16* q, const __bf16* k, const __bf16* v, __bf16* o_rows, float* lse, int BH, hipStream_t st);
int attn16_bwd(const __bf16* q, const __bf16* k, const __bf16* v, const __bf16* o_rows, const __bf16* do_rows,
               const float* lse, float* dsum, __bf16* dqkv_rows, int BH, hipStream_t st);
}

// ------------------------------------------------------------------ workspace
struct t2s_train_ws {
    int cap_seqs = 0;
    int dtype = 0;        // T2S_TRAIN_F32 / T2S_TRAIN_BF16 this workspace was laid out for
    int S = 0;            // sequences of the last forward
    bool has_text = false;
    // packed weights for training (tile-major, both orientations), refreshed by pack_train_weights
    float* warena = nullptr;
    f32x4 *qkv_f[NBLK], *proj_f[NBLK], *fc1_f[NBLK], *fc2_f[NBLK];   // forward:  W   (N,K)
    f32x4 *qkv_t[NBLK], *proj_t[NBLK], *fc1_t[NBLK], *fc2_t[NBLK];   // dgrad:    W^T (K,N)
    f32x4* out_t = nullptr;                                           // unused (final layer is VALU)
    // raw (reference-layout) copies the VALU backward kernels read
    float *w_out = nullptr, *w_pe = nullptr, *w_conv = nullptr, *ln_g = nullptr;
    // saved activations
    float* act = nullptr;
    float *x_in[NBLK + 1], *a1[NBLK], *q[NBLK], *k[NBLK], *v[NBLK], *o[NBLK], *lse[NBLK], *p[NBLK],
        *x_mid[NBLK], *a2[NBLK], *u[NBLK], *f[NBLK];
    float *silu_c = nullptr, *mod = nullptr, *c = nullptr, *lat = nullptr;   // (S,128) (S,3072) (S,128) (S,1920)
    // backward temporaries
    float *dx = nullptr, *t1 = nullptr, *t2a = nullptr, *t2b = nullptr, *t3 = nullptr, *t4 = nullptr, *dsum = nullptr,
          *dmod = nullptr;
    // ---- T2S_TRAIN_BF16: packed bf16 weights (forward W and dgrad W^T) and bf16 activations; the
    // residual stream (x_in, x_mid, dx), lse, dsum, mod, dmod above stay fp32
    __bf16* warena16 = nullptr;
    bf16x8 *qkv_f16[NBLK], *proj_f16[NBLK], *fc1_f16[NBLK], *fc2_f16[NBLK];
    bf16x8 *qkv_t16[NBLK], *proj_t16[NBLK], *fc1_t16[NBLK], *fc2_t16[NBLK];
    __bf16* act16 = nullptr;
    __bf16 *a1h[NBLK], *qh[NBLK], *kh[NBLK], *vh[NBLK], *oh[NBLK], *ph[NBLK], *a2h[NBLK], *uh[NBLK], *fh[NBLK];
    __bf16 *t1h = nullptr, *t2h = nullptr, *t3h = nullptr, *t4h = nullptr;   // (M,128) (M,256) (M,384) (M,128)
    float* wg_scratch = nullptr;   // partial weight-gradient tiles (wgrad16 stage 1 -> stage 2)
    float* colpart = nullptr;      // bf16: (M/32, 3, 128) column-sum partials of the fused LayerNorm backward (BEPI_LNBWD)
    size_t wg_scratch_floats = 0;
    int n_cu = 0;
};

namespace {

size_t r64(size_t n) { return (n + 63) & ~size_t(63); }

// W (N,K) row-major -> packed_index order of W (transpose=0) or of W^T (K,N) (transpose=1)
__global__ void pack_any_kernel(const float* __restrict__ W, float* __restrict__ P, int N, int K, int transpose) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * K) return;
    const int n = idx / K, k = idx - n * K;
    if (!transpose)
        P[packed_index(n, k, K)] = W[idx];
    else
        P[packed_index(k, n, N)] = W[idx];   // W^T is (K rows, N cols)
}

int pack_any(const float* W, f32x4* P, int N, int K, int transpose, hipStream_t st) {
    pack_any_kernel<<<(N * K + 255) / 256, 256, 0, st>>>(W, reinterpret_cast<float*>(P), N, K, transpose);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

// ---------------------------------------------------------------- elementwise / per-sequence kernels
// float4-granular access to a branch tensor stored as fp32 (T2S_TRAIN_F32) or bf16 (T2S_TRAIN_BF16)
__device__ __forceinline__ f32x4 ld4(const float* p, size_t i4) { return reinterpret_cast<const f32x4*>(p)[i4]; }
__device__ __forceinline__ f32x4 ld4(const __bf16* p, size_t i4) { return unpack4(reinterpret_cast<const bf16x4*>(p)[i4]); }
__device__ __forceinline__ void st4(float* p, size_t i4, f32x4 v) { reinterpret_cast<f32x4*>(p)[i4] = v; }
__device__ __forceinline__ void st4(__bf16* p, size_t i4, f32x4 v) { reinterpret_cast<bf16x4*>(p)[i4] = pack4(v); }

// x_out = x_in + gate[seq] * f      (rows of 128)
template <typename T>
__global__ __launch_bounds__(256) void gate_res_kernel(const float* __restrict__ xin, const T* __restrict__ f,
                                                       const float* __restrict__ mod, int gate_off,
                                                       float* __restrict__ xout, int M) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // float4 index
    if (idx >= M * 32) return;
    const int row = idx >> 5, c4 = idx & 31;
    const int seq = row / NTOK;
    const f32x4 g = *reinterpret_cast<const f32x4*>(mod + (size_t)seq * MODROW + gate_off + c4 * 4);
    const f32x4 a = reinterpret_cast<const f32x4*>(xin)[idx];
    const f32x4 b = ld4(f, idx);
    reinterpret_cast<f32x4*>(xout)[idx] = a + g * b;
}

// one workgroup per sequence: t = gate * dx ; dgate[seq] = sum_tok dx * f
template <typename T>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ dx, const T* __restrict__ f,
                                                       const float* __restrict__ mod, int gate_off,
                                                       T* __restrict__ t, float* __restrict__ dmod) {
    __shared__ f32x4 red[8][32];
    const int seq = blockIdx.x;
    const int c4 = threadIdx.x & 31, rg = threadIdx.x >> 5;   // 8 row groups
    const f32x4 g = *reinterpret_cast<const f32x4*>(mod + (size_t)seq * MODROW + gate_off + c4 * 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int tok = rg; tok < NTOK; tok += 8) {
        const size_t idx = ((size_t)seq * NTOK + tok) * 32 + c4;
        const f32x4 d = reinterpret_cast<const f32x4*>(dx)[idx];
        const f32x4 fv = ld4(f, idx);
        st4(t, idx, g * d);
        acc += d * fv;
    }
    red[rg][c4] = acc;
    __syncthreads();
    if (rg == 0) {
        f32x4 s = red[0][c4];
#pragma unroll
        for (int i = 1; i < 8; ++i) s += red[i][c4];
        *reinterpret_cast<f32x4*>(dmod + (size_t)seq * MODROW + gate_off + c4 * 4) = s;
    }
}

// one workgroup per sequence.  da = grad wrt modulate(LN(x)); computes
//   dshift = sum_tok da, dscale = sum_tok da * n, dn = da * (1 + scale),
//   dx += rstd * (dn - mean(dn) - n * mean(dn * n))          (LayerNorm backward, eps 1e-6, no affine)
// and, when f_next != NULL, the gate backward of the branch that is differentiated next
// (gate_bwd_kernel) on the dx it has just produced: t_next = gate_next * dx, dgate_next = sum_tok dx * f_next
// -- that saves one read of the fp32 gradient stream per branch.
template <typename T>
__global__ __launch_bounds__(256) void ln_mod_bwd_kernel(const T* __restrict__ da, const float* __restrict__ x,
                                                         const float* __restrict__ mod, int shift_off, int scale_off,
                                                         float* __restrict__ dx, float* __restrict__ dmod,
                                                         const T* __restrict__ f_next, int gate_next_off,
                                                         T* __restrict__ t_next) {
    __shared__ f32x4 red[3][8][32];
    const int seq = blockIdx.x;
    const int c4 = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(mod + (size_t)seq * MODROW + scale_off + c4 * 4);
    f32x4 gn = {0.f, 0.f, 0.f, 0.f};
    if (f_next != nullptr) gn = *reinterpret_cast<const f32x4*>(mod + (size_t)seq * MODROW + gate_next_off + c4 * 4);
    f32x4 a_sh = {0.f, 0.f, 0.f, 0.f}, a_sc = {0.f, 0.f, 0.f, 0.f}, a_gt = {0.f, 0.f, 0.f, 0.f};
    for (int tok = rg; tok < NTOK; tok += 8) {
        const size_t idx = ((size_t)seq * NTOK + tok) * 32 + c4;
        const f32x4 xv = reinterpret_cast<const f32x4*>(x)[idx];
        const f32x4 g = ld4(da, idx);
        float s = (xv.x + xv.y) + (xv.z + xv.w);
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s * (1.0f / 128.0f);
        const f32x4 d = xv - mean;
        float ss = (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float rstd = rsqrtf(ss * (1.0f / 128.0f) + 1e-6f);
        const f32x4 n = d * rstd;
        a_sh += g;
        a_sc += g * n;
        const f32x4 dn = g * (1.0f + sc);
        float m1 = (dn.x + dn.y) + (dn.z + dn.w);
        float m2 = (dn.x * n.x + dn.y * n.y) + (dn.z * n.z + dn.w * n.w);
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) {
            m1 += __shfl_xor(m1, o, 64);
            m2 += __shfl_xor(m2, o, 64);
        }
        m1 *= (1.0f / 128.0f);
        m2 *= (1.0f / 128.0f);
        const f32x4 r = reinterpret_cast<const f32x4*>(dx)[idx] + (dn - m1 - n * m2) * rstd;
        reinterpret_cast<f32x4*>(dx)[idx] = r;
        if (f_next != nullptr) {
            st4(t_next, idx, gn * r);
            a_gt += r * ld4(f_next, idx);
        }
    }
    red[0][rg][c4] = a_sh;
    red[1][rg][c4] = a_sc;
    red[2][rg][c4] = a_gt;
    __syncthreads();
    if (rg < 3) {
        if (rg == 2 && f_next == nullptr) return;
        f32x4 s = red[rg][0][c4];
#pragma unroll
        for (int i = 1; i < 8; ++i) s += red[rg][i][c4];
        const int off = rg == 0 ? shift_off : (rg == 1 ? scale_off : gate_next_off);
        *reinterpret_cast<f32x4*>(dmod + (size_t)seq * MODROW + off + c4 * 4) = s;
    }
}

__global__ __launch_bounds__(256) void gelu_kernel(const float* __restrict__ u, float* __restrict__ out, size_t n4) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n4) return;
    f32x4 v = reinterpret_cast<const f32x4*>(u)[idx];
    v.x = gelu_tanh(v.x); v.y = gelu_tanh(v.y); v.z = gelu_tanh(v.z); v.w = gelu_tanh(v.w);
    reinterpret_cast<f32x4*>(out)[idx] = v;
}

__global__ __launch_bounds__(256) void silu_kernel(const float* __restrict__ c, float* __restrict__ out, int n) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float v = c[idx];
    out[idx] = v / (1.0f + __expf(-v));
}

// out[n] += sum_rows Y[row][n]   (N <= 3072, N % 4 == 0); slab of rows per workgroup, fp32 atomics
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ Y, float* __restrict__ out, int M, int N,
                                                     int rows_per_wg) {
    const int r0 = blockIdx.x * rows_per_wg;
    const int r1 = min(M, r0 + rows_per_wg);
    for (int n = threadIdx.x; n < N; n += 256) {
        float acc = 0.f;
        for (int r = r0; r < r1; ++r) acc += Y[(size_t)r * N + n];
        atomicAdd(out + n, acc);
    }
}

// Rows per workgroup of the two "tail" backward kernels below.  Their small gradients are reduced in the workgroup and
// written as ONE partial row per workgroup (FINAL_COLS / PATCH_COLS floats); tail_reduce_kernel then adds the partial
// rows in workgroup order -- deterministic, and without the ~800 K same-address fp32 atomics the round-1 kernels
// flushed with (they were most of these kernels' 0.3 ms each at B=1152).
constexpr int TAIL_ROWS = 256;
constexpr int FINAL_COLS = 772;   // ln.weight 128 | ln.bias 128 | linear_emb_to_patch.weight 4x128 | .bias 4
constexpr int PATCH_COLS = 660;   // patch_emb.weight 128x4 | .bias 128 | conv.weight 16 | conv.bias 4

struct TailDst { float* p[4]; int n[4]; };   // consecutive column ranges of a partial row -> gradient tensors
// grid = cols / 32 workgroups of 32 columns x 8 row slices: slice j adds partial rows j, j+8, ... (independent loads, many in
// flight), then the 8 slice sums are added in slice order.  (One thread per column walking all ~1000 rows serially paid a
// full memory latency per row: 260 us.)
__global__ __launch_bounds__(256) void tail_reduce_kernel(const float* __restrict__ part, int n_wg, int cols, int stride, const TailDst d) {
    __shared__ float red[8][32];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (c < cols) {
#pragma unroll 8
        for (int w = sl; w < n_wg; w += 8) s += part[(size_t)w * stride + c];
    }
    red[sl][cl] = s;
    __syncthreads();
    if (sl != 0 || c >= cols) return;
#pragma unroll
    for (int j = 1; j < 8; ++j) s += red[j][cl];
    int off = c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (off < d.n[i]) { d.p[i][off] = s; return; }
        off -= d.n[i];
    }
}

// dgate of the last block's MLP branch from the fused final-layer backward: three workgroup partials per sequence
__global__ void final_gate_reduce_kernel(const float* __restrict__ part, int stride, float* __restrict__ dmod, int gate_off) {
    const int seq = blockIdx.x, f = threadIdx.x;
    const float* p = part + (size_t)seq * 3 * stride + FINAL_COLS + f;
    dmod[(size_t)seq * MODROW + gate_off + f] = (p[0] + p[stride]) + p[2 * stride];
}

// final layer backward (transformer.py:182-191): dout (S,64,30) -> dx (M,128) and grads of
// ln.weight, ln.bias, linear_emb_to_patch.{weight,bias}.  32 lanes per token row.
// FUSED (bf16 training): the layer input x_mid + gate * f of the last block is formed here instead of being read back
// (the forward never writes it), and the gate backward of that MLP branch runs on the dx just produced: t = gate * dx
// (bf16 rows), dgate partial = sum over this workgroup's rows of dx * f in columns FINAL_COLS.. of its partial row.
// FUSED workgroups own ROWS = 160 rows = a third of a sequence, so three consecutive partial rows make one sequence.
template <bool FUSED, int ROWS>
__global__ __launch_bounds__(256) void final_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dout,
                                                        const float* __restrict__ lng, const float* __restrict__ lnb,
                                                        const float* __restrict__ ow, float* __restrict__ dx,
                                                        float* __restrict__ part, int M, const __bf16* __restrict__ f,
                                                        const float* __restrict__ mod, int gate_off, __bf16* __restrict__ t) {
    constexpr int PCOLS = FUSED ? FINAL_COLS + D : FINAL_COLS;
    __shared__ f32x4 red[FUSED ? 7 : 6][8][32];
    __shared__ float redb[8][4];
    const int c4 = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const f32x4 gam = *reinterpret_cast<const f32x4*>(lng + c4 * 4);
    const f32x4 bet = *reinterpret_cast<const f32x4*>(lnb + c4 * 4);
    f32x4 w[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) w[p] = *reinterpret_cast<const f32x4*>(ow + p * D + c4 * 4);
    f32x4 a_g = {0, 0, 0, 0}, a_b = {0, 0, 0, 0}, a_w[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a_ob[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 a_gt = {0, 0, 0, 0}, gt = {0, 0, 0, 0};
    const int row0 = blockIdx.x * ROWS;
    if constexpr (FUSED) gt = *reinterpret_cast<const f32x4*>(mod + (size_t)(row0 / NTOK) * MODROW + gate_off + c4 * 4);
    static_assert(ROWS % 32 == 0, "8 row groups x 4 rows in flight");
    for (int it0 = 0; it0 < ROWS / 8; it0 += 4) {      // 4 rows per 32-lane group in flight: loads first, then the chains
        f32x4 xv4[4], fv4[4];
        float dl4[4][4];
        bool valid4[4];
        size_t idx4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = rg + 8 * (it0 + u);
            valid4[u] = row0 + rr < M;                      // rows past the end: clamped, their contributions masked
            const int row = valid4[u] ? row0 + rr : M - 1;
            idx4[u] = (size_t)row * 32 + c4;
            xv4[u] = reinterpret_cast<const f32x4*>(x)[idx4[u]];
            if constexpr (FUSED) {
                fv4[u] = ld4(f, idx4[u]);
                xv4[u] = xv4[u] + gt * fv4[u];              // the expression of gate_res_kernel
            }
            // gather d(lin)[p] from the unpatchified output gradient
            const int seq = row / NTOK, tok = row - seq * NTOK;
            const int hh = tok >> 5, ww = tok & 31;
            const float keep = valid4[u] ? 1.f : 0.f;
#pragma unroll
            for (int p = 0; p < 4; ++p)
                dl4[u][p] = keep * dout[(size_t)seq * LAT + (2 * ww + (p & 1)) * LATW + 2 * hh + (p >> 1)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x4 xv = xv4[u];
            float s = (xv.x + xv.y) + (xv.z + xv.w);
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
            const float mean = s * (1.0f / 128.0f);
            const f32x4 d = xv - mean;
            float ss = (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
            const float rstd = rsqrtf(ss * (1.0f / 128.0f) + 1e-5f);
            const f32x4 n = d * rstd;
            const f32x4 y = n * gam + bet;
            f32x4 dy = {0, 0, 0, 0};
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                dy += w[p] * dl4[u][p];
                a_w[p] += y * dl4[u][p];
                a_ob[p] += dl4[u][p];
            }
            a_g += dy * n;
            a_b += dy;
            const f32x4 dn = dy * gam;
            float m1 = (dn.x + dn.y) + (dn.z + dn.w);
            float m2 = (dn.x * n.x + dn.y * n.y) + (dn.z * n.z + dn.w * n.w);
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) {
                m1 += __shfl_xor(m1, o, 64);
                m2 += __shfl_xor(m2, o, 64);
            }
            m1 *= (1.0f / 128.0f);
            m2 *= (1.0f / 128.0f);
            const f32x4 r = (dn - m1 - n * m2) * rstd;
            if (valid4[u]) reinterpret_cast<f32x4*>(dx)[idx4[u]] = r;
            if constexpr (FUSED) {
                if (valid4[u]) st4(t, idx4[u], gt * r);
                a_gt += (valid4[u] ? 1.f : 0.f) * (r * fv4[u]);
            }
        }
    }
    red[0][rg][c4] = a_g;
    red[1][rg][c4] = a_b;
#pragma unroll
    for (int p = 0; p < 4; ++p) red[2 + p][rg][c4] = a_w[p];
    if constexpr (FUSED) red[6][rg][c4] = a_gt;
    if (c4 == 0)
#pragma unroll
        for (int p = 0; p < 4; ++p) redb[rg][p] = a_ob[p];
    __syncthreads();
    if (rg < 6) {
        f32x4 s = red[rg][0][c4];
#pragma unroll
        for (int i = 1; i < 8; ++i) s += red[rg][i][c4];
        *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.x * PCOLS + rg * D + c4 * 4) = s;   // lnw | lnb | ow[p]
    } else if (rg == 6) {
        if (c4 < 4) {
            float s = 0.f;
            for (int i = 0; i < 8; ++i) s += redb[i][c4];
            part[(size_t)blockIdx.x * PCOLS + 6 * D + c4] = s;
        }
    } else if constexpr (FUSED) {
        f32x4 s = red[6][0][c4];
#pragma unroll
        for (int i = 1; i < 8; ++i) s += red[6][i][c4];
        *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.x * PCOLS + FINAL_COLS + c4 * 4) = s;   // dgate partial
    }
}

// patchify backward (transformer.py:166-172): dtok (M,128) -> grads of patch_emb.{weight,bias}, conv.{weight,bias}.
// With g = dtok row, px = the token's 2x2 latent patch, cv = conv(px):
//   d patch_emb.weight[d][c] = sum_rows g[d] cv[c] = sum_ij conv.w[c][ij] G[d][ij] + conv.b[c] PB[d]
//   d conv.weight[c][ij]     = sum_rows (sum_d g[d] Wpe[d][c]) px[ij] = sum_d Wpe[d][c] G[d][ij],   d conv.bias[c] = sum_d Wpe[d][c] PB[d]
// where G[d][ij] = sum_rows g[d] px[ij] and PB[d] = sum_rows g[d]: the row loop only accumulates G and PB (20 FMAs per
// lane and row, no cross-lane step) and the contractions with the weights run once per workgroup.
__global__ __launch_bounds__(256) void patchify_bwd_kernel(const float* __restrict__ dtok, const float* __restrict__ lat,
                                                           int B, const float* __restrict__ cw, const float* __restrict__ cb,
                                                           const float* __restrict__ pw, float* __restrict__ part, int M) {
    __shared__ f32x4 red[5][8][32];
    __shared__ float Gs[4][D], PBs[D];
    const int c4 = threadIdx.x & 31, rg = threadIdx.x >> 5;
    f32x4 G[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};   // [ij] over the lane's 4 features
    f32x4 a_pb = {0, 0, 0, 0};
    const int row0 = blockIdx.x * TAIL_ROWS;
    for (int it0 = 0; it0 < TAIL_ROWS / 8; it0 += 4) {
        f32x4 g4[4];
        float px4[4][4];   // [u][i*2+j] = in[2hh+i][2ww+j]
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = rg + 8 * (it0 + u);
            const bool valid = row0 + rr < M;
            const int row = valid ? row0 + rr : M - 1;
            g4[u] = reinterpret_cast<const f32x4*>(dtok)[(size_t)row * 32 + c4];
            if (!valid) g4[u] = g4[u] * 0.f;
            const int seq = row / NTOK, tok = row - seq * NTOK;
            const int hh = tok >> 5, ww = tok & 31;
            const float* xin = lat + (size_t)(seq % B) * LAT;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) px4[u][i * 2 + jj] = xin[(2 * ww + jj) * LATW + 2 * hh + i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int ij = 0; ij < 4; ++ij) G[ij] += g4[u] * px4[u][ij];
            a_pb += g4[u];
        }
    }
#pragma unroll
    for (int ij = 0; ij < 4; ++ij) red[ij][rg][c4] = G[ij];
    red[4][rg][c4] = a_pb;
    __syncthreads();
    if (rg < 5) {
        f32x4 s = red[rg][0][c4];
#pragma unroll
        for (int i = 1; i < 8; ++i) s += red[rg][i][c4];
        if (rg < 4) *reinterpret_cast<f32x4*>(&Gs[rg][c4 * 4]) = s;
        else *reinterpret_cast<f32x4*>(&PBs[c4 * 4]) = s;
    }
    __syncthreads();
    float* prow = part + (size_t)blockIdx.x * PATCH_COLS;
    if (rg < 4) {   // patch_emb.weight[d][c], d = 4*c4+e, c = rg
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int d = c4 * 4 + e;
            float acc = cw[rg * 4] * Gs[0][d];
            acc += cw[rg * 4 + 1] * Gs[1][d];
            acc += cw[rg * 4 + 2] * Gs[2][d];
            acc += cw[rg * 4 + 3] * Gs[3][d];
            prow[d * 4 + rg] = acc + cb[rg] * PBs[d];
        }
    } else if (rg == 4) {
        *reinterpret_cast<f32x4*>(prow + 512 + c4 * 4) = *reinterpret_cast<const f32x4*>(&PBs[c4 * 4]);
    } else if (rg == 5 && c4 < 20) {   // conv.weight[c][ij] (16) | conv.bias[c] (4): one thread per output, d in order
        const int c = c4 < 16 ? c4 >> 2 : c4 - 16;
        const float* src = c4 < 16 ? Gs[c4 & 3] : PBs;
        float acc = 0.f;
        for (int d = 0; d < D; ++d) acc += pw[d * 4 + c] * src[d];
        prow[640 + c4] = acc;
    }
}

// fused AdamW (torch.optim.AdamW semantics, decoupled weight decay, bias-corrected moments)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float pv = p[i];
    const float gv = g[i];
    pv *= 1.0f - lr * wd;
    const float mv = b1 * m[i] + (1.0f - b1) * gv;
    const float vv = b2 * v[i] + (1.0f - b2) * gv * gv;
    m[i] = mv;
    v[i] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] = pv - (lr / bc1) * mv / denom;
}

// the same update for a whole parameter list in ONE launch: workgroup b handles chunk (b - first[t]) of tensor t,
// t found by a short scan of the (<= 64 entry) table staged in LDS
__global__ __launch_bounds__(256) void adamw_multi_kernel(const t2s_adamw_tensor* __restrict__ tab, int n_tensors, float lr,
                                                          float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    __shared__ unsigned int first[65];
    __shared__ t2s_adamw_tensor ent;
    if (threadIdx.x == 0) {
        unsigned int acc = 0;
        int t = 0;
        for (; t < n_tensors; ++t) {
            const unsigned int nb = (unsigned int)((tab[t].n + 1023) / 1024);
            if (blockIdx.x < acc + nb) break;
            acc += nb;
        }
        first[0] = acc;
        ent = tab[t < n_tensors ? t : n_tensors - 1];
    }
    __syncthreads();
    const size_t base = (size_t)(blockIdx.x - first[0]) * 1024;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const size_t i = base + threadIdx.x + 256 * u;
        if (i >= ent.n) return;
        float pv = ent.param[i];
        const float gv = ent.grad[i];
        pv *= 1.0f - lr * wd;
        const float mv = b1 * ent.exp_avg[i] + (1.0f - b1) * gv;
        const float vv = b2 * ent.exp_avg_sq[i] + (1.0f - b2) * gv * gv;
        ent.exp_avg[i] = mv;
        ent.exp_avg_sq[i] = vv;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        ent.param[i] = pv - (lr / bc1) * mv / denom;
    }
}

__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ gout, float* __restrict__ da, size_t n,
                                                      float sign) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    da[i] = sign * 2.0f * (a[i] - b[i]) * (gout[0] / (float)n);
}

// c = temb (+ text)   (transformer.py:176-178)
__global__ void cond_rows_kernel(float* __restrict__ c, const float* __restrict__ temb, int temb_rows,
                                 const float* __restrict__ text, int S) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * D) return;
    const int s = idx >> 7, d = idx & 127;
    float val = temb[(size_t)(temb_rows == 1 ? 0 : s) * D + d];
    if (text) val += text[idx];
    c[idx] = val;
}

// patchify, row-major output (transformer.py:166-172)
__global__ __launch_bounds__(256) void patchify_rows_kernel(const float* __restrict__ x, float* __restrict__ h, int S,
                                                            const float* __restrict__ cw, const float* __restrict__ cb,
                                                            const float* __restrict__ pw, const float* __restrict__ pb,
                                                            const float* __restrict__ pos) {
    // one thread = 4 features of token n for FOUR series: the weights, bias and position row are loaded once per thread
    // and the four 16-byte stores are independent (one series per thread ran at 2.6 TB/s of pure stores)
    constexpr int SPT = 4;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = gid & 31, rest = gid >> 5;
    const int sg = rest / NTOK, n = rest - sg * NTOK;
    if (sg * SPT >= S) return;
    const int hh = n >> 5, ww = n & 31;
    f32x4 w[4], bias, posr;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int d = c4 * 4 + e;
        w[e] = *reinterpret_cast<const f32x4*>(pw + d * 4);
        bias[e] = pb[d];
        posr[e] = pos[n * D + d];
    }
    float px[SPT][4];
#pragma unroll
    for (int u = 0; u < SPT; ++u) {
        const int s = sg * SPT + u < S ? sg * SPT + u : S - 1;
        const float* xin = x + (size_t)s * LAT;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) px[u][i * 2 + j] = xin[(2 * ww + j) * LATW + 2 * hh + i];
    }
#pragma unroll
    for (int u = 0; u < SPT; ++u) {
        const int s = sg * SPT + u;
        float cv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float acc = cw[c * 4 + 0] * px[u][0];
            acc += cw[c * 4 + 1] * px[u][1];
            acc += cw[c * 4 + 2] * px[u][2];
            acc += cw[c * 4 + 3] * px[u][3];
            cv[c] = acc + cb[c];
        }
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float acc = w[e].x * cv[0];
            acc += w[e].y * cv[1];
            acc += w[e].z * cv[2];
            acc += w[e].w * cv[3];
            o[e] = acc + bias[e] + posr[e];
        }
        if (s < S) reinterpret_cast<f32x4*>(h)[((size_t)s * NTOK + n) * 32 + c4] = o;
    }
}

// gradient of patchify with respect to the LATENT (transformer.py:166-172 backwards): token n = hh*32 + ww owns the 2x2 patch
// lat[2ww+j][2hh+i]; dcv[c] = sum_d dx[tok][d] patch_w[d][c]; dlat[2ww+j][2hh+i] = sum_c conv_w[c][i][j] dcv[c].  Every latent
// element belongs to exactly one token: plain stores.  Needed only when something upstream of the latent trains
// (train.py:31-33 with the LA-VAE encoder un-frozen).
__global__ __launch_bounds__(256) void patchify_input_grad_kernel(const float* __restrict__ dx, float* __restrict__ dlat, int S,
                                                                  const float* __restrict__ cw, const float* __restrict__ pw) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= S * NTOK) return;
    const int s = gid / NTOK, n = gid - s * NTOK;
    const int hh = n >> 5, ww = n & 31;
    const f32x4* row = reinterpret_cast<const f32x4*>(dx + (size_t)gid * D);
    float dcv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int d4 = 0; d4 < D / 4; ++d4) {
        const f32x4 g = row[d4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(pw + (d4 * 4 + e) * 4);
            dcv[0] += g[e] * w.x; dcv[1] += g[e] * w.y; dcv[2] += g[e] * w.z; dcv[3] += g[e] * w.w;
        }
    }
    float* out = dlat + (size_t)s * LAT;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) acc += cw[c * 4 + i * 2 + j] * dcv[c];
            out[(2 * ww + j) * LATW + 2 * hh + i] = acc;
        }
}

// final layer, row-major input (transformer.py:182-191)
// f != NULL (bf16 training): the layer input is x_mid + gate * f of the last block, formed here (gate_res_kernel's expression)
__global__ __launch_bounds__(256) void final_rows_kernel(const float* __restrict__ h, int S, const float* __restrict__ lnw,
                                                         const float* __restrict__ lnb, const float* __restrict__ ow,
                                                         const float* __restrict__ ob, float* __restrict__ out,
                                                         const __bf16* __restrict__ f, const float* __restrict__ mod, int gate_off) {
    // 32 lanes per token row, FOUR rows per group in flight: a row is 3 dependent butterfly reductions (mean, variance,
    // the four output dots) and one row per group left the kernel latency-bound at 2.7 TB/s
    constexpr int RPG = 4;
    const int c4 = threadIdx.x & 31;
    const int grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int M = S * NTOK;
    const f32x4 gam = *reinterpret_cast<const f32x4*>(lnw + c4 * 4), bet = *reinterpret_cast<const f32x4*>(lnb + c4 * 4);
    f32x4 w[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) w[p] = *reinterpret_cast<const f32x4*>(ow + p * D + c4 * 4);
    f32x4 v[RPG];
    int tok[RPG];
#pragma unroll
    for (int u = 0; u < RPG; ++u) {
        tok[u] = grp * RPG + u;
        const int tk = tok[u] < M ? tok[u] : M - 1;
        v[u] = reinterpret_cast<const f32x4*>(h)[(size_t)tk * 32 + c4];
        if (f != nullptr) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(mod + (size_t)(tk / NTOK) * MODROW + gate_off + c4 * 4);
            v[u] = v[u] + g * ld4(f, (size_t)tk * 32 + c4);
        }
    }
#pragma unroll
    for (int u = 0; u < RPG; ++u) {
        float s1 = (v[u].x + v[u].y) + (v[u].z + v[u].w);
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) s1 += __shfl_xor(s1, o, 64);
        const float mean = s1 * (1.0f / 128.0f);
        const f32x4 d = v[u] - mean;
        float s2 = (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o, 64);
        const float rstd = rsqrtf(s2 * (1.0f / 128.0f) + 1e-5f);
        const f32x4 y = (d * rstd) * gam + bet;
        float acc[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float a = (y.x * w[p].x + y.y * w[p].y) + (y.z * w[p].z + y.w * w[p].w);
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) a += __shfl_xor(a, o, 64);
            acc[p] = a + ob[p];
        }
        if (tok[u] < M && c4 < 4) {
            const int s = tok[u] / NTOK, n = tok[u] - s * NTOK;
            const int hh = n >> 5, ww = n & 31;
            out[(size_t)s * LAT + (2 * ww + (c4 & 1)) * LATW + 2 * hh + (c4 >> 1)] = acc[c4];
        }
    }
}

// ---------------------------------------------------------------- workspace management
int ensure_ws(t2s_dit* h, int S) {
    const int dtype = h->train_dtype;
    if (h->train && h->train->cap_seqs >= S && h->train->dtype == dtype) return T2S_OK;
    t2s::train_free(h);
    t2s_train_ws* w = new t2s_train_ws();
    // Grow with 12.5 % headroom (within the handle's capacity): the length groups of a mix-train batch fluctuate by a few
    // per cent from step to step (train.py:60-87), and a workspace rebuilt at every new maximum costs a device-wide
    // hipFree + a multi-GB hipMalloc each time -- seconds in a long-lived process (found in round 4: train.py's loop at
    // 65-180 ms per 28 ms step).  Everything below is sized for `S` = the capacity.
    {
        const int want = (S + S / 8 + 63) / 64 * 64;
        S = want < h->max_seqs ? want : (S > h->max_seqs ? S : h->max_seqs);
    }
    w->cap_seqs = S;
    w->dtype = dtype;
    const bool bf = dtype == T2S_TRAIN_BF16;
    auto fail = [&](const char* what, double mb) {
        set_error("t2s train: hipMalloc(%s, %.1f MB) failed", what, mb);
        if (w->warena) (void)hipFree(w->warena);
        if (w->warena16) (void)hipFree(w->warena16);
        if (w->act) (void)hipFree(w->act);
        if (w->act16) (void)hipFree(w->act16);
        delete w;
        return T2S_E_HIP;
    };
    // ---- packed weights (fp32 packs: fc2 tile-major + all W^T; bf16 packs: W and W^T of every linear)
    size_t woff = 0;
    auto wtake = [&](size_t n) { size_t o = woff; woff += r64(n); return o; };
    size_t o_f[4][NBLK], o_t[4][NBLK];
    const size_t wsz[4] = {3 * D * D, D * D, 2 * D * D, 2 * D * D};   // qkv, proj, fc1, fc2
    for (int i = 0; i < NBLK; ++i)
        for (int j = 0; j < 4; ++j) { o_f[j][i] = wtake(wsz[j]); o_t[j][i] = wtake(wsz[j]); }
    if (!bf) {
        if (hipMalloc(&w->warena, woff * sizeof(float)) != hipSuccess) return fail("packed weights", woff * 4 / 1e6);
        for (int i = 0; i < NBLK; ++i) {
            w->qkv_f[i] = (f32x4*)(w->warena + o_f[0][i]); w->proj_f[i] = (f32x4*)(w->warena + o_f[1][i]);
            w->fc1_f[i] = (f32x4*)(w->warena + o_f[2][i]); w->fc2_f[i] = (f32x4*)(w->warena + o_f[3][i]);
            w->qkv_t[i] = (f32x4*)(w->warena + o_t[0][i]); w->proj_t[i] = (f32x4*)(w->warena + o_t[1][i]);
            w->fc1_t[i] = (f32x4*)(w->warena + o_t[2][i]); w->fc2_t[i] = (f32x4*)(w->warena + o_t[3][i]);
        }
    } else {
        if (hipMalloc(&w->warena16, woff * sizeof(__bf16)) != hipSuccess) return fail("packed bf16 weights", woff * 2 / 1e6);
        for (int i = 0; i < NBLK; ++i) {
            w->qkv_f16[i] = (bf16x8*)(w->warena16 + o_f[0][i]); w->proj_f16[i] = (bf16x8*)(w->warena16 + o_f[1][i]);
            w->fc1_f16[i] = (bf16x8*)(w->warena16 + o_f[2][i]); w->fc2_f16[i] = (bf16x8*)(w->warena16 + o_f[3][i]);
            w->qkv_t16[i] = (bf16x8*)(w->warena16 + o_t[0][i]); w->proj_t16[i] = (bf16x8*)(w->warena16 + o_t[1][i]);
            w->fc1_t16[i] = (bf16x8*)(w->warena16 + o_t[2][i]); w->fc2_t16[i] = (bf16x8*)(w->warena16 + o_t[3][i]);
        }
    }
    // ---- fp32 arena: residual stream and statistics always; branch activations only in fp32 mode
    const size_t M = (size_t)S * NTOK;
    size_t aoff = 0;
    auto atake = [&](size_t n) { size_t o = aoff; aoff += r64(n); return o; };
    size_t o_xin[NBLK + 1], o_xm[NBLK], o_lse[NBLK];
    size_t o_a1[NBLK], o_q[NBLK], o_k[NBLK], o_v[NBLK], o_o[NBLK], o_p[NBLK], o_a2[NBLK], o_u[NBLK], o_f32[NBLK];
    for (int i = 0; i <= NBLK; ++i) o_xin[i] = atake(M * D);
    for (int i = 0; i < NBLK; ++i) {
        o_xm[i] = atake(M * D);
        o_lse[i] = atake(M * NH);
        if (!bf) {
            o_a1[i] = atake(M * D); o_q[i] = atake(M * D); o_k[i] = atake(M * D); o_v[i] = atake(M * D); o_o[i] = atake(M * D);
            o_p[i] = atake(M * D); o_a2[i] = atake(M * D); o_u[i] = atake(M * 2 * D); o_f32[i] = atake(M * D);
        }
    }
    const size_t o_silu = atake((size_t)S * D), o_mod = atake((size_t)S * MODROW), o_c = atake((size_t)S * D),
                 o_lat = atake((size_t)S * LAT), o_dx = atake(M * D), o_dsum = atake(M * NH),
                 o_dmod = atake((size_t)S * MODROW);
    // t1 also serves the adaLN weight gradient as an (S,768) fp32 temporary in both modes
    const size_t o_t1 = atake(bf ? (size_t)S * MODW : M * D);
    size_t o_wgs = 0;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail("device query", 0.0);
        w->n_cu = prop.multiProcessorCount;
        w->wg_scratch_floats = wgrad16_scratch_floats((int)M, w->n_cu);
        int rows, gx;                                            // the adaLN linear: M = S rows, (768,128)
        wgrad16_plan(S, MODW, D, w->n_cu, &rows, &gx);
        const size_t ada = (size_t)gx * (MODW / 128) * (128 * 128 + 128);
        if (ada > w->wg_scratch_floats) w->wg_scratch_floats = ada;
        o_wgs = atake(w->wg_scratch_floats);
    }
    const size_t o_colpart = atake(bf ? M / 32 * 384 : 64);
    size_t o_t2a = 0, o_t2b = 0, o_t3 = 0, o_t4 = 0;
    if (!bf) { o_t2a = atake(M * 2 * D); o_t2b = atake(M * 2 * D); o_t3 = atake(M * 3 * D); o_t4 = atake(M * D); }
    if (hipMalloc(&w->act, aoff * sizeof(float)) != hipSuccess) return fail("activations", aoff * 4 / 1e6);
    float* A = w->act;
    for (int i = 0; i <= NBLK; ++i) w->x_in[i] = A + o_xin[i];
    for (int i = 0; i < NBLK; ++i) {
        w->x_mid[i] = A + o_xm[i];
        w->lse[i] = A + o_lse[i];
        if (!bf) {
            w->a1[i] = A + o_a1[i]; w->q[i] = A + o_q[i]; w->k[i] = A + o_k[i]; w->v[i] = A + o_v[i]; w->o[i] = A + o_o[i];
            w->p[i] = A + o_p[i]; w->a2[i] = A + o_a2[i]; w->u[i] = A + o_u[i]; w->f[i] = A + o_f32[i];
        }
    }
    w->silu_c = A + o_silu; w->mod = A + o_mod; w->c = A + o_c; w->lat = A + o_lat; w->dx = A + o_dx;
    w->dsum = A + o_dsum; w->dmod = A + o_dmod; w->t1 = A + o_t1;
    if (!bf) { w->t2a = A + o_t2a; w->t2b = A + o_t2b; w->t3 = A + o_t3; w->t4 = A + o_t4; }
    w->wg_scratch = A + o_wgs;
    w->colpart = A + o_colpart;
    // ---- bf16 arena
    if (bf) {
        size_t hoff = 0;
        auto htake = [&](size_t n) { size_t o = hoff; hoff += r64(n); return o; };
        size_t h_a1[NBLK], h_q[NBLK], h_k[NBLK], h_v[NBLK], h_o[NBLK], h_p[NBLK], h_a2[NBLK], h_u[NBLK], h_f[NBLK];
        for (int i = 0; i < NBLK; ++i) {
            h_a1[i] = htake(M * D); h_q[i] = htake(M * D); h_k[i] = htake(M * D); h_v[i] = htake(M * D); h_o[i] = htake(M * D);
            h_p[i] = htake(M * D); h_a2[i] = htake(M * D); h_u[i] = htake(M * 2 * D); h_f[i] = htake(M * D);
        }
        const size_t h_t1 = htake(M * D), h_t2 = htake(M * 2 * D), h_t3 = htake(M * 3 * D), h_t4 = htake(M * D);
        if (hipMalloc(&w->act16, hoff * sizeof(__bf16)) != hipSuccess) return fail("bf16 activations", hoff * 2 / 1e6);
        __bf16* H = w->act16;
        for (int i = 0; i < NBLK; ++i) {
            w->a1h[i] = H + h_a1[i]; w->qh[i] = H + h_q[i]; w->kh[i] = H + h_k[i]; w->vh[i] = H + h_v[i]; w->oh[i] = H + h_o[i];
            w->ph[i] = H + h_p[i]; w->a2h[i] = H + h_a2[i]; w->uh[i] = H + h_u[i]; w->fh[i] = H + h_f[i];
        }
        w->t1h = H + h_t1; w->t2h = H + h_t2; w->t3h = H + h_t3; w->t4h = H + h_t4;
    }
    h->train = w;
    return T2S_OK;
}

// the 32 bf16 weight packs of a training forward (4 blocks x 4 matrices x 2 orientations) in one launch: job table
// as a kernel argument, blockIdx.y = job (each single launch costs ~3 us against ~1 us of work)
struct Pack16Job {
    const float* W;
    __bf16* P;
    int N, K, transpose;
};
struct Pack16Table {
    Pack16Job j[8 * NBLK];
};
__global__ void pack16_multi_kernel(const Pack16Table t) {
    const Pack16Job& j = t.j[blockIdx.y];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= j.N * j.K) return;
    const int n = idx / j.K, k = idx - n * j.K;
    const size_t dst = j.transpose ? packed16_index(k, n, j.N) : packed16_index(n, k, j.K);
    j.P[dst] = (__bf16)j.W[idx];
}

template <int K, int N, int PRO, int EPI>
int bgemm(const void* A, const bf16x8* Wp, const float* bias, __bf16* out, int M, int n_out, hipStream_t st,
          const float* mod = nullptr, int shift_off = 0, int scale_off = 0, __bf16* save_A = nullptr,
          const __bf16* aux = nullptr, __bf16* q = nullptr, __bf16* k = nullptr, __bf16* v = nullptr,
          const __bf16* res = nullptr, int gate_off = 0, float* x_out = nullptr) {
    BGemmArgs a{};
    a.A = A; a.Wp = Wp; a.bias = bias; a.out = out; a.M = M; a.N = n_out; a.mod = mod; a.shift_off = shift_off;
    a.scale_off = scale_off; a.save_A = save_A; a.aux = aux; a.q = q; a.k = k; a.v = v;
    a.res = res; a.gate_off = gate_off; a.x_out = x_out;
    return launch_bgemm<K, N, PRO, EPI>(a, st);
}

// data-gradient GEMM (M,K) x W^T -> (M,128) whose epilogue is the LayerNorm + modulate backward (BEPI_LNBWD), then the
// per-sequence column sums of its partials into dmod
template <int K>
int bgemm_lnbwd(t2s_train_ws* ws, const __bf16* dY, const bf16x8* Wt, const float* ln_x, int shift_off, int scale_off,
                const __bf16* f_next, int gate_next_off, __bf16* t_next, int M, hipStream_t st) {
    BGemmArgs a{};
    a.A = dY; a.Wp = Wt; a.M = M; a.N = D; a.mod = ws->mod; a.shift_off = shift_off; a.scale_off = scale_off;
    a.res = f_next; a.gate_off = gate_next_off; a.out = t_next; a.ln_x = ln_x; a.dx = ws->dx; a.colpart = ws->colpart;
    int rc;
    if ((rc = launch_bgemm<K, 128, BPRO_BF16, BEPI_LNBWD>(a, st))) return rc;
    lnbwd_colsum_reduce_kernel<<<M / NTOK, f_next != nullptr ? 384 : 256, 0, st>>>(ws->colpart, ws->dmod, shift_off, scale_off, gate_next_off);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

template <int K, int NT, int PRO, int EPI>
int gemm(const float* A, const f32x4* Wp, const float* bias, float* out, int M, int N, hipStream_t st,
         const float* mod = nullptr, int shift_off = 0, int scale_off = 0, float* save_A = nullptr,
         const float* aux = nullptr, float* q = nullptr, float* k = nullptr, float* v = nullptr) {
    GemmArgs a{};
    a.A = A; a.Wp = Wp; a.bias = bias; a.out = out; a.M = M; a.N = N; a.mod = mod; a.shift_off = shift_off;
    a.scale_off = scale_off; a.save_A = save_A; a.aux = aux; a.q = q; a.k = k; a.v = v;
    return launch_gemm_rows<K, NT, PRO, EPI>(a, st);
}

// weight (and, if db != NULL, bias) gradient of a linear layer
int wgrad(t2s_train_ws* ws, const float* dY, const float* X, float* dW, float* db, int M, int N, int K, hipStream_t st) {
    return launch_wgrad32(dY, X, dW, db, M, N, K, ws->wg_scratch, ws->wg_scratch_floats, ws->n_cu, st);
}

int colsum(const float* Y, float* out, int M, int N, hipStream_t st) {
    const int rows_per_wg = 512;
    colsum_kernel<<<(M + rows_per_wg - 1) / rows_per_wg, 256, 0, st>>>(Y, out, M, N, rows_per_wg);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

}  // namespace

namespace t2s {
void train_free(t2s_dit* h) {
    if (!h || !h->train) return;
    t2s_train_ws* w = h->train;
    if (w->warena) (void)hipFree(w->warena);
    if (w->warena16) (void)hipFree(w->warena16);
    if (w->act) (void)hipFree(w->act);
    if (w->act16) (void)hipFree(w->act16);
    delete w;
    h->train = nullptr;
}
}  // namespace t2s

// ------------------------------------------------------------------ C ABI
extern "C" {

int t2s_dit_set_train_dtype(t2s_dit* h, int dtype) {
    T2S_REQUIRE(h, "t2s_dit_set_train_dtype: NULL handle");
    T2S_REQUIRE(dtype == T2S_TRAIN_F32 || dtype == T2S_TRAIN_BF16, "t2s_dit_set_train_dtype: unknown dtype %d", dtype);
    h->train_dtype = dtype;
    return T2S_OK;
}

int t2s_dit_train_forward(t2s_dit* h, const t2s_dit_weights* w, const float* x, const float* temb, int temb_rows,
                          const float* text, float* out, int B, void* stream) {
    t2s::reset_tile_dir();
    T2S_REQUIRE(h && w && x && temb && out, "t2s_dit_train_forward: NULL argument");
    T2S_REQUIRE(B > 0 && B <= h->max_seqs, "t2s_dit_train_forward: B=%d exceeds max_seqs=%d", B, h->max_seqs);
    T2S_REQUIRE(temb_rows == 1 || temb_rows == B, "t2s_dit_train_forward: temb_rows=%d", temb_rows);
    hipStream_t st = (hipStream_t)stream;
    int rc = ensure_ws(h, B);
    if (rc != T2S_OK) return rc;
    t2s_train_ws* ws = h->train;
    const int S = B, M = S * NTOK;
    ws->S = S;
    ws->has_text = text != nullptr;
    // the handle's own parameter copies / forward packs must be current: the caller refreshes them
    // with t2s_dit_update_weights; here only the training-specific packs
    const bool bf = ws->dtype == T2S_TRAIN_BF16;
    for (int i = 0; i < NBLK; ++i) {
        const t2s_dit_block_weights& b = w->blk[i];
        if (!bf) {   // fc2 tile-major, all W^T
            if ((rc = pack_any(b.qkv_w, ws->qkv_t[i], 3 * D, D, 1, st)) || (rc = pack_any(b.proj_w, ws->proj_t[i], D, D, 1, st)) ||
                (rc = pack_any(b.fc1_w, ws->fc1_t[i], 2 * D, D, 1, st)) || (rc = pack_any(b.fc2_w, ws->fc2_f[i], D, 2 * D, 0, st)) ||
                (rc = pack_any(b.fc2_w, ws->fc2_t[i], D, 2 * D, 1, st)))
                return rc;
        }
    }
    if (bf) {        // bf16 copies of the fp32 master weights, both orientations
        Pack16Table pt{};
        int np = 0;
        auto pk = [&](const float* W, bf16x8* plain, bf16x8* transposed, int N, int K) {
            pt.j[np++] = Pack16Job{W, reinterpret_cast<__bf16*>(plain), N, K, 0};
            pt.j[np++] = Pack16Job{W, reinterpret_cast<__bf16*>(transposed), N, K, 1};
        };
        for (int i = 0; i < NBLK; ++i) {
            const t2s_dit_block_weights& b = w->blk[i];
            pk(b.qkv_w, ws->qkv_f16[i], ws->qkv_t16[i], 3 * D, D);
            pk(b.proj_w, ws->proj_f16[i], ws->proj_t16[i], D, D);
            pk(b.fc1_w, ws->fc1_f16[i], ws->fc1_t16[i], 2 * D, D);
            pk(b.fc2_w, ws->fc2_f16[i], ws->fc2_t16[i], D, 2 * D);
        }
        { TimeScope ts(h, TC_TR_TAIL, st);
        pack16_multi_kernel<<<dim3((3 * D * D + 255) / 256, np), 256, 0, st>>>(pt);
        T2S_LAUNCH_CHECK();
        }
    }
    T2S_HIP_CHECK(hipMemcpyAsync(ws->lat, x, (size_t)S * LAT * sizeof(float), hipMemcpyDeviceToDevice, st));
    { TimeScope ts(h, TC_TR_TAIL, st);
    cond_rows_kernel<<<(S * D + 255) / 256, 256, 0, st>>>(ws->c, temb, temb_rows, text, S);
    T2S_LAUNCH_CHECK();
    }
    { TimeScope ts(h, TC_TR_TAIL, st);
    silu_kernel<<<(S * D + 255) / 256, 256, 0, st>>>(ws->c, ws->silu_c, S * D);
    T2S_LAUNCH_CHECK();
    }
    { TimeScope ts(h, TC_TR_TAIL, st);
    if ((rc = gemm<128, 3, PRO_PLAIN, EPI_BIAS>(ws->silu_c, h->ada_p, h->ada_b, ws->mod, S, MODROW, st))) return rc;
    }
    { TimeScope ts(h, TC_TR_TAIL, st);
    patchify_rows_kernel<<<((S + 3) / 4 * NTOK * 32 + 255) / 256, 256, 0, st>>>(x, ws->x_in[0], S, h->conv_w, h->conv_b, h->patch_w,
                                                                      h->patch_b, h->pos);
    T2S_LAUNCH_CHECK();
    }
    for (int i = 0; i < NBLK && bf; ++i) {
        const int base = i * MODW;
        // The gate / residual adds have no kernel of their own: the GEMM that consumes a residual-stream tensor forms it
        // in its prologue (x = x_prev + gate * branch), writes it out for the backward pass and LayerNorm-modulates it.
        // a1 = mod(LN1(x_in)) -> bf16; q,k,v = a1 Wqkv^T + b -> bf16 heads; for i > 0, x_in[i] = x_mid[i-1] + g2 f[i-1]
        { TimeScope ts(h, TC_TR_GEMM, st);
        if (i == 0)
            rc = bgemm<128, 384, BPRO_LN, BEPI_QKV>(ws->x_in[i], ws->qkv_f16[i], h->qkv_b[i], nullptr, M, 3 * D, st, ws->mod,
                                                   base + 0 * D, base + 1 * D, ws->a1h[i], nullptr, ws->qh[i], ws->kh[i], ws->vh[i]);
        else
            rc = bgemm<128, 384, BPRO_LN_RES, BEPI_QKV>(ws->x_mid[i - 1], ws->qkv_f16[i], h->qkv_b[i], nullptr, M, 3 * D, st, ws->mod,
                                                       base + 0 * D, base + 1 * D, ws->a1h[i], nullptr, ws->qh[i], ws->kh[i], ws->vh[i],
                                                       ws->fh[i - 1], (i - 1) * MODW + 5 * D, ws->x_in[i]);
        if (rc) return rc;
        }
        { TimeScope ts(h, TC_TR_ATTN_FWD, st);
        if ((rc = attn16_train_fwd(ws->qh[i], ws->kh[i], ws->vh[i], ws->oh[i], ws->lse[i], S * NH, st))) return rc;
        }
        // p = o Wp^T + b
        { TimeScope ts(h, TC_TR_GEMM, st);
        if ((rc = bgemm<128, 128, BPRO_BF16, BEPI_BF16>(ws->oh[i], ws->proj_f16[i], h->proj_b[i], ws->ph[i], M, D, st))) return rc;
        }
        // x_mid = x_in + g1 * p; a2 = mod(LN2(x_mid)); u = a2 W1^T + b1
        { TimeScope ts(h, TC_TR_GEMM, st);
        if ((rc = bgemm<128, 256, BPRO_LN_RES, BEPI_BF16>(ws->x_in[i], ws->fc1_f16[i], h->fc1_b[i], ws->uh[i], M, 2 * D, st, ws->mod,
                                                      base + 3 * D, base + 4 * D, ws->a2h[i], nullptr, nullptr, nullptr, nullptr,
                                                      ws->ph[i], base + 2 * D, ws->x_mid[i])))
            return rc;
        }
        // f = gelu(u) W2^T + b2   (gelu(u) is not saved: the fc2 weight gradient re-applies it to u)
        { TimeScope ts(h, TC_TR_GEMM, st);
        if ((rc = bgemm<256, 128, BPRO_GELU, BEPI_BF16>(ws->uh[i], ws->fc2_f16[i], h->fc2_b[i], ws->fh[i], M, D, st))) return rc;
        }
        // (the last block's x_mid + g2 * f is formed inside the final-layer kernels, forward and backward: never written)
    }
    for (int i = 0; i < NBLK && !bf; ++i) {
        const int base = i * MODW;
        // a1 = mod(LN1(x_in)); q,k,v = a1 Wqkv^T + b
        { TimeScope ts(h, TC_TR_GEMM, st);
        if ((rc = gemm<128, 3, PRO_LNMOD, EPI_QKV>(ws->x_in[i], h->qkv_p[i], h->qkv_b[i], nullptr, M, 3 * D, st, ws->mod,
                                                    base + 0 * D, base + 1 * D, ws->a1[i], nullptr, ws->q[i], ws->k[i], ws->v[i])))
            return rc;
        }
        { TimeScope ts(h, TC_TR_ATTN_FWD, st);
        if ((rc = attn_plain_train_fwd(ws->q[i], ws->k[i], ws->v[i], ws->o[i], ws->lse[i], S * NH, st))) return rc;
        }
        { TimeScope ts(h, TC_TR_GEMM, st);
        if ((rc = gemm<128, 1, PRO_PLAIN, EPI_BIAS>(ws->o[i], h->proj_p[i], h->proj_b[i], ws->p[i], M, D, st))) return rc;
        }
        { TimeScope ts(h, TC_TR_ELEM, st);
        gate_res_kernel<<<(M * 32 + 255) / 256, 256, 0, st>>>(ws->x_in[i], ws->p[i], ws->mod, base + 2 * D, ws->x_mid[i], M);
        T2S_LAUNCH_CHECK();
        }
        // a2 = mod(LN2(x_mid)); u = a2 W1^T + b1; f = gelu(u) W2^T + b2
        { TimeScope ts(h, TC_TR_GEMM, st);
        if ((rc = gemm<128, 2, PRO_LNMOD, EPI_BIAS>(ws->x_mid[i], h->fc1_p[i], h->fc1_b[i], ws->u[i], M, 2 * D, st, ws->mod,
                                                     base + 3 * D, base + 4 * D, ws->a2[i])))
            return rc;
        }
        { TimeScope ts(h, TC_TR_GEMM, st);
        if ((rc = gemm<256, 1, PRO_GELU, EPI_BIAS>(ws->u[i], ws->fc2_f[i], h->fc2_b[i], ws->f[i], M, D, st))) return rc;
        }
        { TimeScope ts(h, TC_TR_ELEM, st);
        gate_res_kernel<<<(M * 32 + 255) / 256, 256, 0, st>>>(ws->x_mid[i], ws->f[i], ws->mod, base + 5 * D, ws->x_in[i + 1], M);
        T2S_LAUNCH_CHECK();
        }
    }
    { TimeScope ts(h, TC_TR_TAIL, st);
    if (bf)
        final_rows_kernel<<<((M + 3) / 4 * 32 + 255) / 256, 256, 0, st>>>(ws->x_mid[NBLK - 1], S, h->ln_w, h->ln_b, h->out_w, h->out_b, out,
                                                                           ws->fh[NBLK - 1], ws->mod, (NBLK - 1) * MODW + 5 * D);
    else
        final_rows_kernel<<<((M + 3) / 4 * 32 + 255) / 256, 256, 0, st>>>(ws->x_in[NBLK], S, h->ln_w, h->ln_b, h->out_w, h->out_b, out,
                                                                           nullptr, nullptr, 0);
    T2S_LAUNCH_CHECK();
    }
    return T2S_OK;
}

int t2s_dit_train_backward(t2s_dit* h, const float* dout, const t2s_dit_grads* g, int B, void* stream) {
    t2s::reset_tile_dir();
    T2S_REQUIRE(h && dout && g, "t2s_dit_train_backward: NULL argument");
    T2S_REQUIRE(h->train && h->train->S == B, "t2s_dit_train_backward: no matching t2s_dit_train_forward (B=%d)", B);
    T2S_REQUIRE(h->train->dtype == h->train_dtype, "t2s_dit_train_backward: training dtype changed since the forward");
    hipStream_t st = (hipStream_t)stream;
    t2s_train_ws* ws = h->train;
    const int S = B, M = S * NTOK;
    int rc;
    // ---- zero every gradient (the kernels accumulate with atomics)
    struct Z { float* p; size_t n; };
    std::vector<Z> zs = {{g->conv_w, 16}, {g->conv_b, 4}, {g->patch_w, 512}, {g->patch_b, 128}, {g->ln_w, 128},
                         {g->ln_b, 128}, {g->out_w, 512}, {g->out_b, 4}};
    for (int i = 0; i < NBLK; ++i) {
        const t2s_dit_block_grads& b = g->blk[i];
        zs.insert(zs.end(), {{b.qkv_w, 3 * D * D}, {b.qkv_b, 3 * D}, {b.proj_w, D * D}, {b.proj_b, D}, {b.fc1_w, 2 * D * D},
                             {b.fc1_b, 2 * D}, {b.fc2_w, 2 * D * D}, {b.fc2_b, D}, {b.ada_w, MODW * D}, {b.ada_b, MODW}});
    }
    // adjacent ranges are merged: the host mirror carves all of them out of one flat bucket (what the data-parallel
    // all-reduce sends), so this is normally ONE memset instead of 48
    for (size_t i = 0; i < zs.size();) {
        T2S_REQUIRE(zs[i].p, "t2s_dit_train_backward: NULL gradient pointer");
        float* p = zs[i].p;
        size_t n = zs[i].n;
        size_t k = i + 1;
        while (k < zs.size() && zs[k].p == p + n) n += zs[k++].n;
        T2S_HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(float), st));
        i = k;
    }
    // ---- final layer
    const bool bf = ws->dtype == T2S_TRAIN_BF16;
    const int tail_wgs = (M + TAIL_ROWS - 1) / TAIL_ROWS;
    constexpr int FUSED_ROWS = NTOK / 3;                       // 160: three workgroups per sequence
    const int final_wgs = bf ? M / FUSED_ROWS : tail_wgs, final_stride = bf ? FINAL_COLS + D : FINAL_COLS;
    T2S_REQUIRE((size_t)final_wgs * final_stride <= ws->wg_scratch_floats && (size_t)tail_wgs * PATCH_COLS <= ws->wg_scratch_floats,
                "t2s_dit_train_backward: scratch too small for the tail partials");
    { TimeScope ts(h, TC_TR_TAIL, st);
    if (bf) {   // + the gate backward of the last block's MLP branch: t1 = df = g2 * dx, dgate2
        const int goff = (NBLK - 1) * MODW + 5 * D;
        final_bwd_kernel<true, FUSED_ROWS><<<final_wgs, 256, 0, st>>>(ws->x_mid[NBLK - 1], dout, h->ln_w, h->ln_b, h->out_w, ws->dx, ws->wg_scratch, M,
                                                                      ws->fh[NBLK - 1], ws->mod, goff, ws->t1h);
        T2S_LAUNCH_CHECK();
        final_gate_reduce_kernel<<<S, D, 0, st>>>(ws->wg_scratch, final_stride, ws->dmod, goff);
    } else {
        final_bwd_kernel<false, TAIL_ROWS><<<final_wgs, 256, 0, st>>>(ws->x_in[NBLK], dout, h->ln_w, h->ln_b, h->out_w, ws->dx, ws->wg_scratch, M,
                                                                      nullptr, nullptr, 0, nullptr);
    }
    T2S_LAUNCH_CHECK();
    tail_reduce_kernel<<<(FINAL_COLS + 31) / 32, 256, 0, st>>>(ws->wg_scratch, final_wgs, FINAL_COLS, final_stride,
                                                                 TailDst{{g->ln_w, g->ln_b, g->out_w, g->out_b}, {D, D, 4 * D, 4}});
    T2S_LAUNCH_CHECK();
    }
    // (Measured and dropped in round 2: the four weight gradients of a block on a side stream.  Issued as soon as their
    // operands exist they only share the HBM bandwidth with the GEMM / LayerNorm kernels they run beside (13.42 ms per step,
    // the same as in order); issued beside the VALU-bound attention backward they slow it by more than they take alone
    // (attention backward 3.0 -> 4.4 ms per step, 14.1 ms per step): the attention workgroups fill the CUs.
    // Round 3 repeated both as timing-only builds on a stream CALIBRATED to run concurrently (t2s_sampler.hip lane_streams:
    // HIP streams that share a hardware queue serialise, which may have been the "same as in order" of round 2): without any
    // dependency waits 10.92 vs 11.10 ms in order, issued beside the attention backward 11.08 -- the attention backward then
    // takes 3.1-3.7 instead of 2.2 ms: it is bound by vector issue, but its 16 waves per CU leave the other kernel no slot.)
    for (int i = NBLK - 1; i >= 0 && bf; --i) {
        const int base = i * MODW;
        const t2s_dit_block_grads& b = g->blk[i];
        // ---- MLP branch: x_out = x_mid + g2 * f.  t1 = df = g2 * dx and dgate2 come from the kernel that produced dx: the
        // next block's LN1 backward epilogue, or the final-layer backward for the last block
        { TimeScope ts(h, TC_TR_WGRAD, st);     // dW2 = df^T gelu(u): gelu applied to the fetched u chunks (not saved)
        if ((rc = launch_wgrad16<true>(ws->t1h, ws->uh[i], b.fc2_w, b.fc2_b, M, D, 2 * D, ws->wg_scratch, ws->wg_scratch_floats, ws->n_cu, st))) return rc;
        }
        // du = (df W2) * gelu'(u)
        { TimeScope ts(h, TC_TR_GEMM, st);
        if ((rc = bgemm<128, 256, BPRO_BF16, BEPI_GELUBWD>(ws->t1h, ws->fc2_t16[i], nullptr, ws->t2h, M, 2 * D, st, nullptr, 0, 0,
                                                            nullptr, ws->uh[i])))
            return rc;
        }
        { TimeScope ts(h, TC_TR_WGRAD, st);
        if ((rc = launch_wgrad16(ws->t2h, ws->a2h[i], b.fc1_w, b.fc1_b, M, 2 * D, D, ws->wg_scratch, ws->wg_scratch_floats, ws->n_cu, st))) return rc;
        }
        // da2 = du W1, consumed in the accumulators: LN2 backward into dx merged with the attention branch's gate
        // backward (t1 = dp = g1 * dx, dgate1) -- the epilogue of the data-gradient GEMM (BEPI_LNBWD; da2 never reaches HBM)
        { TimeScope ts(h, TC_TR_GEMM, st);
        if ((rc = bgemm_lnbwd<256>(ws, ws->t2h, ws->fc1_t16[i], ws->x_mid[i], base + 3 * D, base + 4 * D, ws->ph[i], base + 2 * D, ws->t1h, M, st)))
            return rc;
        }
        // ---- attention branch: x_mid = x_in + g1 * p
        { TimeScope ts(h, TC_TR_WGRAD, st);
        if ((rc = launch_wgrad16(ws->t1h, ws->oh[i], b.proj_w, b.proj_b, M, D, D, ws->wg_scratch, ws->wg_scratch_floats, ws->n_cu, st))) return rc;
        }
        { TimeScope ts(h, TC_TR_GEMM, st);
        if ((rc = bgemm<128, 128, BPRO_BF16, BEPI_BF16>(ws->t1h, ws->proj_t16[i], nullptr, ws->t4h, M, D, st))) return rc;   // do
        }
        { TimeScope ts(h, TC_TR_ATTN_BWD, st);
        if ((rc = attn16_bwd(ws->qh[i], ws->kh[i], ws->vh[i], ws->oh[i], ws->t4h, ws->lse[i], ws->dsum, ws->t3h, S * NH, st)))
            return rc;
        }
        { TimeScope ts(h, TC_TR_WGRAD, st);
        if ((rc = launch_wgrad16(ws->t3h, ws->a1h[i], b.qkv_w, b.qkv_b, M, 3 * D, D, ws->wg_scratch, ws->wg_scratch_floats, ws->n_cu, st))) return rc;
        }
        // da1 = dqkv Wqkv with LN1 backward into dx as its epilogue, merged with the gate backward of block i-1's MLP branch
        { TimeScope ts(h, TC_TR_GEMM, st);
        if ((rc = bgemm_lnbwd<384>(ws, ws->t3h, ws->qkv_t16[i], ws->x_in[i], base + 0 * D, base + 1 * D,
                                   i > 0 ? ws->fh[i - 1] : (const __bf16*)nullptr, (i - 1) * MODW + 5 * D, ws->t1h, M, st)))
            return rc;
        }
    }
    for (int i = NBLK - 1; i >= 0 && !bf; --i) {
        const int base = i * MODW;
        const t2s_dit_block_grads& b = g->blk[i];
        // ---- MLP branch: x_out = x_mid + g2 * f
        { TimeScope ts(h, TC_TR_ELEM, st);
        gate_bwd_kernel<<<S, 256, 0, st>>>(ws->dx, ws->f[i], ws->mod, base + 5 * D, ws->t1, ws->dmod);      // t1 = df
        T2S_LAUNCH_CHECK();
        }
        { TimeScope ts(h, TC_TR_ELEM, st);
        gelu_kernel<<<((size_t)M * 64 + 255) / 256, 256, 0, st>>>(ws->u[i], ws->t2a, (size_t)M * 64);        // t2a = gelu(u)
        T2S_LAUNCH_CHECK();
        }
        { TimeScope ts(h, TC_TR_WGRAD, st);
        if ((rc = wgrad(ws, ws->t1, ws->t2a, b.fc2_w, b.fc2_b, M, D, 2 * D, st))) return rc;
        }
        // du = (df W2) * gelu'(u)      (dgrad = row GEMM on W2^T: out 256 <- in 128)
        { TimeScope ts(h, TC_TR_GEMM, st);
        if ((rc = gemm<128, 2, PRO_PLAIN, EPI_GELUBWD>(ws->t1, ws->fc2_t[i], nullptr, ws->t2b, M, 2 * D, st, nullptr, 0, 0,
                                                        nullptr, ws->u[i])))
            return rc;
        }
        { TimeScope ts(h, TC_TR_WGRAD, st);
        if ((rc = wgrad(ws, ws->t2b, ws->a2[i], b.fc1_w, b.fc1_b, M, 2 * D, D, st))) return rc;
        }
        // da2 = du W1 (out 128 <- in 256)
        { TimeScope ts(h, TC_TR_GEMM, st);
        if ((rc = gemm<256, 1, PRO_PLAIN, EPI_BIAS>(ws->t2b, ws->fc1_t[i], nullptr, ws->t1, M, D, st))) return rc;
        }
        { TimeScope ts(h, TC_TR_ELEM, st);
        ln_mod_bwd_kernel<<<S, 256, 0, st>>>(ws->t1, ws->x_mid[i], ws->mod, base + 3 * D, base + 4 * D, ws->dx, ws->dmod,
                                             (const float*)nullptr, 0, (float*)nullptr);
        T2S_LAUNCH_CHECK();
        }
        // ---- attention branch: x_mid = x_in + g1 * p
        { TimeScope ts(h, TC_TR_ELEM, st);
        gate_bwd_kernel<<<S, 256, 0, st>>>(ws->dx, ws->p[i], ws->mod, base + 2 * D, ws->t1, ws->dmod);       // t1 = dp
        T2S_LAUNCH_CHECK();
        }
        { TimeScope ts(h, TC_TR_WGRAD, st);
        if ((rc = wgrad(ws, ws->t1, ws->o[i], b.proj_w, b.proj_b, M, D, D, st))) return rc;
        }
        { TimeScope ts(h, TC_TR_GEMM, st);
        if ((rc = gemm<128, 1, PRO_PLAIN, EPI_BIAS>(ws->t1, ws->proj_t[i], nullptr, ws->t4, M, D, st))) return rc;   // do
        }
        { TimeScope ts(h, TC_TR_ATTN_BWD, st);
        if ((rc = attn_bwd(ws->q[i], ws->k[i], ws->v[i], ws->o[i], ws->t4, ws->lse[i], ws->dsum, ws->t3, S * NH, st))) return rc;
        }
        { TimeScope ts(h, TC_TR_WGRAD, st);
        if ((rc = wgrad(ws, ws->t3, ws->a1[i], b.qkv_w, b.qkv_b, M, 3 * D, D, st))) return rc;
        }
        // da1 = dqkv Wqkv (out 128 <- in 384)
        { TimeScope ts(h, TC_TR_GEMM, st);
        if ((rc = gemm<384, 1, PRO_PLAIN, EPI_BIAS>(ws->t3, ws->qkv_t[i], nullptr, ws->t1, M, D, st))) return rc;
        }
        { TimeScope ts(h, TC_TR_ELEM, st);
        ln_mod_bwd_kernel<<<S, 256, 0, st>>>(ws->t1, ws->x_in[i], ws->mod, base + 0 * D, base + 1 * D, ws->dx, ws->dmod,
                                             (const float*)nullptr, 0, (float*)nullptr);
        T2S_LAUNCH_CHECK();
        }
    }
    // ---- patchify
    { TimeScope ts(h, TC_TR_TAIL, st);
    patchify_bwd_kernel<<<tail_wgs, 256, 0, st>>>(ws->dx, ws->lat, S, h->conv_w, h->conv_b, h->patch_w, ws->wg_scratch, M);
    T2S_LAUNCH_CHECK();
    tail_reduce_kernel<<<(PATCH_COLS + 31) / 32, 256, 0, st>>>(ws->wg_scratch, tail_wgs, PATCH_COLS, PATCH_COLS,
                                                                 TailDst{{g->patch_w, g->patch_b, g->conv_w, g->conv_b}, {4 * D, D, 16, 4}});
    T2S_LAUNCH_CHECK();
    }
    // ---- adaLN linear: mod = silu(c) W_ada^T + b_ada  (per block rows [768 i, 768 i + 768) of the (3072,128) stack)
    for (int i = 0; i < NBLK; ++i) {
        // dmod block slice is strided (row stride MODROW): copy to a dense (S,768) temp first
        T2S_HIP_CHECK(hipMemcpy2DAsync(ws->t1, MODW * sizeof(float), ws->dmod + i * MODW, MODROW * sizeof(float),
                                       MODW * sizeof(float), S, hipMemcpyDeviceToDevice, st));
        { TimeScope ts(h, TC_TR_WGRAD, st);
        if ((rc = wgrad(ws, ws->t1, ws->silu_c, g->blk[i].ada_w, g->blk[i].ada_b, S, MODW, D, st))) return rc;
        }
    }
    return T2S_OK;
}

int t2s_dit_train_input_grad(t2s_dit* h, float* dinput, int B, void* stream) {
    T2S_REQUIRE(h && dinput, "t2s_dit_train_input_grad: NULL argument");
    T2S_REQUIRE(h->train && h->train->S == B, "t2s_dit_train_input_grad: no matching t2s_dit_train_backward (B=%d)", B);
    t2s_train_ws* ws = h->train;
    TimeScope ts(h, TC_TR_TAIL, (hipStream_t)stream);
    patchify_input_grad_kernel<<<(B * NTOK + 255) / 256, 256, 0, (hipStream_t)stream>>>(ws->dx, dinput, B, h->conv_w, h->patch_w);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

int t2s_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, void* stream) {
    T2S_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step > 0, "t2s_adamw_step: bad argument");
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2 = 1.0f - powf(beta2, (float)step);
    adamw_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, (size_t)n, lr,
                                                                                beta1, beta2, eps, weight_decay, bc1,
                                                                                sqrtf(bc2));
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

int t2s_adamw_step_multi(const t2s_adamw_tensor* table_dev, int n_tensors, uint64_t total_chunks, float lr, float beta1,
                         float beta2, float eps, float weight_decay, int step, void* stream) {
    T2S_REQUIRE(table_dev && n_tensors > 0 && n_tensors <= 64 && total_chunks > 0 && step > 0, "t2s_adamw_step_multi: bad argument");
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2 = 1.0f - powf(beta2, (float)step);
    adamw_multi_kernel<<<(unsigned)total_chunks, 256, 0, (hipStream_t)stream>>>(table_dev, n_tensors, lr, beta1, beta2, eps,
                                                                                weight_decay, bc1, sqrtf(bc2));
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

int t2s_mse_backward(const float* a, const float* b, const float* grad_out, float* da, float* db, uint64_t n, void* stream) {
    T2S_REQUIRE(a && b && grad_out && n > 0 && (da || db), "t2s_mse_backward: bad argument");
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (da) mse_bwd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(a, b, grad_out, da, (size_t)n, 1.0f);
    if (db) mse_bwd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(a, b, grad_out, db, (size_t)n, -1.0f);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

}  // extern "C"
