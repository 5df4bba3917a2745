// LA-VAE codec (reference model/pretrained/vqvae.py:36-105) for MI355X.
//
// The codec runs once per batch (15-17 MFLOP per series against 1.95 TFLOP for the
// 1000-step loop), so it is a single launch with ONE workgroup per series and time tile: the
// activation stack of a tile (<= 256 channels x <= 32 positions at the L/4 resolution) stays in
// LDS from the latent to the decoded samples, weights (2.7 MB fp32) stream from L2.  Exact fp32 VALU.
// L <= 128 (BASELINE configs: 24 / 48 / 96) is one tile = the whole series.  Longer series (vqvae.py accepts any L;
// the reference's SUSHI set is 2048 long) are cut into tiles of 32 - 2 H core positions with a halo of H = the
// convolutions' receptive radius on either side, recomputed per tile: every kept value sums the same terms in the
// same order as the untiled computation, so the tiling is invisible in the results.
#include "t2s_wgrad.h"

namespace t2s {

constexpr int VAE_TMAX = 32;   // positions after the stride-4 stem: L/4 <= 32  (L <= 128)
constexpr int VAE_CMAX = 256;  // res_hidden
constexpr int VAE_THREADS = 256;
#ifndef T2S_VAE_CP
#define T2S_VAE_CP 8
#endif
constexpr int VAE_CO_PER_THREAD = T2S_VAE_CP;   // output channels per thread in conv1d_lds

struct VaeDev {  // device copies in the reference layouts
    int hidden, res_hidden, n_res, emb;
    const float *dec_conv1_w, *dec_conv1_b, *dec_ct1_w, *dec_ct1_b, *dec_ct2_w, *dec_ct2_b;
    const float *dec_c3[4], *dec_c1[4];
    const float *enc_conv1_w, *enc_conv1_b, *enc_conv2_w, *enc_conv2_b, *enc_conv3_w, *enc_conv3_b;
    const float *enc_c3[4], *enc_c1[4];
    const float *enc_prevq_w, *enc_prevq_b;
};

// out[co][t] (+)= b[co] + sum_{ci,kk} W[co][ci][kk] * in[ci][t*STRIDE + kk - pad]   (Conv1d)
// Buffers are LDS, row stride `ld`.  RELU_OUT applies to the final value; ACCUM adds into out.
// Windows: `in` holds global positions [in_g0, in_g0 + Tin), `out` global positions [out_g0, out_g0 + Tout); both are
// clipped to the series, so "outside the input window" is either true zero padding or a position whose influence
// stays inside the discarded halo.
template <int KS, int STRIDE, bool RELU_OUT, bool ACCUM>
__device__ void conv1d_lds(const float* in, int Cin, int Tin, float* out, int Cout, int Tout,
                           const float* __restrict__ W, const float* __restrict__ bias, int pad,
                           int ld_in, int ld_out, int in_g0 = 0, int out_g0 = 0) {
    // VAE_CO_PER_THREAD (8; measured 2: 21.3, 4: 19.1, 8: 17.9, 16: 17.8 ms per uncached bf16 train step) output channels per thread: the LDS activations are read once for four FMAs and four independent
    // accumulation chains are in flight (one chain per thread was latency-bound at 1.6 TFLOP/s); each output
    // still sums its (ci, kk) terms in the same order.  Cout is a multiple of 4 for every LA-VAE layer but the last.
    constexpr int CP = VAE_CO_PER_THREAD;
    if (Cout % CP == 0) {
        for (int o = threadIdx.x; o < (Cout / CP) * Tout; o += VAE_THREADS) {
            const int co = (o / Tout) * CP, t = o - (o / Tout) * Tout;
            float acc[CP];
#pragma unroll
            for (int u = 0; u < CP; ++u) acc[u] = bias ? bias[co + u] : 0.f;
            const float* w = W + (size_t)co * Cin * KS;
            const size_t ws = (size_t)Cin * KS;
#pragma unroll 2
            for (int ci = 0; ci < Cin; ++ci) {
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    const int ti = (out_g0 + t) * STRIDE + kk - pad - in_g0;
                    if (ti >= 0 && ti < Tin) {
                        const float a = in[ci * ld_in + ti];
#pragma unroll
                        for (int u = 0; u < CP; ++u) acc[u] += w[u * ws + ci * KS + kk] * a;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < CP; ++u) {
                float r = acc[u];
                if (ACCUM) r += out[(co + u) * ld_out + t];
                if (RELU_OUT) r = fmaxf(r, 0.f);
                out[(co + u) * ld_out + t] = r;
            }
        }
        return;
    }
    for (int o = threadIdx.x; o < Cout * Tout; o += VAE_THREADS) {
        const int co = o / Tout, t = o - co * Tout;
        float acc = bias ? bias[co] : 0.f;
        const float* w = W + (size_t)co * Cin * KS;
        for (int ci = 0; ci < Cin; ++ci) {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int ti = (out_g0 + t) * STRIDE + kk - pad - in_g0;
                if (ti >= 0 && ti < Tin) acc += w[ci * KS + kk] * in[ci * ld_in + ti];
            }
        }
        if (ACCUM) acc += out[co * ld_out + t];
        if (RELU_OUT) acc = fmaxf(acc, 0.f);
        out[co * ld_out + t] = acc;
    }
}

// ConvTranspose1d(k=4, stride=2, padding=1): W is (Cin, Cout, 4);
// out[co][t] = b[co] + sum_{ci,kk : t = 2 i - 1 + kk} in[ci][i] * W[ci][co][kk]
template <bool RELU_OUT>
__device__ void convT1d_k4s2_lds(const float* in, int Cin, int Tin, float* out, int Cout,
                                 const float* __restrict__ W, const float* __restrict__ bias,
                                 int ld_in, int ld_out) {
    const int Tout = 2 * Tin;
    for (int o = threadIdx.x; o < Cout * Tout; o += VAE_THREADS) {
        const int co = o / Tout, t = o - co * Tout;
        float acc = bias[co];
        // t+1-kk even  ->  kk has the parity of t+1
        const int k0 = (t + 1) & 1;
        for (int ci = 0; ci < Cin; ++ci) {
#pragma unroll
            for (int kk2 = 0; kk2 < 2; ++kk2) {
                const int kk = k0 + 2 * kk2;
                const int i = (t + 1 - kk) >> 1;
                if (t + 1 - kk >= 0 && i < Tin) acc += in[ci * ld_in + i] * W[((size_t)ci * Cout + co) * 4 + kk];
            }
        }
        if (RELU_OUT) acc = fmaxf(acc, 0.f);
        out[co * ld_out + t] = acc;
    }
}

// F.interpolate(mode='linear', align_corners=True) along the last axis, [C][Tin] -> [C][Tout]
// (`out` holds the global output positions [out_g0, out_g0 + Tw) of Tout)
__device__ void interp_linear_ac(const float* in, int C, int Tin, int ld_in, float* out, int Tout,
                                 int ld_out, int out_g0 = 0, int Tw = -1) {
    const float scale = Tout > 1 ? (float)(Tin - 1) / (float)(Tout - 1) : 0.f;
    if (Tw < 0) Tw = Tout;
    for (int o = threadIdx.x; o < C * Tw; o += VAE_THREADS) {
        const int c = o / Tw, t = o - c * Tw;
        const float real = scale * (float)(out_g0 + t);
        const int i0 = (int)real;
        const int i1 = i0 + (i0 < Tin - 1 ? 1 : 0);
        const float l1 = real - (float)i0;
        const float l0 = 1.0f - l1;
        out[c * ld_out + t] = l0 * in[c * ld_in + i0] + l1 * in[c * ld_in + i1];
    }
}

__device__ void relu_inplace(float* x, int C, int T, int ld) {
    for (int o = threadIdx.x; o < C * T; o += VAE_THREADS) {
        const int c = o / T, t = o - c * T;
        x[c * ld + t] = fmaxf(x[c * ld + t], 0.f);
    }
}

// ResidualStack (vqvae.py:7-33).  nn.ReLU(True) mutates the block input in place, so the skip
// path carries relu(x): x <- relu(x); x <- x + conv1x1(relu(conv3(x))); finally relu(x).
__device__ void residual_stack(float* x, float* tmp, int hidden, int res_hidden, int n_res, int T,
                               const float* const* c3, const float* const* c1, int ld) {
    for (int l = 0; l < n_res; ++l) {
        relu_inplace(x, hidden, T, ld);
        __syncthreads();
        conv1d_lds<3, 1, true, false>(x, hidden, T, tmp, res_hidden, T, c3[l], nullptr, 1, ld, ld);
        __syncthreads();
        conv1d_lds<1, 1, false, true>(tmp, res_hidden, T, x, hidden, T, c1[l], nullptr, 0, ld, ld);
        __syncthreads();
    }
    relu_inplace(x, hidden, T, ld);
    __syncthreads();
}

constexpr int LD = VAE_TMAX + 1;  // LDS row stride (floats)
constexpr int VAE_WIDE_T = 2 * VAE_TMAX + 2;   // positions at the L/2 resolution a tile's stride-2 convolutions touch
constexpr int VAE_LDS_FLOATS = 2 * VAE_CMAX * LD + 128 * 4 + 64 * VAE_WIDE_T;

// Time tiling at the L/4 resolution: tile `ti` keeps core positions [c0, c1) and computes the window [w0, w1) =
// core +- halo, clipped to [0, T).  One tile (T <= 32): the window is the series.
struct VaeTile {
    int c0, c1, w0, w1;
};
__host__ __device__ inline int vae_core(int T, int halo) { return T <= VAE_TMAX ? T : VAE_TMAX - 2 * halo; }
__host__ __device__ inline int vae_tiles(int T, int halo) { const int c = vae_core(T, halo); return (T + c - 1) / c; }
__device__ inline VaeTile vae_tile(int T, int halo, int ti) {
    const int core = vae_core(T, halo);
    VaeTile v;
    v.c0 = ti * core;
    v.c1 = v.c0 + core < T ? v.c0 + core : T;
    v.w0 = v.c0 - halo > 0 ? v.c0 - halo : 0;
    v.w1 = v.c1 + halo < T ? v.c1 + halo : T;
    return v;
}

// Decoder.forward (vqvae.py:97-105).  Receptive radius at the L/4 resolution: conv_1 1 + residual stack n_res + the two
// transposed convolutions 1 (the second one reads half a position beyond the first's window) = n_res + 2.
__global__ __launch_bounds__(VAE_THREADS) void vae_decode_kernel(const VaeDev w,
                                                                 const float* __restrict__ z,
                                                                 float* __restrict__ recon,
                                                                 float* __restrict__ after, int L, int W) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* bufA = smem;                   // [<=256][LD]
    float* bufB = smem + VAE_CMAX * LD;   // [<=256][LD]
    float* wide = bufB + VAE_CMAX * LD;   // [hidden/2][2 Tw] for the first transposed conv
    const int b = blockIdx.x;
    const int T = L / 4;
    const VaeTile tl = vae_tile(T, w.n_res + 2, blockIdx.y);
    const int Tw = tl.w1 - tl.w0;
    // latent (64,W) -> bufB (row stride LD covers W <= 32; W = 30 on the DiT path, L/4 on the MLP-denoiser path)
    for (int o = threadIdx.x; o < w.emb * W; o += VAE_THREADS) {
        const int c = o / W, t = o - c * W;
        bufB[c * LD + t] = z[(size_t)b * w.emb * W + o];
    }
    __syncthreads();
    interp_linear_ac(bufB, w.emb, W, LD, bufA, T, LD, tl.w0, Tw);
    __syncthreads();
    if (after) {
        const int nc = tl.c1 - tl.c0;
        for (int o = threadIdx.x; o < w.emb * nc; o += VAE_THREADS) {
            const int c = o / nc, t = tl.c0 + (o - c * nc);
            after[((size_t)b * w.emb + c) * T + t] = bufA[c * LD + (t - tl.w0)];
        }
    }
    conv1d_lds<3, 1, false, false>(bufA, w.emb, Tw, bufB, w.hidden, Tw, w.dec_conv1_w, w.dec_conv1_b, 1,
                                   LD, LD);
    __syncthreads();
    residual_stack(bufB, bufA, w.hidden, w.res_hidden, w.n_res, Tw, w.dec_c3, w.dec_c1, LD);
    const int ldw = 2 * Tw;      // `wide` holds L/2-resolution positions [2 w0, 2 w1)
    convT1d_k4s2_lds<true>(bufB, w.hidden, Tw, wide, w.hidden / 2, w.dec_ct1_w, w.dec_ct1_b, LD, ldw);
    __syncthreads();
    // last transposed conv (hidden/2 -> 1) writes the core samples [4 c0, 4 c1) straight to global
    {
        const int Tin = 2 * Tw, Cin = w.hidden / 2, i_g0 = 2 * tl.w0;
        for (int t = 4 * tl.c0 + threadIdx.x; t < 4 * tl.c1; t += VAE_THREADS) {
            float acc = w.dec_ct2_b[0];
            const int k0 = (t + 1) & 1;
            for (int ci = 0; ci < Cin; ++ci) {
#pragma unroll
                for (int kk2 = 0; kk2 < 2; ++kk2) {
                    const int kk = k0 + 2 * kk2;
                    const int i = ((t + 1 - kk) >> 1) - i_g0;
                    if (t + 1 - kk >= 0 && i >= 0 && i < Tin) acc += wide[ci * ldw + i] * w.dec_ct2_w[ci * 4 + kk];
                }
            }
            recon[(size_t)b * L + t] = acc;
        }
    }
}

// Encoder.forward (vqvae.py:57-71).  Receptive radius at the L/4 resolution behind conv_2: conv_3 1 + residual stack n_res.
// The final interpolation to 30 positions needs the WHOLE `before` row: fused here when the series is one tile, otherwise
// vae_interp_z_kernel reads the rows the tiles wrote.
__global__ __launch_bounds__(VAE_THREADS) void vae_encode_kernel(const VaeDev w,
                                                                 const float* __restrict__ x,
                                                                 float* __restrict__ z,
                                                                 float* __restrict__ before, int L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* bufA = smem;
    float* bufB = smem + VAE_CMAX * LD;
    float* wide = bufB + VAE_CMAX * LD;   // [hidden/2][Tu]: conv_1 outputs at L/2-resolution positions [u0, u1)
    const int b = blockIdx.x;
    const int T2 = L / 2, T = L / 4;
    const int half_c = w.hidden / 2;
    const VaeTile tl = vae_tile(T, w.n_res + 1, blockIdx.y);
    const int Tw = tl.w1 - tl.w0;
    // conv_2 (k4 s2 p1) output t reads conv_1 outputs 2t-1 .. 2t+2
    const int u0 = 2 * tl.w0 - 1 > 0 ? 2 * tl.w0 - 1 : 0;
    const int u1 = 2 * (tl.w1 - 1) + 3 < T2 ? 2 * (tl.w1 - 1) + 3 : T2;
    const int Tu = u1 - u0;
    // conv_1: 1 -> hidden/2, k4 s2 p1, ReLU ; input straight from global
    for (int o = threadIdx.x; o < half_c * Tu; o += VAE_THREADS) {
        const int co = o / Tu, t = u0 + (o - co * Tu);
        float acc = w.enc_conv1_b[co];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ti = 2 * t + kk - 1;
            if (ti >= 0 && ti < L) acc += w.enc_conv1_w[co * 4 + kk] * x[(size_t)b * L + ti];
        }
        wide[co * Tu + (t - u0)] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    conv1d_lds<4, 2, true, false>(wide, half_c, Tu, bufA, w.hidden, Tw, w.enc_conv2_w, w.enc_conv2_b, 1,
                                  Tu, LD, u0, tl.w0);
    __syncthreads();
    conv1d_lds<3, 1, false, false>(bufA, w.hidden, Tw, bufB, w.hidden, Tw, w.enc_conv3_w, w.enc_conv3_b,
                                   1, LD, LD);
    __syncthreads();
    residual_stack(bufB, bufA, w.hidden, w.res_hidden, w.n_res, Tw, w.enc_c3, w.enc_c1, LD);
    conv1d_lds<1, 1, false, false>(bufB, w.hidden, Tw, bufA, w.emb, Tw, w.enc_prevq_w, w.enc_prevq_b, 0,
                                   LD, LD);
    __syncthreads();
    if (before) {
        const int nc = tl.c1 - tl.c0;
        for (int o = threadIdx.x; o < w.emb * nc; o += VAE_THREADS) {
            const int c = o / nc, t = tl.c0 + (o - c * nc);
            before[((size_t)b * w.emb + c) * T + t] = bufA[c * LD + (t - tl.w0)];
        }
    }
    if (gridDim.y > 1) return;            // tiled: vae_interp_z_kernel finishes from `before`
    interp_linear_ac(bufA, w.emb, T, LD, bufB, LATW, LD);
    __syncthreads();
    for (int o = threadIdx.x; o < w.emb * LATW; o += VAE_THREADS) {
        const int c = o / LATW, t = o - c * LATW;
        z[(size_t)b * w.emb * LATW + o] = bufB[c * LD + t];
    }
}

// z = F.interpolate(before, 30, mode='linear', align_corners=True) (vqvae.py:70) from the (B,64,T) rows in global memory
__global__ __launch_bounds__(VAE_THREADS) void vae_interp_z_kernel(const float* __restrict__ before, float* __restrict__ z,
                                                                   int C, int T) {
    const float scale = (float)(T - 1) / (float)(LATW - 1);
    const float* in = before + (size_t)blockIdx.x * C * T;
    for (int o = threadIdx.x; o < C * LATW; o += VAE_THREADS) {
        const int c = o / LATW, t = o - c * LATW;
        const float real = scale * (float)t;
        const int i0 = (int)real;
        const int i1 = i0 + (i0 < T - 1 ? 1 : 0);
        const float l1 = real - (float)i0;
        const float l0 = 1.0f - l1;
        z[(size_t)blockIdx.x * C * LATW + o] = l0 * in[(size_t)c * T + i0] + l1 * in[(size_t)c * T + i1];
    }
}

// ------------------------------------------------------------------------------------------------ encoder backward
// Backward of Encoder.forward for the one configuration that trains the encoder (train.py:31-33 with `usepretrainedvae`
// false).  Two stages:
//   vae_encode_bwd_kernel   one workgroup per series (L <= 128: one tile): the forward is recomputed in LDS exactly as
//                           vae_encode_kernel computes it, every layer's INPUT leaves as an im2col'd row block X (rows = b * T
//                           + t) and its ReLU pattern stays as one bit word per channel; then the data gradients walk back
//                           through the layers in LDS (transposed convolutions, fixed summation order) and every layer's
//                           OUTPUT gradient leaves as a row block dY.  conv_1 (1 -> hidden/2, 4 taps: 320 values) is reduced
//                           per series into one partial row.
//   launch_wgrad32          dW = dY^T X per layer on the exact-fp32 MFMA (t2s_wgrad.h), deterministic two-stage reduction;
//                           vae_part_reduce_kernel adds the conv_1 partial rows in series order.
struct VaeBwdBufs {
    float *Xc2, *Xc3, *Xp, *dYp, *dY3, *dY2, *part1;
    float *Xr3[4], *Xm[4], *dYc1[4], *dYc3[4];
};
constexpr int VAE_BWD_MASK_BYTES = (128 + 128 + 5 * 128 + 4 * 256) * 4;   // ReLU bit words of vae_encode_bwd_kernel
constexpr int VAE_P1 = 320;   // conv_1 partial row: dW (hidden/2 x 4 = 256) | db (64) at hidden = 128

// din[ci][t'] (+)= sum_{co, kk : t * STRIDE + kk - pad = t'} W[co][ci][kk] * dout[co][t]   -- data gradient of conv1d_lds --
// then zeroed where the layer input's ReLU was off (mask: one word per channel, bit = position; NULL: no ReLU in front).
template <int KS, int STRIDE, bool ACCUM>
__device__ void conv1d_dgrad_lds(const float* dout, int Cout, int Tout, float* din, int Cin, int Tin,
                                 const float* __restrict__ W, int pad, int ld_out, int ld_in, const unsigned* mask,
                                 int mask_words) {
    constexpr int CP = VAE_CO_PER_THREAD;
    for (int o = threadIdx.x; o < (Cin / CP) * Tin; o += VAE_THREADS) {
        const int ci = (o / Tin) * CP, tp = o - (o / Tin) * Tin;
        float acc[CP];
#pragma unroll
        for (int u = 0; u < CP; ++u) acc[u] = 0.f;
        for (int co = 0; co < Cout; ++co) {
            const float* w = W + ((size_t)co * Cin + ci) * KS;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int num = tp + pad - kk;
                if (num < 0 || (STRIDE == 2 && (num & 1))) continue;
                const int t = STRIDE == 2 ? num >> 1 : num;
                if (t >= Tout) continue;
                const float d = dout[co * ld_out + t];
#pragma unroll
                for (int u = 0; u < CP; ++u) acc[u] += w[u * KS + kk] * d;
            }
        }
#pragma unroll
        for (int u = 0; u < CP; ++u) {
            float r = acc[u];
            if (ACCUM) r += din[(ci + u) * ld_in + tp];
            if (mask != nullptr && !((mask[(ci + u) * mask_words + (tp >> 5)] >> (tp & 31)) & 1u)) r = 0.f;
            din[(ci + u) * ld_in + tp] = r;
        }
    }
}

// one bit per (channel, position): buf > 0   (T <= 32 * words)
__device__ void relu_mask(const float* buf, int C, int T, int ld, unsigned* mask, int words) {
    for (int o = threadIdx.x; o < C * words; o += VAE_THREADS) {
        const int c = o / words, wd = o - c * words;
        unsigned m = 0;
        for (int t = 32 * wd; t < T && t < 32 * wd + 32; ++t) m |= (buf[c * ld + t] > 0.f ? 1u : 0u) << (t & 31);
        mask[o] = m;
    }
}

// X[row0 + t][ci * KS + kk] = buf[ci][t * STRIDE + kk - pad] (0 outside): the im2col'd input rows of a convolution
template <int KS, int STRIDE>
__device__ void im2col_rows(const float* buf, int Cin, int Tin, int ld, float* __restrict__ X, size_t row0, int Tout, int pad) {
    const int K = Cin * KS;
    for (int o = threadIdx.x; o < Tout * K; o += VAE_THREADS) {
        const int t = o / K, col = o - t * K;
        const int ci = col / KS, kk = col - ci * KS;
        const int ti = t * STRIDE + kk - pad;
        X[(row0 + t) * K + col] = (ti >= 0 && ti < Tin) ? buf[ci * ld + ti] : 0.f;
    }
}

// Y[row0 + t][c] = buf[c][t] for c < C, 0 for C <= c < Cpad
__device__ void rows_out(const float* buf, int C, int T, int ld, float* __restrict__ Y, size_t row0, int Cpad) {
    for (int o = threadIdx.x; o < T * Cpad; o += VAE_THREADS) {
        const int t = o / Cpad, c = o - t * Cpad;
        Y[(row0 + t) * Cpad + c] = c < C ? buf[c * ld + t] : 0.f;
    }
}

__global__ __launch_bounds__(VAE_THREADS) void vae_encode_bwd_kernel(const VaeDev w, const float* __restrict__ x,
                                                                     const float* __restrict__ dz,
                                                                     const float* __restrict__ dbefore, const VaeBwdBufs s, int L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* bufA = smem;
    float* bufB = smem + VAE_CMAX * LD;
    float* wide = bufB + VAE_CMAX * LD;                            // [hidden/2][T2]
    unsigned* m_w1 = reinterpret_cast<unsigned*>(smem + VAE_LDS_FLOATS);   // [64][2]
    unsigned* m_a = m_w1 + 128;                                    // [128]
    unsigned* m_r = m_a + 128;                                     // [5][128]: r_l = relu(h_l) for l < n_res, then r_final
    unsigned* m_m = m_r + 5 * 128;                                 // [4][256]
    const int b = blockIdx.x;
    const int T2 = L / 2, T = L / 4, H = w.hidden, R = w.res_hidden, half_c = H / 2;
    const size_t row0 = (size_t)b * T;
    // ---------------- forward, recomputed as vae_encode_kernel computes it (one tile)
    for (int o = threadIdx.x; o < half_c * T2; o += VAE_THREADS) {
        const int co = o / T2, t = o - co * T2;
        float acc = w.enc_conv1_b[co];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ti = 2 * t + kk - 1;
            if (ti >= 0 && ti < L) acc += w.enc_conv1_w[co * 4 + kk] * x[(size_t)b * L + ti];
        }
        wide[co * T2 + t] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    relu_mask(wide, half_c, T2, T2, m_w1, 2);
    im2col_rows<4, 2>(wide, half_c, T2, T2, s.Xc2, row0, T, 1);
    conv1d_lds<4, 2, true, false>(wide, half_c, T2, bufA, H, T, w.enc_conv2_w, w.enc_conv2_b, 1, T2, LD, 0, 0);
    __syncthreads();
    relu_mask(bufA, H, T, LD, m_a, 1);
    im2col_rows<3, 1>(bufA, H, T, LD, s.Xc3, row0, T, 1);
    conv1d_lds<3, 1, false, false>(bufA, H, T, bufB, H, T, w.enc_conv3_w, w.enc_conv3_b, 1, LD, LD);
    __syncthreads();
    for (int l = 0; l < w.n_res; ++l) {
        relu_inplace(bufB, H, T, LD);
        __syncthreads();
        relu_mask(bufB, H, T, LD, m_r + l * 128, 1);               // r_l > 0  <=>  h_l > 0
        im2col_rows<3, 1>(bufB, H, T, LD, s.Xr3[l], row0, T, 1);
        conv1d_lds<3, 1, true, false>(bufB, H, T, bufA, R, T, w.enc_c3[l], nullptr, 1, LD, LD);
        __syncthreads();
        relu_mask(bufA, R, T, LD, m_m + l * 256, 1);
        rows_out(bufA, R, T, LD, s.Xm[l], row0, R);
        conv1d_lds<1, 1, false, true>(bufA, R, T, bufB, H, T, w.enc_c1[l], nullptr, 0, LD, LD);
        __syncthreads();
    }
    relu_inplace(bufB, H, T, LD);
    __syncthreads();
    relu_mask(bufB, H, T, LD, m_r + w.n_res * 128, 1);
    rows_out(bufB, H, T, LD, s.Xp, row0, H);
    __syncthreads();
    // ---------------- backward.  dbefore_total = dbefore (if given) + interp^T(dz)  -> bufA [emb][T]
    {
        const float scale = (float)(T - 1) / (float)(LATW - 1);
        for (int o = threadIdx.x; o < w.emb * T; o += VAE_THREADS) {
            const int c = o / T, i = o - c * T;
            float acc = dbefore != nullptr ? dbefore[((size_t)b * w.emb + c) * T + i] : 0.f;
            const float* g = dz + ((size_t)b * w.emb + c) * LATW;
            for (int j = 0; j < LATW; ++j) {                       // the forward's own index arithmetic (interp_linear_ac)
                const float real = scale * (float)j;
                const int i0 = (int)real;
                const int i1 = i0 + (i0 < T - 1 ? 1 : 0);
                const float l1 = real - (float)i0, l0 = 1.0f - l1;
                if (i0 == i) acc += l0 * g[j];
                if (i1 == i) acc += l1 * g[j];
            }
            bufA[c * LD + i] = acc;
        }
    }
    __syncthreads();
    rows_out(bufA, w.emb, T, LD, s.dYp, row0, 128);                // padded to 128 columns for the weight-gradient GEMM
    // d r_final = Wp^T dbefore, masked by r_final > 0: the gradient at the stack's output h
    conv1d_dgrad_lds<1, 1, false>(bufA, w.emb, T, bufB, H, T, w.enc_prevq_w, 0, LD, LD, m_r + w.n_res * 128, 1);
    __syncthreads();
    for (int l = w.n_res - 1; l >= 0; --l) {
        // bufB = dL/dh_out, h_out = r + c1(m), m = relu(c3(r)), r = relu(h_in)
        rows_out(bufB, H, T, LD, s.dYc1[l], row0, H);
        conv1d_dgrad_lds<1, 1, false>(bufB, H, T, bufA, R, T, w.enc_c1[l], 0, LD, LD, m_m + l * 256, 1);       // d(pre-ReLU of m)
        __syncthreads();
        rows_out(bufA, R, T, LD, s.dYc3[l], row0, R);
        conv1d_dgrad_lds<3, 1, true>(bufA, R, T, bufB, H, T, w.enc_c3[l], 1, LD, LD, m_r + l * 128, 1);        // + skip, masked: dL/dh_in
        __syncthreads();
    }
    rows_out(bufB, H, T, LD, s.dY3, row0, H);                       // dL/d(conv_3 output)
    conv1d_dgrad_lds<3, 1, false>(bufB, H, T, bufA, H, T, w.enc_conv3_w, 1, LD, LD, m_a, 1);                     // dL/d(conv_2 pre-ReLU)
    __syncthreads();
    rows_out(bufA, H, T, LD, s.dY2, row0, H);
    conv1d_dgrad_lds<4, 2, false>(bufA, H, T, wide, half_c, T2, w.enc_conv2_w, 1, LD, T2, m_w1, 2);              // dL/d(conv_1 pre-ReLU)
    __syncthreads();
    // conv_1: dW[co][kk] = sum_u d[co][u] x[2u + kk - 1], db[co] = sum_u d[co][u]: one partial row per series
    for (int o = threadIdx.x; o < half_c * 5; o += VAE_THREADS) {
        const int co = o / 5, kk = o - co * 5;
        float acc = 0.f;
        for (int u = 0; u < T2; ++u) {
            const float d = wide[co * T2 + u];
            if (kk == 4) acc += d;
            else {
                const int ti = 2 * u + kk - 1;
                if (ti >= 0 && ti < L) acc += d * x[(size_t)b * L + ti];
            }
        }
        s.part1[(size_t)b * VAE_P1 + (kk == 4 ? half_c * 4 + co : co * 4 + kk)] = acc;
    }
}

// out[c] = sum_b part[b][c] in series order (deterministic): 32 columns x 8 row slices per workgroup, the 8 slice sums added
// in slice order
__global__ __launch_bounds__(256) void vae_part_reduce_kernel(const float* __restrict__ part, int B, int cols, float* __restrict__ dw,
                                                              int n_w, float* __restrict__ db) {
    __shared__ float red[8][32];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), sl = threadIdx.x >> 5;
    float acc = 0.f;
    if (c < cols)
        for (int r = sl; r < B; r += 8) acc += part[(size_t)r * cols + c];
    red[sl][threadIdx.x & 31] = acc;
    __syncthreads();
    if (sl != 0 || c >= cols) return;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) v += red[k][threadIdx.x & 31];
    if (c < n_w) dw[c] = v;
    else db[c - n_w] = v;
}

}  // namespace t2s

using namespace t2s;

struct t2s_vae {
    VaeDev dev{};
    float* arena = nullptr;
    bool has_encoder = false;
    bool has_decoder = false;
    // encoder backward (t2s_vae_encode_backward): row blocks + weight-gradient partial tiles, grown on demand
    float* bwd = nullptr;
    size_t bwd_rows = 0;          // rows (= B * L / 4) the row blocks hold
    int bwd_series = 0;           // series the conv_1 partial rows hold
    float* wg = nullptr;
    size_t wg_floats = 0;
    int n_cu = 0;
};

namespace {
size_t r64(size_t n) { return (n + 63) & ~size_t(63); }

struct VaeItem { const float* src; size_t n; const float** dst; };

// every tensor of *w the handle keeps a device copy of, with the VaeDev slot that points at the copy
int vae_items(const t2s_vae_weights* w, VaeDev& d, bool dec, bool enc, std::vector<VaeItem>& items) {
    const int H = w->hidden, R = w->res_hidden, E = w->emb, NR = w->n_res_layers;
    if (dec)
        items = {{w->dec_conv1_w, (size_t)H * E * 3, &d.dec_conv1_w}, {w->dec_conv1_b, (size_t)H, &d.dec_conv1_b},
                 {w->dec_ct1_w, (size_t)H * (H / 2) * 4, &d.dec_ct1_w}, {w->dec_ct1_b, (size_t)H / 2, &d.dec_ct1_b},
                 {w->dec_ct2_w, (size_t)(H / 2) * 4, &d.dec_ct2_w}, {w->dec_ct2_b, 1, &d.dec_ct2_b}};
    for (int l = 0; dec && l < NR; ++l) {
        T2S_REQUIRE(w->dec_stack.conv3_w[l] && w->dec_stack.conv1_w[l], "t2s_vae: NULL decoder residual weight %d", l);
        items.push_back({w->dec_stack.conv3_w[l], (size_t)R * H * 3, &d.dec_c3[l]});
        items.push_back({w->dec_stack.conv1_w[l], (size_t)H * R, &d.dec_c1[l]});
    }
    if (enc) {
        items.push_back({w->enc_conv1_w, (size_t)(H / 2) * 4, &d.enc_conv1_w});
        items.push_back({w->enc_conv1_b, (size_t)H / 2, &d.enc_conv1_b});
        items.push_back({w->enc_conv2_w, (size_t)H * (H / 2) * 4, &d.enc_conv2_w});
        items.push_back({w->enc_conv2_b, (size_t)H, &d.enc_conv2_b});
        items.push_back({w->enc_conv3_w, (size_t)H * H * 3, &d.enc_conv3_w});
        items.push_back({w->enc_conv3_b, (size_t)H, &d.enc_conv3_b});
        items.push_back({w->enc_prevq_w, (size_t)E * H, &d.enc_prevq_w});
        items.push_back({w->enc_prevq_b, (size_t)E, &d.enc_prevq_b});
        for (int l = 0; l < NR; ++l) {
            T2S_REQUIRE(w->enc_stack.conv3_w[l] && w->enc_stack.conv1_w[l], "t2s_vae: NULL encoder residual weight %d", l);
            items.push_back({w->enc_stack.conv3_w[l], (size_t)R * H * 3, &d.enc_c3[l]});
            items.push_back({w->enc_stack.conv1_w[l], (size_t)H * R, &d.enc_c1[l]});
        }
    }
    return T2S_OK;
}
}

extern "C" int t2s_vae_create(const t2s_vae_weights* w, t2s_vae** out) {
    T2S_REQUIRE(w && out, "t2s_vae_create: NULL argument");
    T2S_REQUIRE(w->hidden > 0 && w->hidden <= 128 && w->hidden % 2 == 0, "t2s_vae_create: hidden=%d unsupported (<=128, even)", w->hidden);
    T2S_REQUIRE(w->res_hidden > 0 && w->res_hidden <= VAE_CMAX, "t2s_vae_create: res_hidden=%d unsupported (<=256)", w->res_hidden);
    T2S_REQUIRE(w->n_res_layers >= 0 && w->n_res_layers <= 4, "t2s_vae_create: n_res_layers=%d unsupported (<=4)", w->n_res_layers);
    T2S_REQUIRE(w->emb == T2S_LAT_C, "t2s_vae_create: embedding_dim=%d must be 64", w->emb);
    const bool dec = w->dec_conv1_w != nullptr;
    const bool enc = w->enc_conv1_w != nullptr;
    T2S_REQUIRE(dec || enc, "t2s_vae_create: neither decoder nor encoder weights given");
    if (dec)
        T2S_REQUIRE(w->dec_conv1_b && w->dec_ct1_w && w->dec_ct1_b && w->dec_ct2_w && w->dec_ct2_b,
                    "t2s_vae_create: partial decoder weights");
    const int H = w->hidden, R = w->res_hidden, E = w->emb, NR = w->n_res_layers;
    if (enc)
        T2S_REQUIRE(w->enc_conv1_b && w->enc_conv2_w && w->enc_conv2_b && w->enc_conv3_w && w->enc_conv3_b &&
                        w->enc_prevq_w && w->enc_prevq_b,
                    "t2s_vae_create: partial encoder weights");
    t2s_vae* h = new t2s_vae();
    VaeDev& d = h->dev;
    d.hidden = H; d.res_hidden = R; d.n_res = NR; d.emb = E;
    std::vector<VaeItem> items;
    {
        const int rc_items = vae_items(w, d, dec, enc, items);
        if (rc_items != T2S_OK) {
            delete h;
            return rc_items;
        }
    }
    for (auto& it : items) {      // sizes follow from hidden / res_hidden / emb: a tensor of another shape is an error code
        const int rc_e = it.src ? check_device_extent(it.src, it.n * sizeof(float), "t2s_vae_create: a weight tensor (hyper-parameters vs tensor sizes)") : T2S_OK;
        if (rc_e != T2S_OK) {
            delete h;
            return rc_e;
        }
    }
    size_t total = 0;
    for (auto& it : items) total += r64(it.n);
    hipError_t e = hipMalloc(&h->arena, total * sizeof(float));
    if (e != hipSuccess) {
        set_error("t2s_vae_create: hipMalloc failed: %s", hipGetErrorString(e));
        delete h;
        return T2S_E_HIP;
    }
    size_t off = 0;
    for (auto& it : items) {
        // (asynchronous on the default stream -- ordered behind whatever the caller's framework enqueued there to produce the
        // tensors -- and ONE synchronisation below; a synchronous hipMemcpy is what HIP refuses while another thread has a
        // stream capture open)
        e = hipMemcpyAsync(h->arena + off, it.src, it.n * sizeof(float), hipMemcpyDeviceToDevice, nullptr);
        if (e != hipSuccess) {
            set_error("t2s_vae_create: weight copy failed: %s", hipGetErrorString(e));
            t2s_vae_destroy(h);
            return T2S_E_HIP;
        }
        *it.dst = h->arena + off;
        off += r64(it.n);
    }
    e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        set_error("t2s_vae_create: weight copy failed: %s", hipGetErrorString(e));
        t2s_vae_destroy(h);
        return T2S_E_HIP;
    }
    h->has_encoder = enc;
    h->has_decoder = dec;
    static bool attr = false;
    if (!attr) {
        const int bytes = VAE_LDS_FLOATS * 4;
        T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(vae_decode_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(vae_encode_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(vae_encode_bwd_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, bytes + VAE_BWD_MASK_BYTES));
        attr = true;
    }
    *out = h;
    return T2S_OK;
}

extern "C" void t2s_vae_destroy(t2s_vae* h) {
    if (!h) return;
    if (h->arena) (void)hipFree(h->arena);
    if (h->bwd) (void)hipFree(h->bwd);
    if (h->wg) (void)hipFree(h->wg);
    delete h;
}

extern "C" int t2s_vae_update_weights(t2s_vae* h, const t2s_vae_weights* w, void* stream) {
    T2S_REQUIRE(h && w, "t2s_vae_update_weights: NULL argument");
    T2S_REQUIRE(w->hidden == h->dev.hidden && w->res_hidden == h->dev.res_hidden && w->n_res_layers == h->dev.n_res && w->emb == h->dev.emb,
                "t2s_vae_update_weights: hyper-parameters differ from the handle's (hidden %d, res_hidden %d, layers %d, emb %d)",
                h->dev.hidden, h->dev.res_hidden, h->dev.n_res, h->dev.emb);
    T2S_REQUIRE((w->dec_conv1_w != nullptr) == h->has_decoder && (w->enc_conv1_w != nullptr) == h->has_encoder,
                "t2s_vae_update_weights: the handle was created with %s%s weights", h->has_encoder ? "encoder " : "", h->has_decoder ? "decoder" : "");
    VaeDev d = h->dev;                       // the slots keep pointing at the handle's copies: only the contents change
    std::vector<VaeItem> items;
    const int rc = vae_items(w, d, h->has_decoder, h->has_encoder, items);
    if (rc != T2S_OK) return rc;
    for (auto& it : items) {
        T2S_REQUIRE(it.src, "t2s_vae_update_weights: NULL weight pointer");
        const int rc_e = check_device_extent(it.src, it.n * sizeof(float), "t2s_vae_update_weights: a weight tensor");
        if (rc_e != T2S_OK) return rc_e;
        T2S_HIP_CHECK(hipMemcpyAsync(const_cast<float*>(*it.dst), it.src, it.n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return T2S_OK;
}

extern "C" int t2s_vae_decode(t2s_vae* h, const float* z, float* recon, float* after, int B, int L,
                              void* stream) {
    T2S_REQUIRE(h && z && recon, "t2s_vae_decode: NULL argument");
    T2S_REQUIRE(h->has_decoder, "t2s_vae_decode: handle was created without decoder weights");
    T2S_REQUIRE(B > 0, "t2s_vae_decode: B=%d", B);
    T2S_REQUIRE(L >= 4 && L % 4 == 0 && L <= (1 << 20), "t2s_vae_decode: L=%d unsupported (a multiple of 4)", L);
    vae_decode_kernel<<<dim3(B, vae_tiles(L / 4, h->dev.n_res + 2)), VAE_THREADS, VAE_LDS_FLOATS * 4, (hipStream_t)stream>>>(
        h->dev, z, recon, after, L, LATW);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

extern "C" int t2s_vae_decode_w(t2s_vae* h, const float* z, float* recon, float* after, int B, int L, int latent_w,
                                void* stream) {
    T2S_REQUIRE(h && z && recon, "t2s_vae_decode_w: NULL argument");
    T2S_REQUIRE(h->has_decoder, "t2s_vae_decode_w: handle was created without decoder weights");
    T2S_REQUIRE(B > 0, "t2s_vae_decode_w: B=%d", B);
    T2S_REQUIRE(L >= 4 && L % 4 == 0 && L <= (1 << 20), "t2s_vae_decode_w: L=%d unsupported (a multiple of 4)", L);
    T2S_REQUIRE(latent_w >= 1 && latent_w <= VAE_TMAX, "t2s_vae_decode_w: latent width %d unsupported (1..%d)", latent_w, VAE_TMAX);
    vae_decode_kernel<<<dim3(B, vae_tiles(L / 4, h->dev.n_res + 2)), VAE_THREADS, VAE_LDS_FLOATS * 4, (hipStream_t)stream>>>(
        h->dev, z, recon, after, L, latent_w);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

extern "C" int t2s_vae_encode(t2s_vae* h, const float* x, float* z, float* before, int B, int L,
                              void* stream) {
    T2S_REQUIRE(h && x && z, "t2s_vae_encode: NULL argument");
    T2S_REQUIRE(h->has_encoder, "t2s_vae_encode: handle was created without encoder weights");
    T2S_REQUIRE(B > 0, "t2s_vae_encode: B=%d", B);
    T2S_REQUIRE(L >= 4 && L % 4 == 0 && L <= (1 << 20), "t2s_vae_encode: L=%d unsupported (a multiple of 4)", L);
    const int tiles = vae_tiles(L / 4, h->dev.n_res + 1);
    T2S_REQUIRE(tiles == 1 || before, "t2s_vae_encode: L=%d > 128 runs in time tiles and needs the `before` output buffer "
                                      "(the interpolation to the latent reads the whole row)", L);
    vae_encode_kernel<<<dim3(B, tiles), VAE_THREADS, VAE_LDS_FLOATS * 4, (hipStream_t)stream>>>(h->dev, x, z, before, L);
    T2S_LAUNCH_CHECK();
    if (tiles > 1) vae_interp_z_kernel<<<B, VAE_THREADS, 0, (hipStream_t)stream>>>(before, z, h->dev.emb, L / 4);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

extern "C" int t2s_vae_encode_backward(t2s_vae* h, const float* x, const float* dz, const float* dbefore, const t2s_vae_enc_grads* g,
                                       int B, int L, void* stream) {
    T2S_REQUIRE(h && x && dz && g, "t2s_vae_encode_backward: NULL argument");
    T2S_REQUIRE(h->has_encoder, "t2s_vae_encode_backward: handle was created without encoder weights");
    const VaeDev& d = h->dev;
    // the weight-gradient GEMMs work on 128-wide tiles: the reference's default LA-VAE (pretrained_lavae_unified.py:119-122)
    T2S_REQUIRE(d.hidden == 128 && d.res_hidden % 128 == 0 && d.emb == 64,
                "t2s_vae_encode_backward: hidden=%d res_hidden=%d emb=%d unsupported (hidden 128, res_hidden 128 / 256, emb 64)", d.hidden,
                d.res_hidden, d.emb);
    T2S_REQUIRE(B > 0 && L >= 8 && L % 4 == 0 && L <= 4 * VAE_TMAX, "t2s_vae_encode_backward: B=%d L=%d unsupported (L <= 128, a multiple of 4)", B, L);
    T2S_REQUIRE(g->conv1_w && g->conv1_b && g->conv2_w && g->conv2_b && g->conv3_w && g->conv3_b && g->prevq_w && g->prevq_b,
                "t2s_vae_encode_backward: NULL gradient pointer");
    for (int l = 0; l < d.n_res; ++l)
        T2S_REQUIRE(g->stack_conv3_w[l] && g->stack_conv1_w[l], "t2s_vae_encode_backward: NULL gradient pointer of residual layer %d", l);
    hipStream_t st = (hipStream_t)stream;
    const int T = L / 4, H = d.hidden, R = d.res_hidden, NR = d.n_res;
    const size_t rows = (size_t)B * T;
    if (h->n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        T2S_HIP_CHECK(hipGetDevice(&dev));
        T2S_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        h->n_cu = prop.multiProcessorCount;
    }
    // row blocks: floats per row, in the order the pointers are handed out below
    const size_t per_row = 256 + 384 + 128 + 128 + 128 + 128 + (size_t)NR * (384 + R + 128 + R);
    if (rows > h->bwd_rows || B > h->bwd_series) {
        const size_t r2 = rows > h->bwd_rows ? rows : h->bwd_rows;
        const int b2 = B > h->bwd_series ? B : h->bwd_series;
        if (h->bwd) T2S_HIP_CHECK(hipFree(h->bwd));
        h->bwd = nullptr;
        h->bwd_rows = 0;          // (a failed hipMalloc below must not leave sizes that vouch for a NULL buffer)
        h->bwd_series = 0;
        T2S_HIP_CHECK(hipMalloc((void**)&h->bwd, (r2 * per_row + (size_t)b2 * VAE_P1 + 128 * 128 + 128) * sizeof(float)));
        h->bwd_rows = r2;
        h->bwd_series = b2;
    }
    {   // weight-gradient partial tiles: the largest of this encoder's shapes
        size_t need = 0;
        const int shapes[5][2] = {{128, 256}, {128, 384}, {R, 384}, {128, R}, {128, 128}};
        for (auto& sh : shapes) {
            const int tiles = (sh[0] / 128) * (sh[1] / 128);
            int per = 3 * h->n_cu / tiles;
            per = per < 1 ? 1 : per;
            const size_t n = (size_t)per * tiles * (128 * 128) + (size_t)per * (sh[0] / 128) * 128;
            need = n > need ? n : need;
        }
        if (need > h->wg_floats) {
            if (h->wg) T2S_HIP_CHECK(hipFree(h->wg));
            h->wg = nullptr;
            h->wg_floats = 0;
            T2S_HIP_CHECK(hipMalloc((void**)&h->wg, need * sizeof(float)));
            h->wg_floats = need;
        }
    }
    VaeBwdBufs s{};
    float* p = h->bwd;
    auto take = [&](size_t cols) { float* q = p; p += h->bwd_rows * cols; return q; };
    s.Xc2 = take(256); s.Xc3 = take(384); s.Xp = take(128); s.dYp = take(128); s.dY3 = take(128); s.dY2 = take(128);
    for (int l = 0; l < NR; ++l) { s.Xr3[l] = take(384); s.Xm[l] = take(R); s.dYc1[l] = take(128); s.dYc3[l] = take(R); }
    s.part1 = p;
    float* tmp_w = p + (size_t)h->bwd_series * VAE_P1;             // (128,128) + (128): prevq padded to 128 outputs
    float* tmp_b = tmp_w + 128 * 128;
    vae_encode_bwd_kernel<<<B, VAE_THREADS, VAE_LDS_FLOATS * 4 + VAE_BWD_MASK_BYTES, st>>>(d, x, dz, dbefore, s, L);
    T2S_LAUNCH_CHECK();
    int rc;
    const int M = (int)rows;
    if ((rc = launch_wgrad32(s.dY2, s.Xc2, g->conv2_w, g->conv2_b, M, 128, 256, h->wg, h->wg_floats, h->n_cu, st))) return rc;
    if ((rc = launch_wgrad32(s.dY3, s.Xc3, g->conv3_w, g->conv3_b, M, 128, 384, h->wg, h->wg_floats, h->n_cu, st))) return rc;
    for (int l = 0; l < NR; ++l) {
        if ((rc = launch_wgrad32(s.dYc3[l], s.Xr3[l], g->stack_conv3_w[l], nullptr, M, R, 384, h->wg, h->wg_floats, h->n_cu, st))) return rc;
        if ((rc = launch_wgrad32(s.dYc1[l], s.Xm[l], g->stack_conv1_w[l], nullptr, M, 128, R, h->wg, h->wg_floats, h->n_cu, st))) return rc;
    }
    if ((rc = launch_wgrad32(s.dYp, s.Xp, tmp_w, tmp_b, M, 128, 128, h->wg, h->wg_floats, h->n_cu, st))) return rc;
    T2S_HIP_CHECK(hipMemcpyAsync(g->prevq_w, tmp_w, (size_t)64 * 128 * sizeof(float), hipMemcpyDeviceToDevice, st));
    T2S_HIP_CHECK(hipMemcpyAsync(g->prevq_b, tmp_b, (size_t)64 * sizeof(float), hipMemcpyDeviceToDevice, st));
    vae_part_reduce_kernel<<<VAE_P1 / 32, 256, 0, st>>>(s.part1, B, VAE_P1, g->conv1_w, 256, g->conv1_b);
    T2S_LAUNCH_CHECK();
    (void)H;
    return T2S_OK;
}
