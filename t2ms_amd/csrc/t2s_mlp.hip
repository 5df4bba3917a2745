// The MLP denoiser of BASELINE configs[0] (reference model/denoiser/mlp.py:49-94) as ONE kernel per forward:
// eight MLPlayers on a (64 channels x 6 positions) latent, one 256-thread workgroup per series, activations in LDS /
// registers from the first layer to the last, fp32 FMA arithmetic (the layers are 0.8 MFLOP each: launch- and
// latency-bound, not a matrix-core problem -- a torch-op evaluation is ~50 launches per layer).
//
// Per layer (mlp.py:71-85), for one series:
//   h[p][c] = x[c][p] + temb[c]                                                  (mlp.py:73-74)
//   text given:  h[p][:] += proj(value(text))                                   (mlp.py:76-79, 21-47)
//       The cross attention's six keys / values are the SAME row (text repeated over the six positions, mlp.py:77), so
//       every softmax row is uniform whatever the queries are and the attended value is value(text) itself: query and
//       key cannot influence the result and are not evaluated (the reference's 6 x (1/6) v differs from v by one fp32
//       rounding).
//   h = LayerNorm_64(h) (norm2, eps 1e-5, affine)                                (mlp.py:80)
//   h = h + W2 relu(W1 h + b1) + b2          (64 -> 256 -> 64, per position)     (mlp.py:81)
//   x'[c][:] = W4 relu(W3 h[:][c] + b3) + b4 (6 -> 256 -> 6, per channel)        (mlp.py:83-84)
//
// Weights: t2s_mlp_pack transposes the torch Linear weights once into one buffer ([in][out], so that the thread that owns
// an output reads its weights coalesced with its neighbours); t2s_mlp_forward takes only that buffer.  Both are
// stateless: the caller owns the packed buffer and re-packs after changing the weights.
#include "t2s_common.h"

namespace t2s {
namespace {

constexpr int MC = T2S_MLP_WIDTH;       // 64 channels
constexpr int MP = T2S_MLP_POSITIONS;   // 6 positions
constexpr int MT = T2S_MLP_TEXT_DIM;    // 128
constexpr int MH = T2S_MLP_HIDDEN;      // 256
constexpr int ML = T2S_MLP_LAYERS;      // 8

// packed layer (floats)
constexpr int O_WV = 0;                    // [128][64]  value.weight^T
constexpr int O_BV = O_WV + MT * MC;       // [64]
constexpr int O_WP = O_BV + MC;            // [64][64]   proj.weight^T
constexpr int O_BP = O_WP + MC * MC;       // [64]
constexpr int O_LG = O_BP + MC;            // [64]       norm2.weight
constexpr int O_LB = O_LG + MC;            // [64]       norm2.bias
constexpr int O_W1 = O_LB + MC;            // [64][256]  mlp.0.weight^T
constexpr int O_B1 = O_W1 + MC * MH;       // [256]
constexpr int O_W2 = O_B1 + MH;            // [256][64]  mlp.2.weight^T
constexpr int O_B2 = O_W2 + MH * MC;       // [64]
constexpr int O_W3 = O_B2 + MC;            // [256][8]   mlp2.0.weight rows (6 used) | b3 | pad
constexpr int O_W4 = O_W3 + MH * 8;        // [256][8]   mlp2.2.weight^T rows (6 used), pad
constexpr int O_B4 = O_W4 + MH * 8;        // [8]        (6 used)
constexpr int LAYER_FLOATS = O_B4 + 8;
static_assert(LAYER_FLOATS * ML == T2S_MLP_PACKED_FLOATS, "include/t2s.h: T2S_MLP_PACKED_FLOATS");
static_assert(LAYER_FLOATS % 4 == 0, "layers stay 16-byte aligned");

struct PackArgs {
    t2s_mlp_weights w;
    float* dst;
};

// one workgroup per (layer, tensor group); plain gathers -- this runs once per set of weights
__global__ __launch_bounds__(256) void mlp_pack_kernel(PackArgs a) {
    const int layer = blockIdx.x;
    const t2s_mlp_layer_weights& w = a.w.layer[layer];
    float* d = a.dst + (size_t)layer * LAYER_FLOATS;
    const int tid = threadIdx.x;
    for (int i = tid; i < MT * MC; i += 256) d[O_WV + i] = w.value_w[(i % MC) * MT + i / MC];      // [k][c] <- [c][k]
    for (int i = tid; i < MC * MC; i += 256) d[O_WP + i] = w.proj_w[(i % MC) * MC + i / MC];
    for (int i = tid; i < MC * MH; i += 256) d[O_W1 + i] = w.mlp0_w[(i % MH) * MC + i / MH];       // [c][j] <- [j][c]
    for (int i = tid; i < MH * MC; i += 256) d[O_W2 + i] = w.mlp2_w[(i % MC) * MH + i / MC];       // [j][c] <- [c][j]
    for (int i = tid; i < MH * 8; i += 256) {
        const int j = i >> 3, p = i & 7;
        d[O_W3 + i] = p < MP ? w.pos0_w[j * MP + p] : (p == 6 ? w.pos0_b[j] : 0.f);               // row j: W3[j][0..5], b3[j]
        d[O_W4 + i] = p < MP ? w.pos2_w[p * MH + j] : 0.f;                                          // row j: W4[0..5][j]
    }
    if (tid < MC) {
        d[O_BV + tid] = w.value_b[tid];
        d[O_BP + tid] = w.proj_b[tid];
        d[O_LG + tid] = w.norm2_w[tid];
        d[O_LB + tid] = w.norm2_b[tid];
        d[O_B2 + tid] = w.mlp2_b[tid];
    }
    d[O_B1 + tid] = w.mlp0_b[tid];
    if (tid < 8) d[O_B4 + tid] = tid < MP ? w.pos2_b[tid] : 0.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// EAGER: every weight of a layer is requested at the layer's start and held in registers (450 of them: one workgroup per CU) --
// the latency form, for launches that do not fill the chip anyway; otherwise each phase fetches its own weights with a short
// unroll (84 registers, several workgroups per CU cover each other's round trips) -- the throughput form.
template <bool EAGER>
__global__ __launch_bounds__(256) void mlp_forward_kernel(const float* __restrict__ packed, const float* __restrict__ x,
                                                          const float* __restrict__ t, const float* __restrict__ freqs,
                                                          const float* __restrict__ text, float* __restrict__ out) {
    __shared__ float xs[MC * MP];          // current latent [c][p]
    __shared__ float hs[MP][MC];           // normalised rows
    __shared__ float hid[MP][MH];          // relu(W1 h + b1)
    __shared__ float red[4][MP][MC];       // partial sums of the four input quarters, added in a fixed order
    __shared__ float txt[MT];
    __shared__ float vs[MC], as[MC];
    __shared__ f32x4 w3s[2 * MH], w4s[2 * MH];   // this layer's position-MLP rows
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < MC * MP; i += 256) xs[i] = x[(size_t)b * MC * MP + i];
    if (text != nullptr && tid < MT) txt[tid] = text[(size_t)b * MT + tid];
    // TimeEmbedding(64), mlp.py:5-18: [sin(100 t / f) | cos(100 t / f)] -- this lane's channel; the same in every layer
    const float targ = (t[b] * 100.0f) / freqs[lane & 31];
    const float te = lane < 32 ? sinf(targ) : cosf(targ);
    __syncthreads();

    for (int layer = 0; layer < ML; ++layer) {
        const float* w = packed + (size_t)layer * LAYER_FLOATS;
        // Every weight this thread needs in the layer is requested HERE, before the first dependent instruction: the layer
        // is a chain of short phases and a phase that starts by fetching its weights pays an L2 round trip per unrolled
        // group (measured: 24 us per layer that way, 190 us per forward whatever the batch).
        constexpr int UNR = EAGER ? MC / 4 : 4;
        float wv1[EAGER ? MC : 1], wv2[EAGER ? MC : 1], wvv[EAGER ? 32 : 1], wvp[EAGER ? 16 : 1];
        const float* w1p = w + O_W1 + tid;                       // fc1 column of hidden unit tid (stride MH)
        const float* w2p = w + O_W2 + 64 * wave * MC + lane;     // fc2: this wave's quarter, channel lane (stride MC)
        const float* wvq = w + O_WV + 32 * wave * MC + lane;
        const float* wpq = w + O_WP + 16 * wave * MC + lane;
        if constexpr (EAGER) {
#pragma unroll
            for (int c = 0; c < MC; ++c) wv1[c] = w1p[c * MH];
#pragma unroll
            for (int jj = 0; jj < MC; ++jj) wv2[jj] = w2p[jj * MC];
            if (text != nullptr) {
#pragma unroll
                for (int kk = 0; kk < 32; ++kk) wvv[kk] = wvq[kk * MC];
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) wvp[kk] = wpq[kk * MC];
            }
        }
        const float b1 = w[O_B1 + tid], b2 = w[O_B2 + lane], g = w[O_LG + lane], be = w[O_LB + lane];
        const float bv = w[O_BV + lane], bp = w[O_BP + lane];
        {   // the position MLP's rows (uniform per wave later): through LDS, one coalesced copy
            const f32x4* src3 = reinterpret_cast<const f32x4*>(w + O_W3) + 2 * tid;
            const f32x4* src4 = reinterpret_cast<const f32x4*>(w + O_W4) + 2 * tid;
            w3s[2 * tid] = src3[0];
            w3s[2 * tid + 1] = src3[1];
            w4s[2 * tid] = src4[0];
            w4s[2 * tid + 1] = src4[1];
        }
        float add = te;                                   // temb[c] (+ the cross-attention row)
        if (text != nullptr) {
            // v = value(text): channel = lane, the 128 inputs in four quarters (one per wave)
            float acc = 0.f;
            if constexpr (EAGER) {
#pragma unroll
                for (int kk = 0; kk < 32; ++kk) acc = fmaf(wvv[kk], txt[32 * wave + kk], acc);
            } else {
#pragma unroll 8
                for (int kk = 0; kk < 32; ++kk) acc = fmaf(wvq[kk * MC], txt[32 * wave + kk], acc);
            }
            red[wave][0][lane] = acc;
            __syncthreads();
            if (wave == 0) vs[lane] = bv + ((red[0][0][lane] + red[1][0][lane]) + (red[2][0][lane] + red[3][0][lane]));
            __syncthreads();
            acc = 0.f;
            if constexpr (EAGER) {
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) acc = fmaf(wvp[kk], vs[16 * wave + kk], acc);
            } else {
#pragma unroll 8
                for (int kk = 0; kk < 16; ++kk) acc = fmaf(wpq[kk * MC], vs[16 * wave + kk], acc);
            }
            red[wave][1][lane] = acc;
            __syncthreads();
            if (wave == 0) as[lane] = bp + ((red[0][1][lane] + red[1][1][lane]) + (red[2][1][lane] + red[3][1][lane]));
            __syncthreads();
            add += as[lane];
        }
        // LayerNorm over the 64 channels of a position: wave w takes positions w and w + 4
        for (int p = wave; p < MP; p += 4) {
            const float v = xs[lane * MP + p] + add;
            const float mean = wave_sum(v) * (1.0f / MC);
            const float dlt = v - mean;
            const float var = wave_sum(dlt * dlt) * (1.0f / MC);
            hs[p][lane] = dlt * (1.0f / sqrtf(var + 1e-5f)) * g + be;
        }
        __syncthreads();
        // fc1: thread j owns hidden unit j for the six positions
        {
            float acc[MP];
#pragma unroll
            for (int p = 0; p < MP; ++p) acc[p] = b1;
#pragma unroll UNR
            for (int c = 0; c < MC; c += 4) {
                float wq[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) wq[e] = EAGER ? wv1[EAGER ? c + e : 0] : w1p[(c + e) * MH];
#pragma unroll
                for (int p = 0; p < MP; ++p) {
                    const f32x4 h4 = *reinterpret_cast<const f32x4*>(&hs[p][c]);
                    acc[p] = fmaf(wq[0], h4.x, acc[p]);
                    acc[p] = fmaf(wq[1], h4.y, acc[p]);
                    acc[p] = fmaf(wq[2], h4.z, acc[p]);
                    acc[p] = fmaf(wq[3], h4.w, acc[p]);
                }
            }
#pragma unroll
            for (int p = 0; p < MP; ++p) hid[p][tid] = fmaxf(acc[p], 0.f);
        }
        __syncthreads();
        // fc2: channel = lane, hidden units in four quarters (one per wave)
        {
            float acc[MP];
#pragma unroll
            for (int p = 0; p < MP; ++p) acc[p] = 0.f;
#pragma unroll UNR
            for (int jj = 0; jj < MC; jj += 4) {
                float wq[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) wq[e] = EAGER ? wv2[EAGER ? jj + e : 0] : w2p[(jj + e) * MC];
#pragma unroll
                for (int p = 0; p < MP; ++p) {
                    const f32x4 h4 = *reinterpret_cast<const f32x4*>(&hid[p][64 * wave + jj]);
                    acc[p] = fmaf(wq[0], h4.x, acc[p]);
                    acc[p] = fmaf(wq[1], h4.y, acc[p]);
                    acc[p] = fmaf(wq[2], h4.z, acc[p]);
                    acc[p] = fmaf(wq[3], h4.w, acc[p]);
                }
            }
#pragma unroll
            for (int p = 0; p < MP; ++p) red[wave][p][lane] = acc[p];
        }
        __syncthreads();
        // g[c][p] = h[p][c] + b2[c] + sum of the four quarters; every wave needs all six of its channel
        float gv[MP];
#pragma unroll
        for (int p = 0; p < MP; ++p)
            gv[p] = hs[p][lane] + (b2 + ((red[0][p][lane] + red[1][p][lane]) + (red[2][p][lane] + red[3][p][lane])));
        __syncthreads();   // red is rewritten below
        // position MLP of channel `lane`: hidden units in four quarters (one per wave); the rows are wave-uniform LDS reads
        {
            float acc[MP];
#pragma unroll
            for (int p = 0; p < MP; ++p) acc[p] = 0.f;
            const f32x4* w3 = w3s + 2 * 64 * wave;
            const f32x4* w4 = w4s + 2 * 64 * wave;
#pragma unroll 8
            for (int j = 0; j < 64; ++j) {
                const f32x4 a0 = w3[2 * j], a1 = w3[2 * j + 1];     // W3[j][0..3] | W3[j][4..5], b3[j], 0
                float hdn = a1.z;
                hdn = fmaf(a0.x, gv[0], hdn);
                hdn = fmaf(a0.y, gv[1], hdn);
                hdn = fmaf(a0.z, gv[2], hdn);
                hdn = fmaf(a0.w, gv[3], hdn);
                hdn = fmaf(a1.x, gv[4], hdn);
                hdn = fmaf(a1.y, gv[5], hdn);
                hdn = fmaxf(hdn, 0.f);
                const f32x4 c0 = w4[2 * j], c1 = w4[2 * j + 1];     // W4[0..3][j] | W4[4..5][j], 0, 0
                acc[0] = fmaf(c0.x, hdn, acc[0]);
                acc[1] = fmaf(c0.y, hdn, acc[1]);
                acc[2] = fmaf(c0.z, hdn, acc[2]);
                acc[3] = fmaf(c0.w, hdn, acc[3]);
                acc[4] = fmaf(c1.x, hdn, acc[4]);
                acc[5] = fmaf(c1.y, hdn, acc[5]);
            }
#pragma unroll
            for (int p = 0; p < MP; ++p) red[wave][p][lane] = acc[p];
        }
        __syncthreads();
        for (int i = tid; i < MC * MP; i += 256) {
            const int c = i / MP, p = i - c * MP;
            xs[i] = w[O_B4 + p] + ((red[0][p][c] + red[1][p][c]) + (red[2][p][c] + red[3][p][c]));
        }
        __syncthreads();
    }
    for (int i = tid; i < MC * MP; i += 256) out[(size_t)b * MC * MP + i] = xs[i];
}


// ------------------------------------------------------------------------------------------------ backward (training)
// One workgroup per series again: the forward is re-run keeping every layer's INPUT in LDS (8 x 384 floats), then the layers are
// walked backwards, each recomputing its own activations from its input (cheap: the layer is 0.8 MFLOP) and producing
//   * the gradient of its input (handed to the layer below),
//   * this series' contribution to the layer's 14 parameter gradients, written to part[b][layer][N_*] in the parameters' own
//     (torch) layouts.
// mlp_grad_reduce_kernel then adds the B contributions of every element in series order (deterministic) into the gradient
// tensors.  cross_attn.query / .key receive exact zeros (they cannot influence the forward, see the top of the file).
constexpr int N_WV = 0;                     // value.weight (64,128)
constexpr int N_BV = N_WV + MC * MT;
constexpr int N_WP = N_BV + MC;             // proj.weight (64,64)
constexpr int N_BP = N_WP + MC * MC;
constexpr int N_LG = N_BP + MC;
constexpr int N_LB = N_LG + MC;
constexpr int N_W1 = N_LB + MC;             // mlp.0.weight (256,64)
constexpr int N_B1 = N_W1 + MH * MC;
constexpr int N_W2 = N_B1 + MH;             // mlp.2.weight (64,256)
constexpr int N_B2 = N_W2 + MC * MH;
constexpr int N_W3 = N_B2 + MC;             // mlp2.0.weight (256,6)
constexpr int N_B3 = N_W3 + MH * MP;
constexpr int N_W4 = N_B3 + MH;             // mlp2.2.weight (6,256)
constexpr int N_B4 = N_W4 + MP * MH;        // (6) + 2 pad
constexpr int N_LAYER = N_B4 + 8;
static_assert(N_LAYER * ML == T2S_MLP_GRAD_PART_FLOATS, "include/t2s.h: T2S_MLP_GRAD_PART_FLOATS");

struct BwdArgs {
    t2s_mlp_weights w;        // native layouts: mlp.0.weight [j][c] and mlp.2.weight [c][j] are read as they are
    const float* packed;
    const float *x, *t, *freqs, *text, *dout;
    float *dx, *part;
};

__global__ __launch_bounds__(256) void mlp_backward_kernel(const BwdArgs a) {
    __shared__ float xin[ML][MC * MP];     // input of every layer, [c][p]
    __shared__ float hs[MP][MC];           // LayerNorm output
    __shared__ float xh[MP][MC];           // normalised (before the affine)
    __shared__ float rs[MP];               // 1 / sqrt(var + eps)
    __shared__ float hid[MP][MH];          // relu(W1 h + b1)
    __shared__ float ys[MC][MP + 2];       // y = h + W2 hid + b2, [c][p] (row padded to 8)
    __shared__ float dys[MP][MC];          // gradient of y / of hs
    __shared__ float dp1[MP][MH];          // gradient of the fc1 pre-activation
    __shared__ float red[4][MP][MC];
    __shared__ float dcur[MC][MP + 2];     // gradient of the layer's output, [c][p]
    __shared__ float txt[MT];
    __shared__ float vs[MC], dadd[MC], dvs[MC];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool has_text = a.text != nullptr;
    for (int i = tid; i < MC * MP; i += 256) xin[0][i] = a.x[(size_t)b * MC * MP + i];
    if (has_text && tid < MT) txt[tid] = a.text[(size_t)b * MT + tid];
    const float targ = (a.t[b] * 100.0f) / a.freqs[lane & 31];
    const float te = lane < 32 ? sinf(targ) : cosf(targ);
    __syncthreads();

    // forward of one layer from xin[layer]: leaves hs, xh, rs, hid, ys (and vs) in LDS; returns the layer's output in ys' place?
    // no: the output goes to `outp` ([c][p], 384 floats) when outp != nullptr
    auto forward_layer = [&](int layer, float* outp) {
        const float* w = a.packed + (size_t)layer * LAYER_FLOATS;
        float add = te;
        if (has_text) {
            float acc = 0.f;
            for (int kk = 0; kk < 32; ++kk) acc = fmaf(w[O_WV + (32 * wave + kk) * MC + lane], txt[32 * wave + kk], acc);
            red[wave][0][lane] = acc;
            __syncthreads();
            if (wave == 0) vs[lane] = w[O_BV + lane] + ((red[0][0][lane] + red[1][0][lane]) + (red[2][0][lane] + red[3][0][lane]));
            __syncthreads();
            acc = 0.f;
            for (int kk = 0; kk < 16; ++kk) acc = fmaf(w[O_WP + (16 * wave + kk) * MC + lane], vs[16 * wave + kk], acc);
            red[wave][1][lane] = acc;
            __syncthreads();
            add += w[O_BP + lane] + ((red[0][1][lane] + red[1][1][lane]) + (red[2][1][lane] + red[3][1][lane]));
            __syncthreads();
        }
        const float g = w[O_LG + lane], be = w[O_LB + lane];
        for (int p = wave; p < MP; p += 4) {
            const float v = xin[layer][lane * MP + p] + add;
            const float mean = wave_sum(v) * (1.0f / MC);
            const float dlt = v - mean;
            const float var = wave_sum(dlt * dlt) * (1.0f / MC);
            const float r = 1.0f / sqrtf(var + 1e-5f);
            xh[p][lane] = dlt * r;
            hs[p][lane] = dlt * r * g + be;
            if (lane == 0) rs[p] = r;
        }
        __syncthreads();
        {
            float acc[MP];
            const float b1 = w[O_B1 + tid];
#pragma unroll
            for (int p = 0; p < MP; ++p) acc[p] = b1;
            for (int c = 0; c < MC; ++c) {
                const float wv = w[O_W1 + c * MH + tid];
#pragma unroll
                for (int p = 0; p < MP; ++p) acc[p] = fmaf(wv, hs[p][c], acc[p]);
            }
#pragma unroll
            for (int p = 0; p < MP; ++p) hid[p][tid] = fmaxf(acc[p], 0.f);
        }
        __syncthreads();
        {
            float acc[MP];
#pragma unroll
            for (int p = 0; p < MP; ++p) acc[p] = 0.f;
            for (int jj = 0; jj < MC; ++jj) {
                const int j = 64 * wave + jj;
                const float wv = w[O_W2 + j * MC + lane];
#pragma unroll
                for (int p = 0; p < MP; ++p) acc[p] = fmaf(wv, hid[p][j], acc[p]);
            }
#pragma unroll
            for (int p = 0; p < MP; ++p) red[wave][p][lane] = acc[p];
        }
        __syncthreads();
        if (wave == 0) {
            const float b2 = w[O_B2 + lane];
#pragma unroll
            for (int p = 0; p < MP; ++p)
                ys[lane][p] = hs[p][lane] + (b2 + ((red[0][p][lane] + red[1][p][lane]) + (red[2][p][lane] + red[3][p][lane])));
        }
        __syncthreads();
        if (outp != nullptr) {
            float gv[MP], acc[MP];
#pragma unroll
            for (int p = 0; p < MP; ++p) { gv[p] = ys[lane][p]; acc[p] = 0.f; }
            const f32x4* w3 = reinterpret_cast<const f32x4*>(w + O_W3) + 2 * 64 * wave;
            const f32x4* w4 = reinterpret_cast<const f32x4*>(w + O_W4) + 2 * 64 * wave;
            for (int j = 0; j < 64; ++j) {
                const f32x4 a0 = w3[2 * j], a1 = w3[2 * j + 1];
                float hdn = a1.z;
                hdn = fmaf(a0.x, gv[0], hdn); hdn = fmaf(a0.y, gv[1], hdn); hdn = fmaf(a0.z, gv[2], hdn);
                hdn = fmaf(a0.w, gv[3], hdn); hdn = fmaf(a1.x, gv[4], hdn); hdn = fmaf(a1.y, gv[5], hdn);
                hdn = fmaxf(hdn, 0.f);
                const f32x4 c0 = w4[2 * j], c1 = w4[2 * j + 1];
                acc[0] = fmaf(c0.x, hdn, acc[0]); acc[1] = fmaf(c0.y, hdn, acc[1]); acc[2] = fmaf(c0.z, hdn, acc[2]);
                acc[3] = fmaf(c0.w, hdn, acc[3]); acc[4] = fmaf(c1.x, hdn, acc[4]); acc[5] = fmaf(c1.y, hdn, acc[5]);
            }
            __syncthreads();   // every wave has read red / ys
#pragma unroll
            for (int p = 0; p < MP; ++p) red[wave][p][lane] = acc[p];
            __syncthreads();
            for (int i = tid; i < MC * MP; i += 256) {
                const int c = i / MP, p = i - c * MP;
                outp[i] = w[O_B4 + p] + ((red[0][p][c] + red[1][p][c]) + (red[2][p][c] + red[3][p][c]));
            }
            __syncthreads();
        }
    };

    for (int layer = 0; layer + 1 < ML; ++layer) forward_layer(layer, xin[layer + 1]);
    for (int i = tid; i < MC * MP; i += 256) dcur[i / MP][i % MP] = a.dout[(size_t)b * MC * MP + i];
    __syncthreads();

    for (int layer = ML - 1; layer >= 0; --layer) {
        forward_layer(layer, nullptr);      // hs, xh, rs, hid, ys, vs of this layer
        const float* w = a.packed + (size_t)layer * LAYER_FLOATS;
        const t2s_mlp_layer_weights& wn = a.w.layer[layer];
        float* gp = a.part + ((size_t)b * ML + layer) * N_LAYER;
        // ---- position MLP, pass A: hidden unit j = tid over the 64 channels -> dW4[:, j], dW3[j, :], db3[j]
        {
            const f32x4 a0 = reinterpret_cast<const f32x4*>(w + O_W3)[2 * tid], a1 = reinterpret_cast<const f32x4*>(w + O_W3)[2 * tid + 1];
            const f32x4 c0 = reinterpret_cast<const f32x4*>(w + O_W4)[2 * tid], c1 = reinterpret_cast<const f32x4*>(w + O_W4)[2 * tid + 1];
            const float w3r[MP] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y}, w4c[MP] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y};
            float g4[MP], g3[MP], gb3 = 0.f;
#pragma unroll
            for (int p = 0; p < MP; ++p) g4[p] = g3[p] = 0.f;
            for (int c = 0; c < MC; ++c) {
                float hdn = a1.z, dh = 0.f;
#pragma unroll
                for (int p = 0; p < MP; ++p) hdn = fmaf(w3r[p], ys[c][p], hdn);
                hdn = fmaxf(hdn, 0.f);
#pragma unroll
                for (int p = 0; p < MP; ++p) {
                    const float d = dcur[c][p];
                    g4[p] = fmaf(d, hdn, g4[p]);
                    dh = fmaf(w4c[p], d, dh);
                }
                const float dpre = hdn > 0.f ? dh : 0.f;
#pragma unroll
                for (int p = 0; p < MP; ++p) g3[p] = fmaf(dpre, ys[c][p], g3[p]);
                gb3 += dpre;
            }
#pragma unroll
            for (int p = 0; p < MP; ++p) {
                gp[N_W3 + tid * MP + p] = g3[p];
                gp[N_W4 + p * MH + tid] = g4[p];
            }
            gp[N_B3 + tid] = gb3;
        }
        if (wave == 0) {    // db4[p] = sum_c dout[c][p]
#pragma unroll
            for (int p = 0; p < MP; ++p) {
                const float s = wave_sum(dcur[lane][p]);
                if (lane == 0) gp[N_B4 + p] = s;
            }
        }
        // ---- pass B: channel = lane, hidden units in quarters -> dy[p][c]
        {
            float gv[MP], dq[MP], acc[MP];
#pragma unroll
            for (int p = 0; p < MP; ++p) { gv[p] = ys[lane][p]; dq[p] = dcur[lane][p]; acc[p] = 0.f; }
            const f32x4* w3 = reinterpret_cast<const f32x4*>(w + O_W3) + 2 * 64 * wave;
            const f32x4* w4 = reinterpret_cast<const f32x4*>(w + O_W4) + 2 * 64 * wave;
            for (int j = 0; j < 64; ++j) {
                const f32x4 a0 = w3[2 * j], a1 = w3[2 * j + 1], c0 = w4[2 * j], c1 = w4[2 * j + 1];
                float hdn = a1.z;
                hdn = fmaf(a0.x, gv[0], hdn); hdn = fmaf(a0.y, gv[1], hdn); hdn = fmaf(a0.z, gv[2], hdn);
                hdn = fmaf(a0.w, gv[3], hdn); hdn = fmaf(a1.x, gv[4], hdn); hdn = fmaf(a1.y, gv[5], hdn);
                float dh = c0.x * dq[0];
                dh = fmaf(c0.y, dq[1], dh); dh = fmaf(c0.z, dq[2], dh); dh = fmaf(c0.w, dq[3], dh);
                dh = fmaf(c1.x, dq[4], dh); dh = fmaf(c1.y, dq[5], dh);
                const float dpre = hdn > 0.f ? dh : 0.f;
                acc[0] = fmaf(a0.x, dpre, acc[0]); acc[1] = fmaf(a0.y, dpre, acc[1]); acc[2] = fmaf(a0.z, dpre, acc[2]);
                acc[3] = fmaf(a0.w, dpre, acc[3]); acc[4] = fmaf(a1.x, dpre, acc[4]); acc[5] = fmaf(a1.y, dpre, acc[5]);
            }
#pragma unroll
            for (int p = 0; p < MP; ++p) red[wave][p][lane] = acc[p];
        }
        __syncthreads();
        if (wave == 0) {
            float sb2 = 0.f;
#pragma unroll
            for (int p = 0; p < MP; ++p) {
                const float d = (red[0][p][lane] + red[1][p][lane]) + (red[2][p][lane] + red[3][p][lane]);
                dys[p][lane] = d;
                sb2 += d;
            }
            gp[N_B2 + lane] = sb2;      // y = hs + W2 hid + b2
        }
        __syncthreads();
        // ---- channel MLP: dW2[c][j], dhid -> dp1, dW1[j][c], db1, d(hs) += W1^T dp1
        {
            float dyc[MP];
#pragma unroll
            for (int p = 0; p < MP; ++p) dyc[p] = dys[p][lane];
            for (int jj = 0; jj < MC; ++jj) {
                const int j = 64 * wave + jj;
                float s = 0.f;
#pragma unroll
                for (int p = 0; p < MP; ++p) s = fmaf(dyc[p], hid[p][j], s);
                gp[N_W2 + lane * MH + j] = s;
            }
        }
        {
            float acc[MP];
#pragma unroll
            for (int p = 0; p < MP; ++p) acc[p] = 0.f;
            for (int c = 0; c < MC; ++c) {
                const float wv = wn.mlp2_w[c * MH + tid];           // native (64,256): coalesced over j = tid
#pragma unroll
                for (int p = 0; p < MP; ++p) acc[p] = fmaf(wv, dys[p][c], acc[p]);
            }
            float sb1 = 0.f;
#pragma unroll
            for (int p = 0; p < MP; ++p) {
                const float d = hid[p][tid] > 0.f ? acc[p] : 0.f;
                dp1[p][tid] = d;
                sb1 += d;
            }
            gp[N_B1 + tid] = sb1;
            for (int c = 0; c < MC; ++c) {
                float s = 0.f;
#pragma unroll
                for (int p = 0; p < MP; ++p) s = fmaf(acc[p] * (hid[p][tid] > 0.f ? 1.f : 0.f), hs[p][c], s);
                gp[N_W1 + tid * MC + c] = s;
            }
        }
        __syncthreads();
        {
            float acc[MP];
#pragma unroll
            for (int p = 0; p < MP; ++p) acc[p] = 0.f;
            for (int jj = 0; jj < MC; ++jj) {
                const int j = 64 * wave + jj;
                const float wv = wn.mlp0_w[j * MC + lane];          // native (256,64): coalesced over c = lane
#pragma unroll
                for (int p = 0; p < MP; ++p) acc[p] = fmaf(wv, dp1[p][j], acc[p]);
            }
            __syncthreads();   // red's previous readers (wave 0 above) are done
#pragma unroll
            for (int p = 0; p < MP; ++p) red[wave][p][lane] = acc[p];
        }
        __syncthreads();
        // ---- LayerNorm backward per position (wave w: positions w, w + 4); dgamma / dbeta summed over the positions afterwards
        {
            const float g = w[O_LG + lane];
            for (int p = wave; p < MP; p += 4) {
                const float dh = dys[p][lane] + ((red[0][p][lane] + red[1][p][lane]) + (red[2][p][lane] + red[3][p][lane]));
                const float xhat = xh[p][lane];
                const float dxh = dh * g;
                const float m1 = wave_sum(dxh) * (1.0f / MC);
                const float m2 = wave_sum(dxh * xhat) * (1.0f / MC);
                const float du = rs[p] * (dxh - m1 - xhat * m2);
                hs[p][lane] = dh;                 // (hs is dead now: keep dh for dgamma / dbeta)
                dp1[p][lane] = du;                // (dp1 rows are free again: du[p][c] in their first 64 columns)
            }
        }
        __syncthreads();
        if (wave == 0) {
            float sg = 0.f, sb = 0.f, sa = 0.f;
#pragma unroll
            for (int p = 0; p < MP; ++p) {
                sg = fmaf(hs[p][lane], xh[p][lane], sg);
                sb += hs[p][lane];
                sa += dp1[p][lane];
            }
            gp[N_LG + lane] = sg;
            gp[N_LB + lane] = sb;
            dadd[lane] = sa;                      // gradient of the broadcast row te + a
        }
        __syncthreads();
        // ---- cross attention: a = proj(value(text)) broadcast over the positions
        if (has_text) {
            const float da = dadd[lane];
            if (wave == 0) gp[N_BP + lane] = da;
            for (int kk = 0; kk < 16; ++kk) gp[N_WP + lane * MC + 16 * wave + kk] = da * vs[16 * wave + kk];
            for (int kk = 0; kk < 16; ++kk) {      // dvs[k] = sum_c proj.weight[c][k] da[c]: packed [k][c], lane = c
                const int k = 16 * wave + kk;
                const float s = wave_sum(w[O_WP + k * MC + lane] * da);
                if (lane == 0) dvs[k] = s;
            }
            __syncthreads();
            const float dv = dvs[lane];
            if (wave == 0) gp[N_BV + lane] = dv;
            for (int tt = 0; tt < 32; ++tt) gp[N_WV + lane * MT + 32 * wave + tt] = dv * txt[32 * wave + tt];
        } else {
            for (int i = tid; i < N_LG; i += 256) gp[i] = 0.f;     // value / proj parameters: untouched by this forward
        }
        if (tid < 2) gp[N_B4 + MP + tid] = 0.f;
        // ---- the gradient of the layer's input is du, as [c][p]
        __syncthreads();
        for (int i = tid; i < MC * MP; i += 256) dcur[i / MP][i % MP] = dp1[i % MP][i / MP];
        __syncthreads();
    }
    if (a.dx != nullptr)
        for (int i = tid; i < MC * MP; i += 256) a.dx[(size_t)b * MC * MP + i] = dcur[i / MP][i % MP];
}

struct ReduceArgs {
    t2s_mlp_grads g;
    const float* part;
    int B;
};

// grad element = sum over the series, in series order; one thread per element of one layer's N_LAYER block
__global__ __launch_bounds__(256) void mlp_grad_reduce_kernel(const ReduceArgs a) {
    const int layer = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N_LAYER) return;
    constexpr int starts[15] = {N_WV, N_BV, N_WP, N_BP, N_LG, N_LB, N_W1, N_B1, N_W2, N_B2, N_W3, N_B3, N_W4, N_B4, N_LAYER};
    float* const* dst = reinterpret_cast<float* const*>(&a.g.layer[layer]);
    int seg = 0;
#pragma unroll
    for (int k = 1; k < 14; ++k) seg += i >= starts[k];
    if (seg == 13 && i - N_B4 >= MP) return;     // padding behind mlp2.2.bias
    const float* p = a.part + (size_t)layer * N_LAYER + i;
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) s += p[(size_t)b * ML * N_LAYER];
    dst[seg][i - starts[seg]] = s;
}

}  // namespace
}  // namespace t2s

extern "C" int t2s_mlp_pack(const t2s_mlp_weights* w, float* packed, void* stream) {
    using namespace t2s;
    T2S_REQUIRE(w && packed, "t2s_mlp_pack: null argument");
    static const struct { size_t off; size_t floats; const char* name; } items[] = {
        {offsetof(t2s_mlp_layer_weights, value_w), (size_t)MC * MT, "cross_attn.value.weight"},
        {offsetof(t2s_mlp_layer_weights, value_b), MC, "cross_attn.value.bias"},
        {offsetof(t2s_mlp_layer_weights, proj_w), (size_t)MC * MC, "cross_attn.proj.weight"},
        {offsetof(t2s_mlp_layer_weights, proj_b), MC, "cross_attn.proj.bias"},
        {offsetof(t2s_mlp_layer_weights, norm2_w), MC, "norm2.weight"},
        {offsetof(t2s_mlp_layer_weights, norm2_b), MC, "norm2.bias"},
        {offsetof(t2s_mlp_layer_weights, mlp0_w), (size_t)MH * MC, "mlp.0.weight"},
        {offsetof(t2s_mlp_layer_weights, mlp0_b), MH, "mlp.0.bias"},
        {offsetof(t2s_mlp_layer_weights, mlp2_w), (size_t)MC * MH, "mlp.2.weight"},
        {offsetof(t2s_mlp_layer_weights, mlp2_b), MC, "mlp.2.bias"},
        {offsetof(t2s_mlp_layer_weights, pos0_w), (size_t)MH * MP, "mlp2.0.weight"},
        {offsetof(t2s_mlp_layer_weights, pos0_b), MH, "mlp2.0.bias"},
        {offsetof(t2s_mlp_layer_weights, pos2_w), (size_t)MP * MH, "mlp2.2.weight"},
        {offsetof(t2s_mlp_layer_weights, pos2_b), MP, "mlp2.2.bias"},
    };
    for (int l = 0; l < ML; ++l)
        for (const auto& it : items) {
            const float* p = *reinterpret_cast<const float* const*>(reinterpret_cast<const char*>(&w->layer[l]) + it.off);
            T2S_REQUIRE(p != nullptr, "t2s_mlp_pack: layers.%d.%s is null", l, it.name);
            char what[96];
            snprintf(what, sizeof what, "t2s_mlp_pack: layers.%d.%s", l, it.name);
            if (int rc = check_device_extent(p, it.floats * sizeof(float), what)) return rc;
        }
    if (int rc = check_device_extent(packed, (size_t)T2S_MLP_PACKED_FLOATS * sizeof(float), "t2s_mlp_pack: packed")) return rc;
    PackArgs a;
    a.w = *w;
    a.dst = packed;
    mlp_pack_kernel<<<ML, 256, 0, (hipStream_t)stream>>>(a);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

extern "C" int t2s_mlp_forward(const float* packed, const float* x, const float* t, const float* freqs, const float* text,
                               float* out, int B, void* stream) {
    using namespace t2s;
    T2S_REQUIRE(packed && x && t && freqs && out, "t2s_mlp_forward: null argument");
    T2S_REQUIRE(B >= 0, "t2s_mlp_forward: B = %d", B);
    if (B == 0) return T2S_OK;
    if (int rc = check_device_extent(packed, (size_t)T2S_MLP_PACKED_FLOATS * sizeof(float), "t2s_mlp_forward: packed")) return rc;
    if (int rc = check_device_extent(x, (size_t)B * MC * MP * sizeof(float), "t2s_mlp_forward: x")) return rc;
    if (int rc = check_device_extent(t, (size_t)B * sizeof(float), "t2s_mlp_forward: t")) return rc;
    if (int rc = check_device_extent(freqs, (size_t)(MC / 2) * sizeof(float), "t2s_mlp_forward: freqs")) return rc;
    if (text)
        if (int rc = check_device_extent(text, (size_t)B * MT * sizeof(float), "t2s_mlp_forward: text")) return rc;
    if (int rc = check_device_extent(out, (size_t)B * MC * MP * sizeof(float), "t2s_mlp_forward: out")) return rc;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    if (B <= n_cu)      // one round of workgroups either way: the latency form (same arithmetic, same bits)
        mlp_forward_kernel<true><<<B, 256, 0, (hipStream_t)stream>>>(packed, x, t, freqs, text, out);
    else
        mlp_forward_kernel<false><<<B, 256, 0, (hipStream_t)stream>>>(packed, x, t, freqs, text, out);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

extern "C" int t2s_mlp_backward(const t2s_mlp_weights* w, const float* packed, const float* x, const float* t, const float* freqs,
                                const float* text, const float* dout, float* dx, const t2s_mlp_grads* grads, float* scratch,
                                uint64_t scratch_floats, int B, void* stream) {
    using namespace t2s;
    T2S_REQUIRE(w && packed && x && t && freqs && dout && grads && scratch, "t2s_mlp_backward: null argument");
    T2S_REQUIRE(B > 0, "t2s_mlp_backward: B = %d", B);
    T2S_REQUIRE(scratch_floats >= (uint64_t)B * T2S_MLP_GRAD_PART_FLOATS, "t2s_mlp_backward: scratch holds %llu floats, B x %d needed",
                (unsigned long long)scratch_floats, T2S_MLP_GRAD_PART_FLOATS);
    if (int rc = check_device_extent(scratch, (size_t)B * T2S_MLP_GRAD_PART_FLOATS * sizeof(float), "t2s_mlp_backward: scratch")) return rc;
    if (int rc = check_device_extent(packed, (size_t)T2S_MLP_PACKED_FLOATS * sizeof(float), "t2s_mlp_backward: packed")) return rc;
    if (int rc = check_device_extent(x, (size_t)B * MC * MP * sizeof(float), "t2s_mlp_backward: x")) return rc;
    if (int rc = check_device_extent(dout, (size_t)B * MC * MP * sizeof(float), "t2s_mlp_backward: dout")) return rc;
    if (dx)
        if (int rc = check_device_extent(dx, (size_t)B * MC * MP * sizeof(float), "t2s_mlp_backward: dx")) return rc;
    static const size_t gsz[14] = {(size_t)MC * MT, MC, (size_t)MC * MC, MC, MC, MC, (size_t)MH * MC, MH, (size_t)MC * MH, MC, (size_t)MH * MP, MH,
                                   (size_t)MP * MH, MP};
    for (int l = 0; l < ML; ++l) {
        float* const* gp = reinterpret_cast<float* const*>(&grads->layer[l]);
        const float* const* wp = reinterpret_cast<const float* const*>(&w->layer[l]);
        for (int k = 0; k < 14; ++k) {
            T2S_REQUIRE(gp[k] && wp[k], "t2s_mlp_backward: layers.%d tensor %d is null", l, k);
            if (int rc = check_device_extent(gp[k], gsz[k] * sizeof(float), "t2s_mlp_backward: a gradient tensor")) return rc;
        }
        if (int rc = check_device_extent(wp[6], gsz[6] * sizeof(float), "t2s_mlp_backward: mlp.0.weight")) return rc;
        if (int rc = check_device_extent(wp[8], gsz[8] * sizeof(float), "t2s_mlp_backward: mlp.2.weight")) return rc;
    }
    hipStream_t st = (hipStream_t)stream;
    BwdArgs a;
    a.w = *w; a.packed = packed; a.x = x; a.t = t; a.freqs = freqs; a.text = text; a.dout = dout; a.dx = dx; a.part = scratch;
    mlp_backward_kernel<<<B, 256, 0, st>>>(a);
    T2S_LAUNCH_CHECK();
    ReduceArgs r;
    r.g = *grads; r.part = scratch; r.B = B;
    mlp_grad_reduce_kernel<<<dim3((N_LAYER + 255) / 256, ML), 256, 0, st>>>(r);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}
