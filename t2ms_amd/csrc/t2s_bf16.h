// bf16-operand building blocks of the mixed-precision TRAINING path (BASELINE config 4:
// "DiT training (train.py) bf16"): v_mfma_f32_32x32x16_bf16 with fp32 accumulation, fp32 master
// weights / residual stream / LayerNorm and softmax statistics / gradients, bf16 saved activations.
//
//   bgemm_kernel   out^T tile = W . A^T      tokens on the lanes (32 per wave), weights pre-packed
//                                            as the A operand; prologues LN+modulate / GELU,
//                                            epilogues bf16 rows / q,k,v heads / gelu-backward
//   wgrad16_kernel dW = dY^T X, db = colsum  contraction over token rows; both operands read
//                                            column-wise from row-major LDS tiles with
//                                            ds_read_b64_tr_b16; bias gradient from a ones-MFMA
//
// With bf16 MFMA 16x faster than the fp32 form every kernel of the step is HBM-bound, so the
// design rules here are bytes and coalescing, not MFMA occupancy: each activation is written once
// as bf16 rows, read as 16-B lane fragments, and the waves of a workgroup are fully independent
// (no LDS, no barrier) in bgemm.
#pragma once
#include "t2s_gemm.h"

// see t2s_x3.h: packed fp32 VALU instructions serialise with bf16 MFMAs on gfx950
#if defined(__HIP_DEVICE_COMPILE__)
#define T2S_NO_PK_F32 __attribute__((target("no-packed-fp32-ops")))
#else
#define T2S_NO_PK_F32
#endif

namespace t2s {

// Consecutive launches of the training step walk their tiles in OPPOSITE directions (bgemm token tiles, weight-gradient row
// slabs, attention heads): a consumer then starts on what its producer wrote -- or the previous reader of the same tensor
// read -- last, which is what the 256 MB Infinity Cache still holds (a tensor is 142-283 MB).  Results do not depend on
// the order (tiles are independent, partial-sum slots keep their slab index).  T2S_TILE_FLIP=0 switches it off for A/B:
// 11.7-11.8 vs 12.0-12.1 ms per step on one box, weight gradients 2.03 -> 1.91 ms.
// The parity counter is per host thread (a handle is driven by one thread at a time; two models training from two
// threads keep their own patterns) and restarts at the top of every training forward / backward.
inline thread_local unsigned g_tile_flip = 0;
// at the top of the training forward / backward: the same pattern every step, starting DESCENDING (the patchify and
// final-layer kernels in front of the first GEMM / weight gradient write ascending)
inline void reset_tile_dir() { g_tile_flip = 1; }
inline int next_tile_dir() {
    static const int enabled = getenv("T2S_TILE_FLIP") ? atoi(getenv("T2S_TILE_FLIP")) : 1;   // read once
    return enabled ? (int)(g_tile_flip++ & 1) : 0;
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// D(32x32) += A(32x16) * B(16x32); lane l (i = l&31, h = l>>5) supplies A[i][8h + j] and
// B[8h + j][i], j = 0..7; the result layout equals the fp32 form's (t2s_common.h: acc_row).
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// The same with the result in registers of its own (vdst != src2): c survives, so a loop-invariant C operand -- the
// negated softmax reference, log-sum-exp or D_i broadcast over an accumulator -- subtracts for free, block after block.
// (The builtin ties vdst to src2 and the compiler would copy c first: 16 v_mov per use.)  Follow it with builtin MFMAs that
// accumulate in place on the result before any VALU instruction reads it.
__device__ __forceinline__ f32x16 mfma16_from(bf16x8 a, bf16x8 b, const f32x16& c) {
    f32x16 d;
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// Attention of the bf16 training path keeps q PRE-SCALED by log2(e) / sqrt(head_dim): the qkv GEMM's epilogue applies
// it before the bf16 rounding, so q.k is the score in the log2 domain and no kernel multiplies scores again.
constexpr float ATT_QS = 0.17677669529663687f * 1.4426950408889634f;

__device__ __forceinline__ bf16x4 pack4(f32x4 v) { return __builtin_convertvector(v, bf16x4); }
__device__ __forceinline__ f32x4 unpack4(bf16x4 v) { return __builtin_convertvector(v, f32x4); }
__device__ __forceinline__ bf16x8 pack8(f32x4 lo, f32x4 hi) {
    const f32x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_convertvector(v, bf16x8);
}
__device__ __forceinline__ f32x8 unpack8(bf16x8 v) { return __builtin_convertvector(v, f32x8); }

// Registers 8s..8s+7 of a 32x32 accumulator as the operand of k-step s of a following MFMA that
// sums over the accumulator's ROW index.  Element j of lane half h is row 16s + 8(j>>2) + 4h + (j&3)
// ("permuted k order"): the other operand must use the same order (tr_frag below does).
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& c, int s) {
    const f32x8 v = {c[8 * s + 0], c[8 * s + 1], c[8 * s + 2], c[8 * s + 3],
                     c[8 * s + 4], c[8 * s + 5], c[8 * s + 6], c[8 * s + 7]};
    return __builtin_convertvector(v, bf16x8);
}

// ds_read_b64_tr_b16: per group of 16 lanes a block of 4 rows x 16 columns of 16-bit elements is
// read from LDS and delivered column-major: lane 4q+p of the group supplies the address of row q,
// columns 4p..4p+3 (8 bytes, 8-byte aligned); lane i of the group receives column i, row q in
// element q.  EXEC must be all ones.
__device__ __forceinline__ s16x4 lds_tr16(const void* lds_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_addr));
}
__device__ __forceinline__ bf16x8 join_tr(s16x4 lo, s16x4 hi) {
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// A-operand packing of a linear weight W (N,K) row-major (the GEMM is x @ W^T): n-tile nt (32
// outputs), k-step s (16 inputs): lane l = 32h + (n & 31) owns the 8 bf16 W[n][16s + 8h + 0..7],
// stored at packed[(nt * (K/16) + s) * 64 + l] -> one wave-level load = 1 KiB contiguous.
__host__ __device__ inline size_t packed16_index(int n, int k, int K) {
    const int nt = n >> 5, i = n & 31, s = k >> 4, h = (k >> 3) & 1, j = k & 7;
    return ((((size_t)nt * (K >> 4) + s) * 64) + (h * 32 + i)) * 8 + j;
}

// W fp32 (N,K) -> packed16 of W (transpose = 0) or of W^T, a (K,N) weight (transpose = 1)
static __global__ void pack16_kernel(const float* __restrict__ W, __bf16* __restrict__ P, int N, int K, int transpose) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * K) return;
    const int n = idx / K, k = idx - n * K;
    const size_t dst = transpose ? packed16_index(k, n, N) : packed16_index(n, k, K);
    P[dst] = (__bf16)W[idx];
}

// ------------------------------------------------------------------------------------------ bgemm
enum { BPRO_BF16 = 0, BPRO_LN = 1, BPRO_GELU = 2, BPRO_LN_RES = 3 };
enum { BEPI_BF16 = 0, BEPI_QKV = 1, BEPI_GELUBWD = 2, BEPI_LNBWD = 3 };

struct BGemmArgs {
    const void* A;        // BPRO_LN: float (M,128) residual stream; otherwise bf16 (M,K)
    const bf16x8* Wp;     // packed16 weights (N/32, K/16, 64)
    const float* bias;    // (N) or NULL
    __bf16* out;          // BEPI_BF16 / BEPI_GELUBWD: (M,N)
    int M;                // rows, a multiple of 32
    int N;                // outputs, a multiple of 32
    const float* mod;     // BPRO_LN: (S, MODROW) adaLN table
    int shift_off;
    int scale_off;
    __bf16* save_A;       // optional (M,K): the prologue-transformed rows (saved for the weight gradient)
    const __bf16* aux;    // BEPI_GELUBWD: pre-activation u (M,N); out = acc * gelu'(u)
    const __bf16* res;    // BPRO_LN_RES: the branch output (M,128) whose gated residual add produces this layer's input:
    int gate_off;         //   x = A + mod[seq][gate_off + c] * res; x is written to x_out (fp32 (M,128)) and then
    float* x_out;         //   LayerNorm-modulated like BPRO_LN -- the stand-alone gate/residual kernel fused away
    __bf16* q;            // BEPI_QKV destinations, each (S*4, 480, 32)
    __bf16* k;
    __bf16* v;
    // BEPI_LNBWD (N = 128): the product is da = grad wrt modulate(LN(ln_x)); the epilogue is the LayerNorm + modulate
    // backward of ln_mod_bwd_kernel (t2s_train.hip) on the accumulators: dx += rstd (dn - mean(dn) - n mean(dn n)),
    // dn = da (1 + scale), and -- when res != NULL -- the gate backward of the branch differentiated next on the dx just
    // produced: out = mod[gate_off] * dx (bf16), dgate = sum_tok dx * res.  The three per-sequence column sums
    // (dshift = sum da, dscale = sum da n, dgate) leave as one partial row per 32-token tile: colpart (M/32, 3, 128),
    // added over the 15 tiles of a sequence, in tile order, by lnbwd_colsum_reduce_kernel.
    const float* ln_x;
    float* dx;
    float* colpart;
    int reverse;          // tile order: 0 ascending, 1 descending
};

// Weights-stationary streaming GEMM: one persistent workgroup of 12 waves per CU keeps the whole
// packed weight (<= 96 KB bf16) and the bias in LDS; every wave then streams 32-token tiles:
// rows from HBM -> registers (prologue) -> N/32 x K/16 MFMAs with the A operand read from LDS
// (lane-linear 16-byte fragments: conflict-free ds_read_b128) -> epilogue stores.  No barrier after
// the weight load; HBM latency is hidden by the three waves per SIMD.
constexpr int BG_THREADS = 768;
// The LayerNorm-backward epilogue holds a 32 x 128 fp32 product AND the matching LayerNorm input per wave (2 x 64
// registers per lane before any temporaries): 8 waves per workgroup (256 registers per lane) instead of 12 (168, where
// hipcc spilled 94 of them to scratch).
constexpr int bg_threads(int epi) { return epi == 3 ? 512 : 768; }
constexpr int BG_STAGE_STRIDE = 144;                     // bytes per staged token row (128 data + 16 pad)
constexpr int BG_STAGE_BYTES = 32 * BG_STAGE_STRIDE;     // per wave

// BEPI_LNBWD on one 32-token tile: acc = da (lane = token i, half h; register 4g+e of n-tile nt = feature 32nt + 8g + 4h + e).
// The pointers are __restrict__ ON PURPOSE although dx_rows and dx_tile are the same tensor: no element is loaded after
// it was stored, and without the promise every load inside the n-tile loops waits behind the previous n-tile's stores
// (measured: 36 us per tile and wave, a dozen serialized HBM round trips).
//   x_rows / dx_rows / res_rows: this lane's token row + 4h;  dx_tile / out_tile: row 0 of the tile;  mrow: LDS, scale
//   row (+128: next gate row) + 4h;  stage: the wave's 32 x 144-byte staging tile, used in program order for the column
//   sums over the 32 tokens (each lane writes its row pieces, then adds one column over 16 rows + the other half), the
//   fp32 dx rows and the bf16 out rows, which leave as whole 128-byte lines.
__device__ __forceinline__ void lnbwd_epilogue(f32x16 (&acc)[4], const float* __restrict__ x_rows, const float* __restrict__ dx_rows,
                                               float* __restrict__ dx_tile, const __bf16* __restrict__ res_rows,
                                               __bf16* __restrict__ out_tile, float* __restrict__ cp, const float* mrow,
                                               char* stage, int lane) {
    const int h = lane >> 5, i = lane & 31;
    f32x4 nv[4][4];
    float s = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            nv[nt][g] = *reinterpret_cast<const f32x4*>(x_rows + nt * 32 + 8 * g);
            s += (nv[nt][g].x + nv[nt][g].y) + (nv[nt][g].z + nv[nt][g].w);
        }
    s += xhalf(s);
    const float mean = s * (1.0f / 128.0f);
    float ss = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            nv[nt][g] -= mean;
            ss += (nv[nt][g].x * nv[nt][g].x + nv[nt][g].y * nv[nt][g].y) + (nv[nt][g].z * nv[nt][g].z + nv[nt][g].w * nv[nt][g].w);
        }
    ss += xhalf(ss);
    const float rstd = rsqrtf(ss * (1.0f / 128.0f) + 1e-6f);
    float* srow = reinterpret_cast<float*>(stage + i * BG_STAGE_STRIDE) + 4 * h;
    const char* scol = stage + (16 * h) * BG_STAGE_STRIDE + 4 * i;
    auto colsum = [&](float* dst) {
        float cs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) cs += *reinterpret_cast<const float*>(scol + r * BG_STAGE_STRIDE);
        cs += xhalf(cs);
        if (h == 0) dst[i] = cs;
    };
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            nv[nt][g] = nv[nt][g] * rstd;
            const f32x4 da = {acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]};
            *reinterpret_cast<f32x4*>(srow + 8 * g) = da;
        }
        colsum(cp + nt * 32);                                   // dshift
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 da = {acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]};
            *reinterpret_cast<f32x4*>(srow + 8 * g) = da * nv[nt][g];
        }
        colsum(cp + 128 + nt * 32);                             // dscale
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(mrow + nt * 32 + 8 * g);
            const f32x4 da = {acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]};
            const f32x4 dn = da * (1.0f + sc);
            const f32x4 n = nv[nt][g];
            m1 += (dn.x + dn.y) + (dn.z + dn.w);
            m2 += (dn.x * n.x + dn.y * n.y) + (dn.z * n.z + dn.w * n.w);
            acc[nt][4 * g] = dn.x; acc[nt][4 * g + 1] = dn.y; acc[nt][4 * g + 2] = dn.z; acc[nt][4 * g + 3] = dn.w;
        }
    }
    m1 += xhalf(m1);
    m2 += xhalf(m2);
    m1 *= (1.0f / 128.0f);
    m2 *= (1.0f / 128.0f);
    bf16x4 dpk[4];   // out pieces of the even n-tile, staged together with the odd one's (64 + 64 B = one line per row)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        f32x4 rr[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 dn = {acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]};
            rr[g] = *reinterpret_cast<const f32x4*>(dx_rows + nt * 32 + 8 * g) + (dn - m1 - nv[nt][g] * m2) * rstd;
            *reinterpret_cast<f32x4*>(srow + 8 * g) = rr[g];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {                           // 8 rows x 128 B per store
            const int r = 8 * u + (lane >> 3), c16 = lane & 7;
            *reinterpret_cast<f32x4*>(dx_tile + (size_t)r * 128 + nt * 32 + c16 * 4) =
                *reinterpret_cast<const f32x4*>(stage + r * BG_STAGE_STRIDE + c16 * 16);
        }
        if (res_rows != nullptr) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 pv = unpack4(*reinterpret_cast<const bf16x4*>(res_rows + nt * 32 + 8 * g));
                *reinterpret_cast<f32x4*>(srow + 8 * g) = rr[g] * pv;
            }
            colsum(cp + 256 + nt * 32);                         // dgate of the next branch
            bf16x4 dp4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 gn = *reinterpret_cast<const f32x4*>(mrow + 128 + nt * 32 + 8 * g);
                dp4[g] = pack4(gn * rr[g]);
            }
            if (!(nt & 1)) {
#pragma unroll
                for (int g = 0; g < 4; ++g) dpk[g] = dp4[g];
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    *reinterpret_cast<bf16x4*>(stage + i * BG_STAGE_STRIDE + 16 * g + 8 * h) = dpk[g];
                    *reinterpret_cast<bf16x4*>(stage + i * BG_STAGE_STRIDE + 64 + 16 * g + 8 * h) = dp4[g];
                }
                char* dst = reinterpret_cast<char*>(out_tile + (nt - 1) * 32);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = 8 * u + (lane >> 3), c16 = lane & 7;
                    *reinterpret_cast<bf16x8*>(dst + (size_t)r * 256 + c16 * 16) = *reinterpret_cast<const bf16x8*>(stage + r * BG_STAGE_STRIDE + c16 * 16);
                }
            }
        }
    }
}

// -DT2S_CHAIN_BOUND (tools/chain_bound.sh; results INVALID, timing only): the forward GEMMs stop reading exactly the tensors
// a register-resident forward chain per half block (proj -> gate/res -> LN -> fc1 -> GELU -> fc2 -> gate/res -> LN -> qkv)
// would keep in registers -- p in fc1's prologue, u in fc2's, x_mid and f in the next qkv's -- while every byte the backward
// pass needs is still WRITTEN.  Whatever such a chain costs on top (weights through a ring or from L2, fewer waves per SIMD),
// it cannot beat these launches by more than their four launch boundaries per block: an UPPER bound of its gain, measured.
#ifdef T2S_CHAIN_BOUND
#define T2S_BOUND(...) __VA_ARGS__
#else
#define T2S_BOUND(...)
#endif

template <int K, int N, int PRO, int EPI>
__global__ __launch_bounds__(bg_threads(EPI)) T2S_NO_PK_F32 void bgemm_kernel(const BGemmArgs a) {
    constexpr int KS = K / 16, NT = N / 32, THREADS = bg_threads(EPI);
    static_assert((PRO != BPRO_LN && PRO != BPRO_LN_RES) || K == 128, "LayerNorm prologue is over d_model = 128");
    extern __shared__ __attribute__((aligned(16))) char wl[];
    bf16x8* wlds = reinterpret_cast<bf16x8*>(wl);
    float* blds = reinterpret_cast<float*>(wl + (size_t)N * K * 2);
    // per-wave output staging tile: 32 token rows x (128 + 16) bytes.  The accumulator hands each lane 8-byte
    // pieces of a row; written straight to HBM they arrive as partial lines (measured 2.6 TB/s of stores against
    // 5.6 TB/s of loads), so two 32-column n-tiles are gathered in LDS and leave as whole 128-byte lines, 16 bytes per
    // lane (q / k / v head tiles: 64-byte rows that are contiguous across tokens, one 2 KiB run per n-tile).
    char* stage = wl + (size_t)N * K * 2 + (size_t)N * 4 + (size_t)(threadIdx.x >> 6) * BG_STAGE_BYTES;
    {   // the weights into LDS: every load first, then every ds_write -- as a load -> store loop hipcc waits vmcnt(0) in front of
        // each store, 3 ... 8 serial L2 round trips per workgroup and launch (2-5 us of launches that take 50-280 us)
        constexpr int ITER = (N * K / 8 + THREADS - 1) / THREADS;
        bf16x8 wt[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int c = threadIdx.x + it * THREADS;
            if (c < N * K / 8) wt[it] = a.Wp[c];
        }
        float bt = 0.f;
        if (threadIdx.x < N && a.bias != nullptr) bt = a.bias[threadIdx.x];
        static_assert(N <= THREADS, "one bias element per thread");
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int c = threadIdx.x + it * THREADS;
            if (c < N * K / 8) wlds[c] = wt[it];
        }
        if (threadIdx.x < N) blds[threadIdx.x] = bt;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, i = lane & 31;
    const int n_tiles = a.M >> 5;
    for (int tile_lin = blockIdx.x * (THREADS / 64) + wave; tile_lin < n_tiles; tile_lin += gridDim.x * (THREADS / 64)) {
        const int tile = a.reverse ? n_tiles - 1 - tile_lin : tile_lin;
        const size_t row = (size_t)tile * 32 + i;   // a 32-token tile never straddles a sequence (480 = 15 x 32)
        const int seq = tile / (NTOK / 32);
        const int tok = (tile - seq * (NTOK / 32)) * 32 + i;

        // ---- B operand: this lane's token, k = 16s + 8h + 0..7
        bf16x8 xf[KS];
        if constexpr (PRO == BPRO_LN || PRO == BPRO_LN_RES) {
            const float* xr = reinterpret_cast<const float*>(a.A) + row * K + 8 * h;
            f32x4 v[KS][2];
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < KS; ++t) {
#ifdef T2S_CHAIN_BOUND
                if constexpr (PRO == BPRO_LN_RES && N == 384) {        // the next block's qkv: x_mid would be in registers
                    const float c = (float)((lane + 3 * t) & 15) * 0.125f;
                    v[t][0] = f32x4{c, -c, c + 1.f, 0.5f};
                    v[t][1] = f32x4{-c, c, 0.25f, c - 1.f};
                    (void)xr;
                    continue;
                }
#endif
                v[t][0] = *reinterpret_cast<const f32x4*>(xr + 16 * t);
                v[t][1] = *reinterpret_cast<const f32x4*>(xr + 16 * t + 4);
            }
            if constexpr (PRO == BPRO_LN_RES) {
                // x = x_prev + gate * branch (the expression of gate_res_kernel, same operation order), written back as
                // whole 128-byte lines through the wave's staging tile: 32 features (two k-steps) of the 32 rows per pass
                const __bf16* rr = a.res + row * K + 8 * h;
                const float* grow = a.mod + (size_t)seq * MODROW + a.gate_off + 8 * h;
                float* xo = a.x_out + (size_t)tile * 32 * K;
#pragma unroll
                for (int t = 0; t < KS; ++t) {
#ifdef T2S_CHAIN_BOUND
                    (void)rr;                                           // p (fc1) / f (qkv): the chain has them in registers
                    const f32x8 b = {0.5f, -0.5f, 0.25f, 1.f, -1.f, 0.125f, 0.f, 0.75f};
#else
                    const f32x8 b = unpack8(*reinterpret_cast<const bf16x8*>(rr + 16 * t));
#endif
                    const f32x4 g0 = *reinterpret_cast<const f32x4*>(grow + 16 * t);
                    const f32x4 g1 = *reinterpret_cast<const f32x4*>(grow + 16 * t + 4);
                    const f32x4 b0 = {b[0], b[1], b[2], b[3]}, b1 = {b[4], b[5], b[6], b[7]};
                    v[t][0] = v[t][0] + g0 * b0;
                    v[t][1] = v[t][1] + g1 * b1;
                    char* sp = stage + i * BG_STAGE_STRIDE + (t & 1) * 64 + 32 * h;
                    *reinterpret_cast<f32x4*>(sp) = v[t][0];
                    *reinterpret_cast<f32x4*>(sp + 16) = v[t][1];
                    if (t & 1) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int r = 8 * u + (lane >> 3), c16 = lane & 7;
                            *reinterpret_cast<f32x4*>(xo + (size_t)r * K + (t >> 1) * 32 + c16 * 4) =
                                *reinterpret_cast<const f32x4*>(stage + r * BG_STAGE_STRIDE + c16 * 16);
                        }
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < KS; ++t)
                s += ((v[t][0].x + v[t][0].y) + (v[t][0].z + v[t][0].w)) + ((v[t][1].x + v[t][1].y) + (v[t][1].z + v[t][1].w));
            s += xhalf(s);
            const float mean = s * (1.0f / 128.0f);
            float ss = 0.f;
#pragma unroll
            for (int t = 0; t < KS; ++t) {
                v[t][0] -= mean;
                v[t][1] -= mean;
                ss += ((v[t][0].x * v[t][0].x + v[t][0].y * v[t][0].y) + (v[t][0].z * v[t][0].z + v[t][0].w * v[t][0].w)) +
                      ((v[t][1].x * v[t][1].x + v[t][1].y * v[t][1].y) + (v[t][1].z * v[t][1].z + v[t][1].w * v[t][1].w));
            }
            ss += xhalf(ss);
            const float rstd = rsqrtf(ss * (1.0f / 128.0f) + 1e-6f);
            const float* mrow = a.mod + (size_t)seq * MODROW + 8 * h;
#pragma unroll
            for (int t = 0; t < KS; ++t) {
                const f32x4 sc0 = *reinterpret_cast<const f32x4*>(mrow + a.scale_off + 16 * t);
                const f32x4 sc1 = *reinterpret_cast<const f32x4*>(mrow + a.scale_off + 16 * t + 4);
                const f32x4 sh0 = *reinterpret_cast<const f32x4*>(mrow + a.shift_off + 16 * t);
                const f32x4 sh1 = *reinterpret_cast<const f32x4*>(mrow + a.shift_off + 16 * t + 4);
                xf[t] = pack8((v[t][0] * rstd) * (1.0f + sc0) + sh0, (v[t][1] * rstd) * (1.0f + sc1) + sh1);
                if (a.save_A != nullptr) {
                    // saved for the weight gradient: through the staging tile, four k-steps = 64 features = one 128-byte
                    // line per row (the lane's own 16-byte pieces would reach HBM as partial lines)
                    *reinterpret_cast<bf16x8*>(stage + i * BG_STAGE_STRIDE + (t & 3) * 32 + 16 * h) = xf[t];
                    if ((t & 3) == 3) {
                        char* dst = reinterpret_cast<char*>(a.save_A + (size_t)tile * 32 * K + (t >> 2) * 64);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int r = 8 * u + (lane >> 3), c16 = lane & 7;
                            *reinterpret_cast<bf16x8*>(dst + (size_t)r * K * 2 + c16 * 16) = *reinterpret_cast<const bf16x8*>(stage + r * BG_STAGE_STRIDE + c16 * 16);
                        }
                    }
                }
            }
        } else {
            const __bf16* ar = reinterpret_cast<const __bf16*>(a.A) + row * K + 8 * h;
#ifdef T2S_CHAIN_BOUND
            if constexpr (PRO == BPRO_GELU) {                           // fc2: u would be in registers
                const f32x8 c = {0.5f, -0.5f, 0.25f, 1.f, -1.f, 0.125f, 0.f, 0.75f};
#pragma unroll
                for (int t = 0; t < KS; ++t) xf[t] = __builtin_convertvector(c * (float)(1 + ((lane + t) & 3)), bf16x8);
                (void)ar;
            } else
#endif
#pragma unroll
            for (int t = 0; t < KS; ++t) xf[t] = *reinterpret_cast<const bf16x8*>(ar + 16 * t);
            if constexpr (PRO == BPRO_GELU) {
#pragma unroll
                for (int t = 0; t < KS; ++t) {
                    f32x8 u = unpack8(xf[t]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) u[e] = gelu_tanh(u[e]);
                    xf[t] = __builtin_convertvector(u, bf16x8);
                    if (a.save_A != nullptr) *reinterpret_cast<bf16x8*>(a.save_A + row * K + 16 * t + 8 * h) = xf[t];
                }
            }
        }

        // ---- n-tiles: 32 outputs each, K/16 MFMAs; result register r = output 8(r>>2) + 4h + (r&3).
        // The LDS offset is made opaque per token tile: the weights are loop-invariant, and hipcc would
        // otherwise hoist ALL their LDS reads out of the persistent loop into (spilled) registers.
        int wo = lane;
        asm volatile("" : "+v"(wo));
        if constexpr (EPI == BEPI_LNBWD) {
            static_assert(EPI != BEPI_LNBWD || N == 128, "LayerNorm backward epilogue is over d_model = 128");
            f32x16 acc[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
                for (int t = 0; t < KS; ++t) acc[nt] = mfma16(wlds[(nt * KS + t) * 64 + wo], xf[t], acc[nt]);
            }
            float* mlds = reinterpret_cast<float*>(wl + (size_t)N * K * 2 + (size_t)N * 4 + (size_t)(THREADS / 64) * BG_STAGE_BYTES) + wave * 256;
            const bool has_gate = a.res != nullptr;
            {   // this sequence's scale and next-gate rows (2 x 128 fp32) into the wave's own LDS kilobyte
                const float* src = a.mod + (size_t)seq * MODROW + (lane < 32 || !has_gate ? a.scale_off : a.gate_off) + (lane & 31) * 4;
                *reinterpret_cast<f32x4*>(mlds + lane * 4) = *reinterpret_cast<const f32x4*>(src);
            }
            lnbwd_epilogue(acc, a.ln_x + row * 128 + 4 * h, a.dx + row * 128 + 4 * h, a.dx + (size_t)tile * 32 * 128,
                           has_gate ? a.res + row * 128 + 4 * h : nullptr, a.out + (size_t)tile * 32 * 128,
                           a.colpart + (size_t)tile * 384, mlds + 4 * h, stage, lane);
            continue;
        }
#pragma unroll 2
        for (int nt = 0; nt < NT; ++nt) {
            bf16x4 aux4[4];
            if constexpr (EPI == BEPI_GELUBWD) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    aux4[g] = *reinterpret_cast<const bf16x4*>(a.aux + row * N + nt * 32 + 8 * g + 4 * h);
            }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int t = 0; t < KS; ++t) acc = mfma16(wlds[(nt * KS + t) * 64 + wo], xf[t], acc);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = nt * 32 + 8 * g + 4 * h;
                f32x4 y = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                y += *reinterpret_cast<const f32x4*>(blds + col);
                if constexpr (EPI == BEPI_QKV) {
                    if ((nt >> 2) == 0) y = y * ATT_QS;        // q heads: scores come out in the log2 domain
                }
                if constexpr (EPI == BEPI_GELUBWD) {
                    const f32x4 u = unpack4(aux4[g]);
                    y.x *= gelu_tanh_grad(u.x); y.y *= gelu_tanh_grad(u.y);
                    y.z *= gelu_tanh_grad(u.z); y.w *= gelu_tanh_grad(u.w);
                }
                // stage: row i, byte (nt & 1) * 64 + 2 * (8g + 4h)
                *reinterpret_cast<bf16x4*>(stage + i * BG_STAGE_STRIDE + (EPI == BEPI_QKV ? 0 : (nt & 1) * 64) + 16 * g + 8 * h) = pack4(y);
            }
            if constexpr (EPI == BEPI_QKV) {
                // one head tile: 32 tokens x 64 B, contiguous in the (bh, tok, d) tensor: two 1 KiB stores
                __bf16* base = (nt >> 2) == 0 ? a.q : ((nt >> 2) == 1 ? a.k : a.v);
                char* dst = reinterpret_cast<char*>(base + (((size_t)seq * NH + (nt & 3)) * NTOK + (tok - i)) * DH);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int r = 16 * u + (lane >> 2), c16 = lane & 3;
                    *reinterpret_cast<bf16x8*>(dst + r * 64 + c16 * 16) = *reinterpret_cast<const bf16x8*>(stage + r * BG_STAGE_STRIDE + c16 * 16);
                }
            } else if (nt & 1) {
                // two n-tiles = 64 columns = one 128-byte line per row: four stores of 8 rows x 128 B
                char* dst = reinterpret_cast<char*>(a.out + ((size_t)tile * 32) * N + (nt - 1) * 32);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = 8 * u + (lane >> 3), c16 = lane & 7;
                    *reinterpret_cast<bf16x8*>(dst + (size_t)r * N * 2 + c16 * 16) = *reinterpret_cast<const bf16x8*>(stage + r * BG_STAGE_STRIDE + c16 * 16);
                }
            }
        }
    }
}

template <int K, int N, int PRO, int EPI>
inline int launch_bgemm(const BGemmArgs& a, hipStream_t st) {
    constexpr int THREADS = bg_threads(EPI);
    if (a.M <= 0 || a.M % 32 != 0 || a.N != N) {
        set_error("bgemm: M=%d must be a positive multiple of 32 and N=%d must equal %d", a.M, a.N, N);
        return T2S_E_INVALID;
    }
    constexpr int lds = N * K * 2 + N * 4 + (THREADS / 64) * BG_STAGE_BYTES + (EPI == BEPI_LNBWD ? (THREADS / 64) * 1024 : 0);
    static int n_cu = 0;
    static bool attr = false;   // first call is never under stream capture (training is not captured)
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        T2S_HIP_CHECK(hipGetDevice(&dev));
        T2S_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount;
    }
    if (!attr) {
        T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(bgemm_kernel<K, N, PRO, EPI>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr = true;
    }
    const int tiles = a.M / 32, per_wg = THREADS / 64;
    int grid = (tiles + per_wg - 1) / per_wg < n_cu ? (tiles + per_wg - 1) / per_wg : n_cu;
    BGemmArgs a2 = a;
    a2.reverse = next_tile_dir();
    bgemm_kernel<K, N, PRO, EPI><<<grid, THREADS, lds, st>>>(a2);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

// colpart (M/32, 3, 128) of BEPI_LNBWD -> dmod[seq][shift_off | scale_off | gate_off + f]: the 15 tile rows of a sequence
// added in tile order (deterministic).  grid = sequences, 384 threads (256 when there is no gate column).
static __global__ void lnbwd_colsum_reduce_kernel(const float* __restrict__ part, float* __restrict__ dmod, int shift_off,
                                                  int scale_off, int gate_off) {
    const int seq = blockIdx.x, t = threadIdx.x, q = t >> 7, f = t & 127;
    const float* p = part + (size_t)seq * (NTOK / 32) * 384 + t;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NTOK / 32; ++k) s += p[k * 384];
    dmod[(size_t)seq * MODROW + (q == 0 ? shift_off : (q == 1 ? scale_off : gate_off)) + f] = s;
}

// ------------------------------------------------------------------------------------------ wgrad
// dW[n][k] = sum_rows dY[row][n] X[row][k],  db[n] = sum_rows dY[row][n], in two deterministic
// stages: wgrad16_kernel writes one fp32 partial tile (128 x 128, + 128 bias sums) per workgroup
// with plain coalesced stores, wgrad16_reduce_kernel adds the partials of all row slabs in slab
// order.  (fp32 atomics into the 64 KB gradient tile from ~500 workgroups ran at 0.3 TB/s and
// cost 100 us per call -- more than streaming the operands.)
// (Round 2 measured the alternative of ONE workgroup per row slab owning all (n, k) chunk pairs -- 4 / 8 / 8 / 12 waves
// sharing one LDS copy of dY and X, each operand crossing HBM once instead of X of the qkv gradient three times and dY of
// fc2 / X of fc1 twice: 8 % SLOWER on the same box, 2.15 vs 1.99 ms per step for the 16 launches (tools/ab_lib.sh); the
// wider barrier domain costs more than the re-reads, which the 256 MB Infinity Cache largely serves.  Kept per chunk pair.)
// grid (row slabs, N/128, K/128); wave w of a workgroup owns outputs [128 y + 32 w, +32) x inputs
// [128 z, +128).  MFMA: D[n][k] = sum_row A[n][row] B[row][k]: both operands are read column-wise
// out of row-major LDS tiles (64 rows per pass, row stride 320 B: four consecutive rows land in
// disjoint bank quarters, so the transposed reads are conflict-free).
// XGELU: the X operand is gelu(X) (the fc2 weight gradient needs gelu(u); applying it to the fetched chunks here
// costs VALU time the HBM-bound kernel has to spare and saves writing + re-reading a (M,256) tensor per block).
template <bool XGELU>
static __global__ __launch_bounds__(256) void wgrad16_kernel(const __bf16* __restrict__ dY, const __bf16* __restrict__ X,
                                                      float* __restrict__ part, float* __restrict__ bpart, int M, int N,
                                                      int K, int rows_per_wg, int gx, int gy, int gz, int reverse) {
    // XCD-aware 1-D grid: workgroup ids go round-robin over the 8 XCDs, so id = (slab8 * tiles + tile) * 8 + xcd puts the
    // gy * gz workgroups that read the SAME row slab (different 128-column chunks of dY / X) on one XCD, back to back:
    // the slab crosses HBM once and the siblings hit that XCD's L2.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, tiles = gy * gz;
    const int tile_id = slot % tiles, bx_lin = (slot / tiles) * 8 + xcd;
    if (bx_lin >= gx) return;
    const int bx = reverse ? gx - 1 - bx_lin : bx_lin;
    const int by = tile_id / gz, bz = tile_id - by * gz;
    constexpr int KT = 4;          // 32-wide k-tiles per workgroup
    constexpr int SLAB = 64;
    constexpr int STR = 320;
    __shared__ __attribute__((aligned(16))) char ys[SLAB * STR];
    __shared__ __attribute__((aligned(16))) char xs[SLAB * STR];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int ncol0 = by * 128;
    const int kcol0 = bz * 128;
    const int r0 = bx * rows_per_wg;
    const int r1 = min(M, r0 + rows_per_wg);
    // transposed-read addressing of this lane
    const int grp = (lane >> 4) & 3, nhalf = grp & 1, q = (lane & 15) >> 2, p = lane & 3;
    const int rbase = 8 * half + q;                      // row inside a 16-row k-step (second read: +4)
    const char* ya = ys + rbase * STR + (wave * 32 + 16 * nhalf + 4 * p) * 2;
    const char* xa = xs + rbase * STR + (16 * nhalf + 4 * p) * 2;

    f32x16 acc[KT], accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const f32x8 onesf = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    const bf16x8 ones = __builtin_convertvector(onesf, bf16x8);
    const bf16x8 zero8 = __builtin_convertvector(onesf * 0.f, bf16x8);

    // Register-staged, software-pipelined row slabs: the NEXT slab's 16-byte chunks are in flight
    // while the current one is multiplied (one chunk at a time behind vmcnt(0) cost 8 HBM round
    // trips per slab).  Rows past the end are clamped for the load and zeroed afterwards.
    constexpr int CH = SLAB * 16 / 256;             // 16-byte chunks of each operand per thread and slab
    bf16x8 py[CH], px[CH];
    auto fetch = [&](int rs) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int idx = tid + 256 * u, rr = idx >> 4, c = idx & 15;
            const int row = min(rs + rr, r1 - 1);
            py[u] = *reinterpret_cast<const bf16x8*>(dY + (size_t)row * N + ncol0 + c * 8);
            px[u] = *reinterpret_cast<const bf16x8*>(X + (size_t)row * K + kcol0 + c * 8);
        }
    };
    fetch(r0);
    for (int rs = r0; rs < r1; rs += SLAB) {
        __syncthreads();   // previous pass fully consumed
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int idx = tid + 256 * u, rr = idx >> 4, c = idx & 15;
            const bool in = rs + rr < r1;
            *reinterpret_cast<bf16x8*>(ys + rr * STR + c * 16) = in ? py[u] : zero8;
            bf16x8 xv = px[u];
            if constexpr (XGELU) {
                f32x8 g = unpack8(xv);
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = gelu_tanh(g[e]);
                xv = __builtin_convertvector(g, bf16x8);
            }
            *reinterpret_cast<bf16x8*>(xs + rr * STR + c * 16) = in ? xv : zero8;
        }
        __syncthreads();
        if (rs + SLAB < r1) fetch(rs + SLAB);
#pragma unroll
        for (int s = 0; s < SLAB / 16; ++s) {
            const bf16x8 af = join_tr(lds_tr16(ya + (16 * s) * STR), lds_tr16(ya + (16 * s + 4) * STR));
            if (bz == 0) accb = mfma16(af, ones, accb);
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                const bf16x8 bf = join_tr(lds_tr16(xa + (16 * s) * STR + t * 64), lds_tr16(xa + (16 * s + 4) * STR + t * 64));
                acc[t] = mfma16(af, bf, acc[t]);
            }
        }
    }
    // partial tile of this workgroup: [(x * gy + y) * gz + z][128 n][128 k], bias sums [(x * gy + y)][128]
    const int j = lane & 31;
    float* pt = part + ((size_t)(bx * gy + by) * gz + bz) * (128 * 128);
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) pt[(wave * 32 + acc_row(r, half)) * 128 + t * 32 + j] = acc[t][r];
    if (j == 0 && bz == 0) {
        float* bp = bpart + (size_t)(bx * gy + by) * 128;
#pragma unroll
        for (int r = 0; r < 16; ++r) bp[wave * 32 + acc_row(r, half)] = accb[r];
    }
}

// Stage 2: dW[128 y + n][128 z + k] = sum_x part[x][y][z][n][k]; db[128 y + n] = sum_x bpart[x][y][n].
// grid (65, gy * gz): blockIdx.x < 64 = a 1 KiB run of the tile, 64 = the bias; the four 64-lane
// groups of a workgroup take every 4th slab, then their sums are added in a fixed order.
static __global__ __launch_bounds__(256) void wgrad16_reduce_kernel(const float* __restrict__ part,
                                                                    const float* __restrict__ bpart, float* __restrict__ dW,
                                                                    float* __restrict__ db, int gx, int gy, int gz, int K) {
    __shared__ f32x4 red[4][64];
    const int c = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int tile = blockIdx.y, y = tile / gz, z = tile - y * gz;
    const bool bias = blockIdx.x == 64;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (!bias) {
        const f32x4* p = reinterpret_cast<const f32x4*>(part) + (size_t)tile * 4096 + blockIdx.x * 64 + c;
#pragma unroll 8
        for (int x = sl; x < gx; x += 4) acc += p[(size_t)x * gy * gz * 4096];
    } else if (z == 0 && c < 32) {
        const f32x4* p = reinterpret_cast<const f32x4*>(bpart) + (size_t)y * 32 + c;
#pragma unroll 8
        for (int x = sl; x < gx; x += 4) acc += p[(size_t)x * gy * 32];
    }
    red[sl][c] = acc;
    __syncthreads();
    if (sl != 0) return;
    const f32x4 s = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    if (!bias) {
        const int e = blockIdx.x * 256 + c * 4, n = e >> 7, k = e & 127;
        *reinterpret_cast<f32x4*>(dW + (size_t)(y * 128 + n) * K + z * 128 + k) = s;
    } else if (z == 0 && c < 32 && db != nullptr) {
        *reinterpret_cast<f32x4*>(db + y * 128 + c * 4) = s;
    }
}

// workgroups of stage 1 for a given shape (about three per CU in total) and the scratch they need
inline void wgrad16_plan(int M, int N, int K, int n_cu, int* rows_per_wg, int* gx) {
    const int tiles = (N / 128) * (K / 128);
    int per = 3 * n_cu / tiles;                               // row slabs
    per = per < 1 ? 1 : per;
    int rows = ((M + per - 1) / per + 63) / 64 * 64;
    rows = rows < 64 ? 64 : rows;
    *rows_per_wg = rows;
    *gx = (M + rows - 1) / rows;
}
inline size_t wgrad16_scratch_floats(int M, int n_cu) {      // worst case over the DiT's four linears AND over every M' <= M
    // gx is not monotonic in M (rows_per_wg is rounded up to 64, so a smaller M can need one more slab than a larger one): a
    // workspace built for cap_seqs sequences serves every smaller batch, so it is sized for the bound gx <= per slabs
    (void)M;
    size_t worst = 0;
    const int shapes[4][2] = {{128, 128}, {256, 128}, {384, 128}, {128, 256}};
    for (auto& sh : shapes) {
        const int tiles = (sh[0] / 128) * (sh[1] / 128);
        int per = 3 * n_cu / tiles;
        per = per < 1 ? 1 : per;
        const size_t need = (size_t)per * tiles * (128 * 128) + (size_t)per * (sh[0] / 128) * 128;
        worst = need > worst ? need : worst;
    }
    return worst;
}

template <bool XGELU = false>
inline int launch_wgrad16(const __bf16* dY, const __bf16* X, float* dW, float* db, int M, int N, int K, float* scratch,
                          size_t scratch_floats, int n_cu, hipStream_t st) {
    if (N % 128 != 0 || K % 128 != 0 || M <= 0) {
        set_error("wgrad16: unsupported shape M=%d N=%d K=%d", M, N, K);
        return T2S_E_INVALID;
    }
    int rows_per_wg, gx;
    wgrad16_plan(M, N, K, n_cu, &rows_per_wg, &gx);
    const int gy = N / 128, gz = K / 128;
    const size_t part_floats = (size_t)gx * gy * gz * (128 * 128);
    if (part_floats + (size_t)gx * gy * 128 > scratch_floats) {
        set_error("wgrad16: scratch of %zu floats is too small for M=%d N=%d K=%d", scratch_floats, M, N, K);
        return T2S_E_INVALID;
    }
    float* part = scratch;
    float* bpart = scratch + part_floats;
    wgrad16_kernel<XGELU><<<(gx + 7) / 8 * 8 * gy * gz, 256, 0, st>>>(dY, X, part, bpart, M, N, K, rows_per_wg, gx, gy, gz, next_tile_dir());
    T2S_LAUNCH_CHECK();
    wgrad16_reduce_kernel<<<dim3(65, gy * gz), 256, 0, st>>>(part, bpart, dW, db, gx, gy, gz, K);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

}  // namespace t2s
