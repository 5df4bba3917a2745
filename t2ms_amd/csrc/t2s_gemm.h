// Row-tile GEMM on v_mfma_f32_32x32x2_f32 for the DiT's small-K linears
// (K = 128 or 256, N = 128..3072), with fused prologues and epilogues.
//
//   out[M,N] = epi( pro(A)[M,K] @ W[N,K]^T + bias )
//
// Design (MI355X): one workgroup = 4 waves = a 64-row tile.  The tile's A rows
// are transformed by the prologue (LayerNorm + adaLN modulate, or SiLU) while
// they are staged once into LDS (row stride K+4 floats -> conflict-free
// ds_read_b128).  The weights never touch LDS: they are pre-packed in MFMA
// B-fragment order (t2s_common.h: packed_index) so every wave-level load is a
// contiguous 1 KiB served from L1/L2 (the whole DiT is 3.7 MB and stays in the
// 4 MiB per-XCD L2).  Each wave owns all 64 rows x (NT*32) columns.
// Arithmetic is exact fp32 (k-ordered fma chain in the matrix core).
#pragma once
#include "t2s_common.h"

namespace t2s {

enum { PRO_PLAIN = 0, PRO_LNMOD = 1, PRO_SILU = 2, PRO_GELU = 3 };
enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_QKV = 2, EPI_GATERES = 3, EPI_GELUBWD = 4 };

struct GemmArgs {
    const float* A;     // (M,K)
    const f32x4* Wp;    // packed weights, (N/32, K/8, 64) float4
    const float* bias;  // (N)
    float* out;         // EPI_BIAS/EPI_GELU: (M,N);  EPI_GATERES: residual stream (M,128), in place
    int M;
    int N;
    const float* mod;   // (S, MODROW) adaLN table for PRO_LNMOD / EPI_GATERES
    int shift_off;      // column offsets inside a MODROW row
    int scale_off;
    int gate_off;
    float* q;           // EPI_QKV destinations, each (S,4,480,32)
    float* k;
    float* v;
    float* save_A;      // optional (M,K): the prologue-transformed A rows (training saves LN-modulated inputs)
    const float* aux;   // EPI_GELUBWD: pre-activation u (M,N); out = acc * gelu'(u)
};

// d/du [ u * sigmoid(2 z(u)) ],  z = sqrt(2/pi) (u + 0.044715 u^3)   (GELU tanh form)
__device__ __forceinline__ float gelu_tanh_grad(float u) {
    const float c = 0.7978845608028654f;
    const float z = c * (u + 0.044715f * u * u * u);
    const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * z));
    return sg + u * sg * (1.0f - sg) * 2.0f * c * (1.0f + 3.0f * 0.044715f * u * u);
}

__device__ __forceinline__ float gelu_tanh(float x) {
    // 0.5 x (1 + tanh(u)) == x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3)
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * u));
}

template <int K, int NT, int PRO, int EPI>
__global__ __launch_bounds__(256) void gemm_rows_kernel(const GemmArgs a) {
    constexpr int BM = 64;
    constexpr int LDA = K + 4;
    constexpr int KG = K / 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int j = lane & 31;
    const int m0 = blockIdx.x * BM;

    // ---------------- prologue: stage (and transform) the A tile ----------------
    if constexpr (PRO == PRO_LNMOD) {
        // LayerNorm + modulate over d_model = 128: 32 lanes (float4 each) per row, 8 rows per pass
        static_assert(PRO != PRO_LNMOD || K == 128, "LayerNorm prologue is over d_model=128");
        const int c4 = tid & 31;
        const int rip = tid >> 5;
#pragma unroll 4
        for (int pass = 0; pass < BM / 8; ++pass) {
            const int row = pass * 8 + rip;
            const int grow = m0 + row;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (grow < a.M) v = *reinterpret_cast<const f32x4*>(a.A + (size_t)grow * K + c4 * 4);
            float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
            const float mean = s * (1.0f / 128.0f);
            f32x4 d = v - mean;
            float ss = (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
            const float rstd = rsqrtf(ss * (1.0f / 128.0f) + 1e-6f);
            const int seq = (grow < a.M ? grow : 0) / NTOK;
            const float* mrow = a.mod + (size_t)seq * MODROW;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(mrow + a.scale_off + c4 * 4);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(mrow + a.shift_off + c4 * 4);
            v = (d * rstd) * (1.0f + sc) + sh;
            if (a.save_A != nullptr && blockIdx.y == 0 && grow < a.M)
                *reinterpret_cast<f32x4*>(a.save_A + (size_t)grow * K + c4 * 4) = v;
            *reinterpret_cast<f32x4*>(smem + row * LDA + c4 * 4) = v;
        }
    } else {
        constexpr int F4 = K / 4;   // float4 per row
        for (int idx = tid; idx < BM * F4; idx += 256) {
            const int row = idx / F4, c4 = idx - row * F4;
            const int grow = m0 + row;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (grow < a.M) v = *reinterpret_cast<const f32x4*>(a.A + (size_t)grow * K + c4 * 4);
            if constexpr (PRO == PRO_SILU) {
                v.x = v.x / (1.0f + __expf(-v.x));
                v.y = v.y / (1.0f + __expf(-v.y));
                v.z = v.z / (1.0f + __expf(-v.z));
                v.w = v.w / (1.0f + __expf(-v.w));
            } else if constexpr (PRO == PRO_GELU) {
                v.x = gelu_tanh(v.x); v.y = gelu_tanh(v.y); v.z = gelu_tanh(v.z); v.w = gelu_tanh(v.w);
            }
            if (a.save_A != nullptr && blockIdx.y == 0 && grow < a.M)
                *reinterpret_cast<f32x4*>(a.save_A + (size_t)grow * K + c4 * 4) = v;
            *reinterpret_cast<f32x4*>(smem + row * LDA + c4 * 4) = v;
        }
    }
    __syncthreads();

    // ---------------- main loop: 2 x NT tiles of 32x32 per wave ----------------
    const int nt0 = (blockIdx.y * 4 + wave) * NT;  // first n-tile of this wave
    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const f32x4* wp = a.Wp + (size_t)nt0 * KG * 64 + lane;
    const float* ar = smem + j * LDA + 4 * half;

    f32x4 bcur[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bcur[nt] = wp[(size_t)nt * KG * 64];

#pragma unroll 4
    for (int g = 0; g < KG; ++g) {
        f32x4 bnext[NT];
        const int gn = (g + 1 < KG) ? g + 1 : g;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bnext[nt] = wp[((size_t)nt * KG + gn) * 64];
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(ar + 8 * g);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(ar + 32 * LDA + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[0][nt] = mfma32(a0[e], bcur[nt][e], acc[0][nt]);
                acc[1][nt] = mfma32(a1[e], bcur[nt][e], acc[1][nt]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bcur[nt] = bnext[nt];
    }

    // ---------------- epilogue ----------------
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = (nt0 + nt) * 32 + j;
        const float bias = a.bias ? a.bias[col] : 0.f;
        float* qkv_base = nullptr;
        int head = 0;
        if constexpr (EPI == EPI_QKV) {
            const int which = col >> 7;
            head = (col >> 5) & 3;
            qkv_base = which == 0 ? a.q : (which == 1 ? a.k : a.v);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int grow = m0 + mt * 32 + acc_row(r, half);
                if (grow >= a.M) continue;
                const float y = acc[mt][nt][r] + bias;
                if constexpr (EPI == EPI_BIAS) {
                    a.out[(size_t)grow * a.N + col] = y;
                } else if constexpr (EPI == EPI_GELU) {
                    a.out[(size_t)grow * a.N + col] = gelu_tanh(y);
                } else if constexpr (EPI == EPI_GELUBWD) {
                    a.out[(size_t)grow * a.N + col] = y * gelu_tanh_grad(a.aux[(size_t)grow * a.N + col]);
                } else if constexpr (EPI == EPI_QKV) {
                    const int seq = grow / NTOK;
                    const int tok = grow - seq * NTOK;
                    qkv_base[(((size_t)seq * NH + head) * NTOK + tok) * DH + j] = y;
                } else {  // EPI_GATERES
                    const int seq = grow / NTOK;
                    const float gate = a.mod[(size_t)seq * MODROW + a.gate_off + col];
                    float* px = a.out + (size_t)grow * D + col;
                    *px = *px + gate * y;
                }
            }
        }
    }
}

// Raise the dynamic-LDS cap of an instantiation that needs > 64 KB (call once, outside any
// stream capture).
template <int K, int NT, int PRO, int EPI>
inline int gemm_rows_init() {
    constexpr int lds = 64 * (K + 4) * (int)sizeof(float);
    T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rows_kernel<K, NT, PRO, EPI>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    return T2S_OK;
}

template <int K, int NT, int PRO, int EPI>
inline int launch_gemm_rows(const GemmArgs& a, hipStream_t st) {
    constexpr int BN = 4 * NT * 32;
    if (a.N % BN != 0 || a.M <= 0) {
        set_error("gemm_rows: N=%d not a multiple of %d or M=%d <= 0", a.N, BN, a.M);
        return T2S_E_INVALID;
    }
    constexpr size_t lds = (size_t)64 * (K + 4) * sizeof(float);
    if constexpr (lds > 48 * 1024) {
        static bool attr = false;   // first call must not be under stream capture (training never is)
        if (!attr) {
            T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rows_kernel<K, NT, PRO, EPI>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr = true;
        }
    }
    dim3 grid((a.M + 63) / 64, a.N / BN);
    gemm_rows_kernel<K, NT, PRO, EPI><<<grid, 256, lds, st>>>(a);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

}  // namespace t2s
