// fp32 weight-gradient GEMM shared by the DiT training step (t2s_train.hip, T2S_TRAIN_F32) and the LA-VAE encoder backward
// (t2s_vae.hip): dW = dY^T X over M rows on v_mfma_f32_32x32x2_f32 (exact fp32), two deterministic stages (one partial tile
// per row slab, added in slab order by wgrad16_reduce_kernel of t2s_bf16.h).
#pragma once
#include "t2s_bf16.h"

namespace t2s {

// dW[n][k] += sum_rows dY[row][n] * X[row][k]   and   db[n] += sum_rows dY[row][n]
// (weight + bias gradient of a linear; N % 32 == 0, K % 128 == 0).
// A operand = dY^T straight from row-major global memory (lane = output feature n: 128-B coalesced
// segments, two token rows per MFMA); B operand = X rows, staged ONCE per workgroup in LDS
// (64-row sub-slabs, coalesced float4) and shared by the 4 waves, which own different n-tiles of
// the same 128-wide k-chunk.  Partial tiles are added to the gradient with fp32 atomics (full
// 128-B rows per wave instruction).  The bias gradient falls out of the A operand for free.
constexpr int WG_ROWS = 64;   // rows staged per LDS pass
static __global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                    float* __restrict__ part, float* __restrict__ bpart, int M, int N,
                                                    int K, int rows_per_wg) {
    __shared__ __attribute__((aligned(16))) float xs[WG_ROWS * 132];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, j = lane & 31;
    const int kc = blockIdx.z;                        // 128-wide k-chunk
    const int nt = blockIdx.y * 4 + wave;             // this wave's n-tile (N % 128 == 0)
    const int r0 = blockIdx.x * rows_per_wg;
    const int r1 = min(M, r0 + rows_per_wg);
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;
    const float* ya = dY + nt * 32 + j;
    for (int rs = r0; rs < r1; rs += WG_ROWS) {
        __syncthreads();   // previous sub-slab fully consumed
        for (int idx = threadIdx.x; idx < WG_ROWS * 32; idx += 256) {
            const int rr = idx >> 5, c4 = idx & 31;
            const int row = rs + rr;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < r1) v = *reinterpret_cast<const f32x4*>(X + (size_t)row * K + kc * 128 + c4 * 4);
            *reinterpret_cast<f32x4*>(xs + rr * 132 + c4 * 4) = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int rr = 0; rr < WG_ROWS; rr += 2) {
            const int row = rs + rr + half;
            const float a = row < r1 ? ya[(size_t)row * N] : 0.f;
            bsum += a;
            const float* xb = xs + (rr + half) * 132 + j;
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma32(a, xb[t * 32], acc[t]);
        }
    }
    // partial tile of this workgroup (summed over the row slabs by wgrad16_reduce_kernel, t2s_bf16.h)
    float* pt = part + ((size_t)(blockIdx.x * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z) * (128 * 128);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) pt[(wave * 32 + acc_row(r, half)) * 128 + t * 32 + j] = acc[t][r];
    if (kc == 0) {
        bsum += xhalf(bsum);
        if (half == 0) bpart[(size_t)(blockIdx.x * gridDim.y + blockIdx.y) * 128 + wave * 32 + j] = bsum;
    }
}


// dW (N,K) = dY^T X, db (N) = column sums of dY (db may be NULL); N % 128 == 0, K % 128 == 0.  scratch: wgrad16_scratch_floats().
inline int launch_wgrad32(const float* dY, const float* X, float* dW, float* db, int M, int N, int K, float* scratch,
                          size_t scratch_floats, int n_cu, hipStream_t st) {
    T2S_REQUIRE(N % 128 == 0 && K % 128 == 0 && M > 0, "wgrad: unsupported shape M=%d N=%d K=%d", M, N, K);
    int rows_per_wg, gx;
    wgrad16_plan(M, N, K, n_cu, &rows_per_wg, &gx);
    const int gy = N / 128, gz = K / 128;
    const size_t part_floats = (size_t)gx * gy * gz * (128 * 128);
    T2S_REQUIRE(part_floats + (size_t)gx * gy * 128 <= scratch_floats, "wgrad: scratch too small for M=%d N=%d K=%d", M, N, K);
    float* part = scratch;
    float* bpart = scratch + part_floats;
    wgrad_kernel<<<dim3(gx, gy, gz), 256, 0, st>>>(dY, X, part, bpart, M, N, K, rows_per_wg);
    T2S_LAUNCH_CHECK();
    wgrad16_reduce_kernel<<<dim3(65, gy * gz), 256, 0, st>>>(part, bpart, dW, db, gx, gy, gz, K);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

}  // namespace t2s
