// Reconstruction metrics of the generate -> evaluate loop on the GPU: MSE and WAPE exactly as
// evaluation.py:166-206 defines them on the (N, L, n_series) arrays infer.py writes (x_1.npy / x_t.npy).
//   mse_i  = mean over (L, series) of (ori - gen)^2              MSE  = mean_i mse_i
//   wape_i = sum |ori - gen| / sum |ori|   (NaN if the sum is 0)  WAPE = nanmean_i wape_i
// Deterministic: one wave per sample with a fixed summation order, then one workgroup over the samples.
#include "t2s_common.h"

namespace t2s {
namespace {

__global__ __launch_bounds__(64) void eval_per_sample_kernel(const float* __restrict__ ori, const float* __restrict__ gen,
                                                             float* __restrict__ per_sample, int len) {
    const int i = blockIdx.x;
    const float* a = ori + (size_t)i * len;
    const float* b = gen + (size_t)i * len;
    float se = 0.f, ae = 0.f, av = 0.f;
    for (int k = threadIdx.x; k < len; k += 64) {
        const float d = a[k] - b[k];
        se += d * d;
        ae += fabsf(d);
        av += fabsf(a[k]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        se += __shfl_xor(se, o, 64);
        ae += __shfl_xor(ae, o, 64);
        av += __shfl_xor(av, o, 64);
    }
    if (threadIdx.x == 0) {
        per_sample[2 * i] = se / (float)len;
        per_sample[2 * i + 1] = av != 0.f ? ae / av : __builtin_nanf("");
    }
}

__global__ __launch_bounds__(256) void eval_reduce_kernel(const float* __restrict__ per_sample, float* __restrict__ out, int n) {
    __shared__ double red[3][256];
    double sm = 0.0, sw = 0.0, cnt = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        sm += (double)per_sample[2 * i];
        const float w = per_sample[2 * i + 1];
        if (w == w) { sw += (double)w; cnt += 1.0; }
    }
    red[0][threadIdx.x] = sm; red[1][threadIdx.x] = sw; red[2][threadIdx.x] = cnt;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int c = 0; c < 3; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = (float)(red[0][0] / (double)n);
        out[1] = red[2][0] > 0.0 ? (float)(red[1][0] / red[2][0]) : __builtin_nanf("");
    }
}

}  // namespace
}  // namespace t2s

extern "C" int t2s_eval_mse_wape(const float* ori, const float* gen, float* per_sample, float* out, int n, int len,
                                 void* stream) {
    using namespace t2s;
    T2S_REQUIRE(ori && gen && per_sample && out && n > 0 && len > 0, "t2s_eval_mse_wape: bad argument");
    hipStream_t st = (hipStream_t)stream;
    eval_per_sample_kernel<<<n, 64, 0, st>>>(ori, gen, per_sample, len);
    T2S_LAUNCH_CHECK();
    eval_reduce_kernel<<<1, 256, 0, st>>>(per_sample, out, n);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}
