// Reconstruction metrics of the generate -> evaluate loop on the GPU: MSE and WAPE exactly as
// evaluation.py:166-206 defines them on the (N, L, n_series) arrays infer.py writes (x_1.npy / x_t.npy).
//   mse_i  = mean over (L, series) of (ori - gen)^2              MSE  = mean_i mse_i
//   wape_i = sum |ori - gen| / sum |ori|   (NaN if the sum is 0)  WAPE = nanmean_i wape_i
// Deterministic: one wave per sample with a fixed summation order, then one workgroup over the samples.
//
// MRR as evaluation.py:21-45 defines it over the G generated runs of each sample (x_t.npy of run_0..run_{G-1}):
//   sim_g = <ori, gen_g> / (|ori| |gen_g|) over the flattened (L, series) values, 0 where that is not finite
//   (cosine_similarity, Dataset_Construction_Pipeline/Evaluate_Datasets.py:6-15 with nan_to_num);
//   the runs are visited by descending sim and the first one above the threshold ends the visit, so only the
//   best run can: score_i = 1 / (g* + 1) if sim_{g*} > threshold else 0, where g* is the RUN INDEX of the best
//   run (evaluation.py:37-41 take `idx + 1` of the argsort entry, not its position); ties go to the largest
//   index (reversed stable argsort).  MRR = mean_i score_i.
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

// one wave per sample: double accumulation, so the only rounding is the final cast of each similarity
__global__ __launch_bounds__(64) void eval_mrr_kernel(const float* __restrict__ ori, const float* __restrict__ gen,
                                                      float* __restrict__ sims, float* __restrict__ score,
                                                      int n, int len, int runs, float threshold) {
    const int i = blockIdx.x;
    const float* a = ori + (size_t)i * len;
    double aa = 0.0;
    for (int k = threadIdx.x; k < len; k += 64) aa += (double)a[k] * (double)a[k];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) aa += __shfl_xor(aa, o, 64);
    float best = 0.f;
    int best_g = -1;
    for (int g = 0; g < runs; ++g) {
        const float* b = gen + ((size_t)g * n + i) * len;
        double ab = 0.0, bb = 0.0;
        for (int k = threadIdx.x; k < len; k += 64) {
            ab += (double)a[k] * (double)b[k];
            bb += (double)b[k] * (double)b[k];
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            ab += __shfl_xor(ab, o, 64);
            bb += __shfl_xor(bb, o, 64);
        }
        const double den = sqrt(aa) * sqrt(bb);
        const float sf = den > 0.0 ? (float)(ab / den) : 0.f;   // 0/0 -> NaN -> nan_to_num -> 0 (a zero vector)
        if (best_g < 0 || sf >= best) { best = sf; best_g = g; }
        if (threadIdx.x == 0) sims[(size_t)i * runs + g] = sf;
    }
    if (threadIdx.x == 0) score[i] = (best_g >= 0 && best > threshold) ? 1.f / (float)(best_g + 1) : 0.f;
}

__global__ __launch_bounds__(256) void eval_mean_kernel(const float* __restrict__ v, float* __restrict__ out, int n) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)v[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] / (double)n);
}

}  // namespace
}  // namespace t2s

extern "C" int t2s_eval_mrr(const float* ori, const float* gen, float* sims, float* score, float* out, int n, int len,
                            int runs, float threshold, void* stream) {
    using namespace t2s;
    T2S_REQUIRE(ori && gen && sims && score && out && n > 0 && len > 0 && runs > 0, "t2s_eval_mrr: bad argument");
    hipStream_t st = (hipStream_t)stream;
    eval_mrr_kernel<<<n, 64, 0, st>>>(ori, gen, sims, score, n, len, runs, threshold);
    T2S_LAUNCH_CHECK();
    eval_mean_kernel<<<1, 256, 0, st>>>(score, out, n);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

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
