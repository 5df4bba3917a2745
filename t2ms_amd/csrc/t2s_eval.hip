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


// ---- ED (evaluation.py:137-150): per sample the mean over series of || ori[:, j] - gen[:, j] ||_2 over time
__global__ __launch_bounds__(64) void eval_ed_kernel(const float* __restrict__ ori, const float* __restrict__ gen,
                                                     float* __restrict__ per_sample, int L, int S) {
    const int i = blockIdx.x;
    const float* a = ori + (size_t)i * L * S;
    const float* b = gen + (size_t)i * L * S;
    double total = 0.0;
    for (int j = 0; j < S; ++j) {
        double ss = 0.0;
        for (int t = threadIdx.x; t < L; t += 64) {
            const double d = (double)a[t * S + j] - (double)b[t * S + j];
            ss += d * d;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
        total += sqrt(ss);
    }
    if (threadIdx.x == 0) per_sample[i] = (float)(total / (double)S);
}

// ---- CRPS (evaluation.py:51-83): per (sample, series, run) a Gaussian N(mean, std) fitted to the generated series over
// TIME (population std, 1e-8 when 0); mean over time of (1[obs >= mean] - Phi((obs - mean) / std))^2; then the mean over
// runs and series.  gen is run-major (runs, n, L, S).  One wave per sample, fp64 statistics.
__global__ __launch_bounds__(64) void eval_crps_kernel(const float* __restrict__ ori, const float* __restrict__ gen,
                                                       float* __restrict__ per_sample, int n, int L, int S, int runs) {
    const int i = blockIdx.x;
    const float* a = ori + (size_t)i * L * S;
    double total = 0.0;
    for (int j = 0; j < S; ++j) {
        double over_runs = 0.0;
        for (int k = 0; k < runs; ++k) {
            const float* g = gen + ((size_t)k * n + i) * L * S;
            double sm = 0.0;
            for (int t = threadIdx.x; t < L; t += 64) sm += (double)g[t * S + j];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) sm += __shfl_xor(sm, o, 64);
            const double mean = sm / (double)L;
            double sv = 0.0;
            for (int t = threadIdx.x; t < L; t += 64) {
                const double d = (double)g[t * S + j] - mean;
                sv += d * d;
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) sv += __shfl_xor(sv, o, 64);
            // the reference takes mean / std of float32 data in float32 and compares float32 values
            const float meanf = (float)mean;
            float stdf = (float)sqrt(sv / (double)L);
            if (stdf == 0.f) stdf += 1e-8f;
            double acc = 0.0;
            for (int t = threadIdx.x; t < L; t += 64) {
                const float obs = a[t * S + j];
                const double step = obs < meanf ? 0.0 : 1.0;
                const double z = ((double)obs - (double)meanf) / (double)stdf;
                const double cdf = 0.5 * erfc(-z * 0.70710678118654752440);
                acc += (step - cdf) * (step - cdf);
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
            over_runs += acc / (double)L;
        }
        total += over_runs / (double)runs;
    }
    if (threadIdx.x == 0) per_sample[i] = (float)(total / (double)S);
}

// ---- DTW (evaluation.py:152-163 = dtaidistance dtw_ndim.distance, no window / penalty): sqrt of the minimal warping-path
// cost with squared-Euclidean point costs between the (L, S) sequences.  One workgroup per sample walks the anti-diagonals
// of the L x L table (cells of a diagonal are independent); three diagonals of fp64 partial costs live in LDS.
__global__ __launch_bounds__(256) void eval_dtw_kernel(const float* __restrict__ ori, const float* __restrict__ gen,
                                                       float* __restrict__ per_sample, int L, int S) {
    extern __shared__ double diag[];          // [3][L]
    const int smp = blockIdx.x;
    const float* a = ori + (size_t)smp * L * S;
    const float* b = gen + (size_t)smp * L * S;
    const double INF = 1e300;
    for (int d = 0; d <= 2 * L - 2; ++d) {
        double* cur = diag + (d % 3) * L;
        const double* p1 = diag + ((d + 2) % 3) * L;    // diagonal d-1, indexed by row i
        const double* p2 = diag + ((d + 1) % 3) * L;    // diagonal d-2
        const int lo = d - (L - 1) > 0 ? d - (L - 1) : 0, hi = d < L - 1 ? d : L - 1;
        for (int i = lo + (int)threadIdx.x; i <= hi; i += 256) {
            const int j = d - i;
            double c = 0.0;
            for (int s = 0; s < S; ++s) {
                const double df = (double)a[i * S + s] - (double)b[j * S + s];
                c += df * df;
            }
            double best;
            if (i == 0 && j == 0) {
                best = 0.0;
            } else {
                const double up = i > 0 ? p1[i - 1] : INF;                 // (i-1, j)   on diagonal d-1
                const double left = j > 0 ? p1[i] : INF;                   // (i, j-1)   on diagonal d-1
                const double dg = (i > 0 && j > 0) ? p2[i - 1] : INF;      // (i-1, j-1) on diagonal d-2
                best = fmin(dg, fmin(up, left));
            }
            cur[i] = c + best;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) per_sample[smp] = (float)sqrt(diag[((2 * L - 2) % 3) * L + (L - 1)]);
}

// ---- TS2Vec encoder forward (evaluate/ts2vec.py:366-399 in eval mode, mask 'all_true'): one workgroup per series, the
// three (channels x T) activation planes live in LDS; weights (< 1 MB) are read through L1 / L2.
struct Ts2vecDev {
    int cin, hidden, cout, depth, T;
    const float *fc_w, *fc_b;
    const float *c1w[T2S_TS2VEC_MAX_BLOCKS], *c1b[T2S_TS2VEC_MAX_BLOCKS], *c2w[T2S_TS2VEC_MAX_BLOCKS], *c2b[T2S_TS2VEC_MAX_BLOCKS];
    const float *pw, *pb;
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// out[co][t] = bias[co] + sum_ci sum_k w[co][ci][k] * gelu(in[ci][t + (k-1) dil]) (zero outside [0,T)) [+ res[co][t]]
__device__ void ts2vec_conv(const float* __restrict__ w, const float* __restrict__ bias, const float* in, float* out,
                            const float* res, int ci_n, int co_n, int T, int dil) {
    for (int idx = threadIdx.x; idx < co_n * T; idx += 256) {
        const int co = idx / T, t = idx - co * T;
        const float* wr = w + (size_t)co * ci_n * 3;
        const int tl = t - dil, tr = t + dil;
        float acc = bias[co];
        for (int ci = 0; ci < ci_n; ++ci) {
            const float* row = in + ci * T;
            float v = wr[3 * ci + 1] * row[t];
            if (tl >= 0) v += wr[3 * ci] * row[tl];
            if (tr < T) v += wr[3 * ci + 2] * row[tr];
            acc += v;
        }
        out[idx] = res != nullptr ? acc + res[idx] : acc;
    }
}

__global__ __launch_bounds__(256) void ts2vec_encode_kernel(const Ts2vecDev w, const float* __restrict__ x,
                                                            float* __restrict__ rep, float* __restrict__ full) {
    extern __shared__ float planes[];
    const int T = w.T, cmax = w.hidden > w.cout ? w.hidden : w.cout;
    float* H = planes;                        // current activation (C, T)
    float* G = planes + (size_t)cmax * T;     // gelu(.) / conv1 output
    float* R = planes + (size_t)2 * cmax * T; // projector output of the final block
    const float* xs = x + (size_t)blockIdx.x * T * w.cin;
    // input_fc with the NaN rule: a time step holding a NaN is zeroed before AND after the linear (:367-368,388-389)
    for (int idx = threadIdx.x; idx < w.hidden * T; idx += 256) {
        const int c = idx / T, t = idx - c * T;
        bool ok = true;
        float acc = w.fc_b[c];
        for (int k = 0; k < w.cin; ++k) {
            const float v = xs[t * w.cin + k];
            ok = ok && (v == v);
            acc += w.fc_w[c * w.cin + k] * v;
        }
        H[idx] = ok ? acc : 0.f;
    }
    __syncthreads();
    for (int blk = 0; blk <= w.depth; ++blk) {
        const bool last = blk == w.depth;
        const int ci_n = w.hidden, co_n = last ? w.cout : w.hidden;
        const int dil = blk < 30 ? (1 << blk) : (1 << 30);
        for (int idx = threadIdx.x; idx < ci_n * T; idx += 256) G[idx] = gelu_erf(H[idx]);
        if (last) {   // 1x1 projector of the raw (not activated) input
            for (int idx = threadIdx.x; idx < co_n * T; idx += 256) {
                const int co = idx / T, t = idx - co * T;
                float acc = w.pb[co];
                for (int ci = 0; ci < ci_n; ++ci) acc += w.pw[co * ci_n + ci] * H[ci * T + t];
                R[idx] = acc;
            }
        }
        __syncthreads();
        // conv1(gelu(h)) -> overwrite H is not possible in place for co != ci; H is dead once R / the residual is taken:
        // non-final blocks add the residual H element-wise at the very end, so conv1 goes to R's plane as scratch
        float* Y1 = last ? H : R;             // final block: H is dead after the projector; others: R is free
        ts2vec_conv(w.c1w[blk], w.c1b[blk], G, Y1, nullptr, ci_n, co_n, T, dil);
        __syncthreads();
        for (int idx = threadIdx.x; idx < co_n * T; idx += 256) G[idx] = gelu_erf(Y1[idx]);
        __syncthreads();
        if (last) {
            ts2vec_conv(w.c2w[blk], w.c2b[blk], G, H, R, co_n, co_n, T, dil);       // H = conv2 + projector
        } else {
            ts2vec_conv(w.c2w[blk], w.c2b[blk], G, R, H, co_n, co_n, T, dil);       // R = conv2 + H (element-wise residual)
            __syncthreads();
            for (int idx = threadIdx.x; idx < co_n * T; idx += 256) H[idx] = R[idx];
        }
        __syncthreads();
    }
    // outputs: rep (B, T, cout) time-major like the reference's transpose; full = max over time
    if (rep != nullptr)
        for (int idx = threadIdx.x; idx < w.cout * T; idx += 256) {
            const int t = idx / w.cout, c = idx - t * w.cout;
            rep[((size_t)blockIdx.x * T + t) * w.cout + c] = H[c * T + t];
        }
    for (int c = threadIdx.x; c < w.cout; c += 256) {
        float m = H[c * T];
        for (int t = 1; t < T; ++t) m = fmaxf(m, H[c * T + t]);
        full[(size_t)blockIdx.x * w.cout + c] = m;
    }
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


extern "C" int t2s_eval_ed(const float* ori, const float* gen, float* per_sample, float* out, int n, int L, int n_series,
                           void* stream) {
    using namespace t2s;
    T2S_REQUIRE(ori && gen && per_sample && out && n > 0 && L > 0 && n_series > 0, "t2s_eval_ed: bad argument");
    hipStream_t st = (hipStream_t)stream;
    eval_ed_kernel<<<n, 64, 0, st>>>(ori, gen, per_sample, L, n_series);
    T2S_LAUNCH_CHECK();
    eval_mean_kernel<<<1, 256, 0, st>>>(per_sample, out, n);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

extern "C" int t2s_eval_crps(const float* ori, const float* gen, float* per_sample, float* out, int n, int L, int n_series,
                             int runs, void* stream) {
    using namespace t2s;
    T2S_REQUIRE(ori && gen && per_sample && out && n > 0 && L > 0 && n_series > 0 && runs > 0, "t2s_eval_crps: bad argument");
    hipStream_t st = (hipStream_t)stream;
    eval_crps_kernel<<<n, 64, 0, st>>>(ori, gen, per_sample, n, L, n_series, runs);
    T2S_LAUNCH_CHECK();
    eval_mean_kernel<<<1, 256, 0, st>>>(per_sample, out, n);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

extern "C" int t2s_eval_dtw(const float* ori, const float* gen, float* per_sample, float* out, int n, int L, int n_series,
                            void* stream) {
    using namespace t2s;
    T2S_REQUIRE(ori && gen && per_sample && out && n > 0 && L > 0 && n_series > 0, "t2s_eval_dtw: bad argument");
    T2S_REQUIRE(L <= 4096, "t2s_eval_dtw: L=%d exceeds 4096 (three fp64 diagonals must fit in LDS)", L);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)3 * L * sizeof(double);
    if (lds > 48 * 1024)
        T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(eval_dtw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    eval_dtw_kernel<<<n, 256, lds, st>>>(ori, gen, per_sample, L, n_series);
    T2S_LAUNCH_CHECK();
    eval_mean_kernel<<<1, 256, 0, st>>>(per_sample, out, n);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

extern "C" int t2s_ts2vec_encode(const t2s_ts2vec_weights* w, const float* x, float* rep, float* full, int B, int T,
                                 void* stream) {
    using namespace t2s;
    T2S_REQUIRE(w && x && full && B > 0 && T > 0, "t2s_ts2vec_encode: bad argument");
    T2S_REQUIRE(w->depth >= 0 && w->depth < T2S_TS2VEC_MAX_BLOCKS && w->input_dims > 0 && w->hidden > 0 && w->output_dims > 0,
                "t2s_ts2vec_encode: unsupported sizes (depth=%d, max %d)", w->depth, T2S_TS2VEC_MAX_BLOCKS - 1);
    T2S_REQUIRE(w->fc_w && w->fc_b && w->proj_w && w->proj_b, "t2s_ts2vec_encode: NULL weight");
    Ts2vecDev d{};
    d.cin = w->input_dims; d.hidden = w->hidden; d.cout = w->output_dims; d.depth = w->depth; d.T = T;
    d.fc_w = w->fc_w; d.fc_b = w->fc_b; d.pw = w->proj_w; d.pb = w->proj_b;
    for (int i = 0; i <= w->depth; ++i) {
        T2S_REQUIRE(w->conv1_w[i] && w->conv1_b[i] && w->conv2_w[i] && w->conv2_b[i], "t2s_ts2vec_encode: NULL weight of block %d", i);
        d.c1w[i] = w->conv1_w[i]; d.c1b[i] = w->conv1_b[i]; d.c2w[i] = w->conv2_w[i]; d.c2b[i] = w->conv2_b[i];
    }
    const int cmax = d.hidden > d.cout ? d.hidden : d.cout;
    const size_t lds = (size_t)3 * cmax * T * sizeof(float);
    T2S_REQUIRE(lds <= 160 * 1024, "t2s_ts2vec_encode: 3 x %d channels x T=%d fp32 planes exceed the 160 KB LDS of a CU", cmax, T);
    T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ts2vec_encode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ts2vec_encode_kernel<<<B, 256, lds, (hipStream_t)stream>>>(d, x, rep, full);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}
