// Diffusion backbones and the fused sampling loop for MI355X.
//   DDPM.p_sample / q_sample      reference model/backbone/DDPM.py:19-36
//   RectifiedFlow.euler / flow    reference model/backbone/rectified_flow.py:5-12
//   CFG loop                      reference infer.py:75-95
// One sampling step (2B-sequence DiT forward + CFG combine + sampler update) is captured once
// into a hipGraph and replayed `steps` times; everything that changes from step to step
// (time-embedding row, DDPM coefficients, noise stream / injected-noise slice) is looked up on
// the device through a step counter that the last node of the graph advances.
#include <mutex>
#include <stdlib.h>
#include <vector>

#include "t2s_common.h"

struct t2s_dit;
struct t2s_vae;

namespace t2s {
int dit_forward_cfg_step(t2s_dit* h, const float* x, const float* temb_table, const int* step_ptr,
                         const float* text, float* out_u, float* out_c, int B, hipStream_t st, int ws_seq0,
                         const float* mod_table, int mod_rows, int mod_row0);
int dit_adaln_table(t2s_dit* h, const float* temb_table, int steps, const float* text, int B, float* table, hipStream_t st);

// ---------------------------------------------------------------- Philox4x32-10 + Box-Muller
struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        u32x4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// 4 N(0,1) draws for quad `quad` of global row `row` in stream `stream_id`
__device__ __forceinline__ f32x4 normal4(uint64_t seed, uint32_t stream_id, uint32_t row, uint32_t quad) {
    const u32x4 r = philox4x32_10(u32x4{quad, row, stream_id, 0u}, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float two_pi = 6.283185307179586f;
    const float two_m24 = 1.0f / 16777216.0f;
    // 24-bit uniforms (exact in fp32): u1 in (0,1], u2 in [0,1)
    const float u1a = ((float)(r.x >> 8) + 1.0f) * two_m24;
    const float u1b = ((float)(r.z >> 8) + 1.0f) * two_m24;
    const float u2a = (float)(r.y >> 8) * two_m24;
    const float u2b = (float)(r.w >> 8) * two_m24;
    const float ra = sqrtf(-2.0f * logf(u1a));
    const float rb = sqrtf(-2.0f * logf(u1b));
    f32x4 z;
    z.x = ra * cosf(two_pi * u2a);
    z.y = ra * sinf(two_pi * u2a);
    z.z = rb * cosf(two_pi * u2b);
    z.w = rb * sinf(two_pi * u2b);
    return z;
}

__global__ void philox_normal_kernel(float* __restrict__ out, uint64_t seed, uint32_t stream_id,
                                     uint32_t row0, int n_rows, int quads_per_row) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * quads_per_row) return;
    const int row = idx / quads_per_row, quad = idx - row * quads_per_row;
    reinterpret_cast<f32x4*>(out)[idx] = normal4(seed, stream_id, row0 + (uint32_t)row, (uint32_t)quad);
}

// U[0,1) draws of the same counter layout: element e of global row `row` = lane e % 4 of counter (e / 4, row, stream_id);
// 24-bit uniforms (exact in fp32, never 1.0)
__global__ void philox_uniform_kernel(float* __restrict__ out, uint64_t seed, uint32_t stream_id, uint32_t row0, int n_rows,
                                      int row_elems) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * row_elems) return;
    const int row = idx / row_elems, e = idx - row * row_elems;
    const u32x4 r = philox4x32_10(u32x4{(uint32_t)(e >> 2), row0 + (uint32_t)row, stream_id, 0u}, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t v = (e & 3) == 0 ? r.x : ((e & 3) == 1 ? r.y : ((e & 3) == 2 ? r.z : r.w));
    out[idx] = (float)(v >> 8) * (1.0f / 16777216.0f);
}

// ---------------------------------------------------------------- sampler update kernels
struct StepArgs {
    float* x;             // (B,1920) in place
    const float* eps_u;   // (B,1920)
    const float* eps_c;   // (B,1920) or NULL
    const float* noise;   // injected draws: (steps,noise_rows,1920) indexed by step (this shard's first row), or
                          // (B,1920) if step_ptr NULL
    const float* coef;    // DEVICE (T,3)
    const int* step_ptr;  // device {loop index j, global row of the first series}, or NULL (then t_index / stream_id /
                          // row0 are immediate)
    int steps;            // T (t = steps-1-j when step_ptr != NULL)
    int t_index;
    float cfg;
    uint64_t seed;
    uint32_t stream_id;
    uint32_t row0;
    int B;
    int noise_rows;       // rows per step of the injected-noise array (>= B: a lane steps a row range of the batch)
    int* advance;         // sampling loop: device {loop index, row0, arrival counter}; the last workgroup increments the index
};

// Last workgroup to finish advances the device loop index (step[0]; step[2] is the arrival counter): every
// workgroup has read step[0] before it arrives, so the increment cannot overtake a reader, and the stand-alone
// one-thread set_step launch per step (4 us of a 690 us step at 32 series) is gone.
__device__ __forceinline__ void advance_step_when_last(int* step) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&step[2], 1) == (int)gridDim.x - 1) {
            step[2] = 0;
            step[0] += 1;
        }
    }
}

// Grid-stride over the quads with at most STEP_MAX_WGS workgroups: the arrival counter is ONE address, and same-address
// atomics serialise at ~25 ns each -- 480 of them (one per 256 quads at B = 256) made this 6 us kernel 18 us.
constexpr int STEP_MAX_WGS = 96;
__global__ __launch_bounds__(256) void ddpm_step_kernel(const StepArgs a) {
    constexpr int QPR = LAT / 4;
    int t = a.t_index;
    uint32_t sid = a.stream_id, row0 = a.row0;
    const float* noise = a.noise;
    if (a.step_ptr) {
        const int j = a.step_ptr[0];
        row0 = (uint32_t)a.step_ptr[1];   // read on the device so a captured graph serves every shard position
        t = a.steps - 1 - j;
        sid = (uint32_t)j;
        if (noise) noise += (size_t)j * a.noise_rows * LAT;
    }
    const float c0 = a.coef[t * 3 + 0], c1 = a.coef[t * 3 + 1], c2 = a.coef[t * 3 + 2];
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < a.B * QPR; idx += gridDim.x * blockDim.x) {  // one float4 (quad) at a time
        const f32x4 x = reinterpret_cast<const f32x4*>(a.x)[idx];
        const f32x4 u = reinterpret_cast<const f32x4*>(a.eps_u)[idx];
        f32x4 pred = u;
        if (a.eps_c) {
            const f32x4 c = reinterpret_cast<const f32x4*>(a.eps_c)[idx];
            pred = u + a.cfg * (c - u);
        }
        f32x4 z;
        if (noise) {
            z = reinterpret_cast<const f32x4*>(noise)[idx];
        } else {
            const int row = idx / QPR, quad = idx - row * QPR;
            z = normal4(a.seed, sid, row0 + (uint32_t)row, (uint32_t)quad);
        }
        const f32x4 mean = c0 * (x - c1 * pred);
        reinterpret_cast<f32x4*>(a.x)[idx] = mean + c2 * z;
    }
    if (a.advance) advance_step_when_last(a.advance);
}

__global__ __launch_bounds__(256) void rf_step_kernel(float* __restrict__ x, const float* __restrict__ vu,
                                                      const float* __restrict__ vc, float cfg, float dt,
                                                      int n4, int* advance) {
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n4; idx += gridDim.x * blockDim.x) {
        const f32x4 u = reinterpret_cast<const f32x4*>(vu)[idx];
        f32x4 pred = u;
        if (vc) pred = u + cfg * (reinterpret_cast<const f32x4*>(vc)[idx] - u);
        f32x4 xv = reinterpret_cast<f32x4*>(x)[idx];
        reinterpret_cast<f32x4*>(x)[idx] = xv + pred * dt;
    }
    if (advance) advance_step_when_last(advance);
}

__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ eps,
                                                       const int32_t* __restrict__ t, const float* __restrict__ sab,
                                                       const float* __restrict__ s1m, float* __restrict__ out, int B,
                                                       int QPR, int T) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * QPR) return;
    const int tt = t[idx / QPR];
    if (tt < 0 || tt >= T) {      // the reference's gather raises (DDPM.py:7-9): no table read, a loud NaN row
        reinterpret_cast<f32x4*>(out)[idx] = f32x4{NAN, NAN, NAN, NAN};
        return;
    }
    const f32x4 a = reinterpret_cast<const f32x4*>(x0)[idx];
    const f32x4 e = reinterpret_cast<const f32x4*>(eps)[idx];
    reinterpret_cast<f32x4*>(out)[idx] = sab[tt] * a + s1m[tt] * e;
}

__global__ __launch_bounds__(256) void create_flow_kernel(const float* __restrict__ x1, const float* __restrict__ x0,
                                                          const float* __restrict__ t, float* __restrict__ out, int B) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int QPR = LAT / 4;
    if (idx >= B * QPR) return;
    const float tt = t[idx / QPR];
    const f32x4 a = reinterpret_cast<const f32x4*>(x1)[idx];
    const f32x4 b = reinterpret_cast<const f32x4*>(x0)[idx];
    reinterpret_cast<f32x4*>(out)[idx] = tt * a + (1.0f - tt) * b;
}

// DDPM.p_sample with a per-row timestep (the class API allows t to differ per row), out of place
__global__ __launch_bounds__(256) void p_sample_rows_kernel(const float* __restrict__ xt, const float* __restrict__ eh,
                                                            const int32_t* __restrict__ t, const float* __restrict__ noise,
                                                            const float* __restrict__ coef, float* __restrict__ out, int B,
                                                            int QPR, int T) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * QPR) return;
    const int tt = t[idx / QPR];
    if (tt < 0 || tt >= T) {
        reinterpret_cast<f32x4*>(out)[idx] = f32x4{NAN, NAN, NAN, NAN};
        return;
    }
    const float c0 = coef[tt * 3 + 0], c1 = coef[tt * 3 + 1], c2 = coef[tt * 3 + 2];
    const f32x4 x = reinterpret_cast<const f32x4*>(xt)[idx];
    const f32x4 e = reinterpret_cast<const f32x4*>(eh)[idx];
    const f32x4 z = reinterpret_cast<const f32x4*>(noise)[idx];
    reinterpret_cast<f32x4*>(out)[idx] = c0 * (x - c1 * e) + c2 * z;
}

// F.mse_loss (mean over all elements), deterministic and STATELESS: stage 1, every workgroup sums a fixed contiguous
// chunk in a fixed order into part[blockIdx.x] of the CALLER's scratch; stage 2, one workgroup adds the partials in index
// order.  No device globals, no arrival counter, nothing to zero: calls on different streams / from different threads
// cannot meet (until round 5 the partials lived in one __device__ array per device and two overlapping calls raced
// silently).
constexpr int MSE_MAX_WGS = 1024;
static_assert(MSE_MAX_WGS <= T2S_MSE_SCRATCH_FLOATS, "t2s.h: T2S_MSE_SCRATCH_FLOATS too small");
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ part_out, size_t n) {
    __shared__ float part[4];
    const size_t n4 = n >> 2;
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;          // float4 per workgroup
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
    float acc = 0.f;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
        const f32x4 d = reinterpret_cast<const f32x4*>(a)[i] - reinterpret_cast<const f32x4*>(b)[i];
        acc += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    }
    if (blockIdx.x == gridDim.x - 1)                                // ragged tail (n % 4 elements)
        for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
            const float d = a[i] - b[i];
            acc += d * d;
        }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part_out[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ __launch_bounds__(256) void mse_final_kernel(const float* __restrict__ part_in, int n_part, float* __restrict__ out,
                                                        size_t n) {
    __shared__ float part[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n_part; i += 256) s += part_in[i];      // fixed assignment, fixed order below
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = ((part[0] + part[1]) + (part[2] + part[3])) / (float)n;
}

// step[0] = loop index, step[1] = global row index of the lane's first series (Philox key), step[2] = arrival counter
// of the update kernel (advance_step_when_last)
__global__ void set_step_kernel(int* step, int value, uint32_t row0) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        step[0] = value;
        step[1] = (int)row0;
        step[2] = 0;
    }
}

}  // namespace t2s

using namespace t2s;

extern "C" int t2s_time_embedding(const t2s_dit* h, const float* t, float* out, int B, void* stream);

// ---------------------------------------------------------------- C ABI: single-step entry points
extern "C" int t2s_philox_normal(float* out, uint64_t seed, uint32_t stream_id, uint32_t row0, int n_rows,
                                 int row_elems, void* stream) {
    T2S_REQUIRE(out && n_rows > 0 && row_elems > 0 && row_elems % 4 == 0,
                "t2s_philox_normal: bad argument (n_rows=%d,row_elems=%d; row_elems %% 4 must be 0)", n_rows, row_elems);
    const int q = row_elems / 4, total = n_rows * q;
    philox_normal_kernel<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(out, seed, stream_id, row0, n_rows, q);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

extern "C" int t2s_philox_uniform(float* out, uint64_t seed, uint32_t stream_id, uint32_t row0, int n_rows, int row_elems,
                                  void* stream) {
    T2S_REQUIRE(out && n_rows > 0 && row_elems > 0 && (long long)n_rows * row_elems < (1ll << 31),
                "t2s_philox_uniform: bad argument (n_rows=%d,row_elems=%d)", n_rows, row_elems);
    const int total = n_rows * row_elems;
    philox_uniform_kernel<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(out, seed, stream_id, row0, n_rows, row_elems);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

extern "C" int t2s_ddpm_step(float* x, const float* eps_u, const float* eps_c, const float* noise,
                             const float* coef, int t_index, float cfg, uint64_t seed, uint32_t stream_id,
                             uint32_t row0, int B, void* stream) {
    T2S_REQUIRE(x && eps_u && coef, "t2s_ddpm_step: NULL argument");
    T2S_REQUIRE(B > 0 && t_index >= 0, "t2s_ddpm_step: B=%d t_index=%d", B, t_index);
    StepArgs a{};
    a.x = x; a.eps_u = eps_u; a.eps_c = eps_c; a.noise = noise; a.coef = coef; a.step_ptr = nullptr;
    a.t_index = t_index; a.cfg = cfg; a.seed = seed; a.stream_id = stream_id; a.row0 = row0; a.B = B;
    const int total = B * (LAT / 4);
    ddpm_step_kernel<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(a);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

extern "C" int t2s_rf_step(float* x, const float* v_u, const float* v_c, float cfg, float dt, int B, void* stream) {
    T2S_REQUIRE(x && v_u && B > 0, "t2s_rf_step: bad argument");
    const int n4 = B * (LAT / 4);
    rf_step_kernel<<<(n4 + 255) / 256, 256, 0, (hipStream_t)stream>>>(x, v_u, v_c, cfg, dt, n4, nullptr);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

extern "C" int t2s_ddpm_q_sample_n(const float* x0, const float* eps, const int32_t* t, const float* sqrt_ab,
                                   const float* sqrt_1mab, float* out, int B, int row_elems, int n_steps, void* stream) {
    T2S_REQUIRE(x0 && eps && t && sqrt_ab && sqrt_1mab && out && B > 0 && n_steps > 0, "t2s_ddpm_q_sample: bad argument");
    T2S_REQUIRE(row_elems > 0 && row_elems % 4 == 0, "t2s_ddpm_q_sample: row_elems=%d must be a positive multiple of 4", row_elems);
    const int total = B * (row_elems / 4);
    q_sample_kernel<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(x0, eps, t, sqrt_ab, sqrt_1mab, out, B, row_elems / 4, n_steps);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

extern "C" int t2s_ddpm_q_sample(const float* x0, const float* eps, const int32_t* t, const float* sqrt_ab,
                                 const float* sqrt_1mab, float* out, int B, int n_steps, void* stream) {
    return t2s_ddpm_q_sample_n(x0, eps, t, sqrt_ab, sqrt_1mab, out, B, LAT, n_steps, stream);
}

extern "C" int t2s_ddpm_p_sample_n(const float* xt, const float* eps_hat, const int32_t* t, const float* noise,
                                   const float* coef, float* out, int B, int row_elems, int n_steps, void* stream) {
    T2S_REQUIRE(xt && eps_hat && t && noise && coef && out && B > 0 && n_steps > 0, "t2s_ddpm_p_sample: bad argument");
    T2S_REQUIRE(row_elems > 0 && row_elems % 4 == 0, "t2s_ddpm_p_sample: row_elems=%d must be a positive multiple of 4", row_elems);
    const int total = B * (row_elems / 4);
    p_sample_rows_kernel<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(xt, eps_hat, t, noise, coef, out, B, row_elems / 4, n_steps);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

extern "C" int t2s_ddpm_p_sample(const float* xt, const float* eps_hat, const int32_t* t, const float* noise,
                                 const float* coef, float* out, int B, int n_steps, void* stream) {
    return t2s_ddpm_p_sample_n(xt, eps_hat, t, noise, coef, out, B, LAT, n_steps, stream);
}

extern "C" int t2s_mse_ws(const float* a, const float* b, float* out, uint64_t n, float* scratch, void* stream) {
    T2S_REQUIRE(a && b && out && scratch && n > 0, "t2s_mse_ws: bad argument");
    const size_t n4 = n / 4;
    const int wgs = (int)(n4 / 1024 < 1 ? 1 : (n4 / 1024 > MSE_MAX_WGS ? MSE_MAX_WGS : n4 / 1024));   // >= 4 float4 per thread
    mse_partial_kernel<<<wgs, 256, 0, (hipStream_t)stream>>>(a, b, scratch, (size_t)n);
    T2S_LAUNCH_CHECK();
    mse_final_kernel<<<1, 256, 0, (hipStream_t)stream>>>(scratch, wgs, out, (size_t)n);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

// t2s_mse keeps its scratch-free signature: the library lends one scratch per (device, stream), allocated the first time that
// stream calls (so: not capturable THEN; later calls on the stream are).  Calls on one stream are ordered by the stream,
// calls on different streams use different scratch.
extern "C" int t2s_mse(const float* a, const float* b, float* out, uint64_t n, void* stream) {
    T2S_REQUIRE(a && b && out && n > 0, "t2s_mse: bad argument");
    static std::mutex guard;
    static std::vector<std::pair<std::pair<int, void*>, float*>> lent;
    int dev = 0;
    T2S_HIP_CHECK(hipGetDevice(&dev));
    float* scratch = nullptr;
    {
        std::lock_guard<std::mutex> lock(guard);
        for (auto& e : lent)
            if (e.first.first == dev && e.first.second == stream) scratch = e.second;
        if (!scratch) {
            T2S_HIP_CHECK(hipMalloc((void**)&scratch, T2S_MSE_SCRATCH_FLOATS * sizeof(float)));
            lent.push_back({{dev, stream}, scratch});
        }
    }
    return t2s_mse_ws(a, b, out, n, scratch, stream);
}

extern "C" int t2s_rf_create_flow(const float* x1, const float* x0, const float* t, float* out, int B, void* stream) {
    T2S_REQUIRE(x1 && x0 && t && out && B > 0, "t2s_rf_create_flow: bad argument");
    const int total = B * (LAT / 4);
    create_flow_kernel<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(x1, x0, t, out, B);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

// ---------------------------------------------------------------- fused sampling loop
struct t2s_sampler {
    t2s_dit* dit = nullptr;
    t2s_vae* vae = nullptr;
    t2s_sample_config cfg{};
    float* temb_table = nullptr;  // (steps,128)
    float* coef = nullptr;        // (steps,3)  DDPM only
    float* eps_u = nullptr;       // (B,1920)
    float* eps_c = nullptr;
    float* tvals = nullptr;       // (steps)
    float* mod_table = nullptr;   // (steps, batch + 1, 3072): the adaLN modulation of every step, refreshed at the start of a run
    int* step = nullptr;          // device loop indices, one per lane (16 ints apart)
    // Lanes: the rows of the batch are independent through the whole loop, so the batch can run as TWO half
    // batches, each a complete chain (own step counter, own hipGraph, own slice of the DiT workspace) on its own
    // stream, joined only before the decode.  One chain alone drains and refills the chip at each of its 9 kernel
    // boundaries per pass and the row-chain kernel quantises to 7.5 tiles per SIMD at 512 sequences; the other
    // lane's kernels fill those holes.  Results are bitwise those of one lane (batch-invariant kernels).
    int lanes_req = 0;            // 0 = automatic, 1 .. MAX_LANES (t2s_sampler_set_lanes)
    int lanes_cap = 0;            // lanes the graphs below were captured for
    int whole_req = -1;           // t2s_sampler_set_loop_graph: 1 = the WHOLE loop is one graph per lane, 0 = one step, -1 = default
    int whole_cap = 0;            // what the graphs below hold
    static constexpr int MAX_LANES = 4;
    hipGraph_t graph[MAX_LANES] = {};
    hipGraphExec_t exec[MAX_LANES] = {};
    hipStream_t side[MAX_LANES] = {};          // lanes 1 .. (index 0 unused: lane 0 runs on the caller's stream); BORROWED
                                               // from the per-device pool below, never destroyed by a sampler
    hipEvent_t ev_fork = nullptr, ev_join[MAX_LANES] = {};
    // Runs with more than one lane -- and graph runs on the default stream (stream == NULL at the C ABI), which cannot be
    // captured -- go through the library's pooled streams (lane 0 on pool stream 0), forked from and joined to the
    // caller's stream by events inside the call
    hipStream_t own = nullptr;    // borrowed from the pool
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    // pointers the captured graphs were built for
    float* g_x = nullptr;
    const float* g_text = nullptr;
    const float* g_noise = nullptr;
};

namespace {

// The lanes' streams come from ONE pool per device, created together the first time a sampler needs it.  HIP maps
// streams onto a few hardware queues (four by default) in creation order, and two lanes whose streams share a queue run
// one after the other: with a capture stream per Sampler object (the Python wrapper's, then) lane 0 landed on lane 1's
// queue for every fourth sampler a process built -- 52 instead of 61 series/s at 64 series, 58 instead of 62 at 96,
// reproducibly by construction order (tools/strong_probe.py).  Four streams created back to back take four different
// queues, and because lane 0 runs on pool stream 0 too, the caller's stream -- whatever queue it sits on -- only carries
// the fork and the join ...  (GPU_MAX_HW_QUEUES=8 / 16 did not help and cost 3-5 % at 128 / 256.)
// ... in THEORY: measured, pool streams 0 and 2 of four created back to back still shared a queue (96 series as three
// lanes: 57.9 against 62.3 series/s).  So the pool is CALIBRATED once per device: a 400 us spin kernel on each of two
// streams tells whether they overlap (the second started before the first ended); candidates are created until four
// mutually concurrent streams are found (at most 12 candidates, ~10-40 ms once per process).
__global__ void spin_kernel(unsigned long long ticks, unsigned long long* stamp) {
    const unsigned long long t0 = wall_clock64();      // constant 100 MHz
    if (threadIdx.x == 0) stamp[0] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) stamp[1] = wall_clock64();
}

bool streams_overlap(hipStream_t a, hipStream_t b, unsigned long long* stamps_dev) {
    // 400 us spins, and a second look before a pair is declared serial: the test needs the host to issue the second launch
    // while the first kernel still runs, and on a node where eight ranks calibrate at once a host hiccup of 100 us would
    // have rejected a perfectly concurrent pair (the lane then shares a queue: correct, but the lanes no longer overlap)
    for (int attempt = 0; attempt < 2; ++attempt) {
        unsigned long long h[4] = {0, 0, 0, 0};
        spin_kernel<<<1, 64, 0, a>>>(40000ull, stamps_dev);          // 400 us at the 100 MHz wall clock
        spin_kernel<<<1, 64, 0, b>>>(40000ull, stamps_dev + 2);
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return false;
        // (read back on stream a, not with a legacy-stream hipMemcpy: see t2s_sampler_create)
        if (hipMemcpyAsync(h, stamps_dev, sizeof(h), hipMemcpyDeviceToHost, a) != hipSuccess || hipStreamSynchronize(a) != hipSuccess) return false;
        if (h[2] < h[1] && h[0] < h[3]) return true;
    }
    return false;
}

int g_lane_pool_concurrent[16] = {};      // per device: mutually concurrent streams the calibration found

// The pool is shared by every sampler of the process on a device, lane 0 included: two host threads driving two samplers
// would capture, record events and launch graphs on the SAME streams at once (one thread's work pulled into the other's
// capture, or hipErrorStreamCaptureIsolation).  A run that uses pool streams holds this lock from its first event to its
// join: the enqueue of a run is host work of a few ms, the GPU side stays asynchronous.
std::recursive_mutex g_pool_use[16];     // recursive: a failing t2s_sampler_create destroys its half-built sampler under the lock

// one non-blocking stream per device for the library's own set-up work (t2s_sampler_create); see there
hipStream_t setup_stream(int dev) {
    static hipStream_t st[16] = {};
    static std::mutex guard;
    std::lock_guard<std::mutex> lock(guard);
    if (dev < 0 || dev >= 16) return nullptr;
    if (!st[dev] && hipStreamCreateWithFlags(&st[dev], hipStreamNonBlocking) != hipSuccess) {
        st[dev] = nullptr;
        (void)hipGetLastError();
    }
    return st[dev];
}

hipStream_t* lane_streams() {
    constexpr int ML = t2s_sampler::MAX_LANES;
    static hipStream_t pool[16][ML] = {};
    static bool ready[16] = {}, failed[16] = {};
    static std::mutex guard;                    // the pool is process-wide: samplers of different threads may meet here
    std::lock_guard<std::mutex> lock(guard);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (ready[dev]) return pool[dev];
    if (failed[dev]) return nullptr;            // calibrated once per process: a device without streams stays on one lane
    unsigned long long* stamps = nullptr;
    if (hipMalloc((void**)&stamps, 4 * sizeof(unsigned long long)) != hipSuccess) return nullptr;
    int have = 0;
    hipStream_t spare[12] = {};
    int n_spare = 0;
    for (int c = 0; c < 12 && have < ML; ++c) {
        hipStream_t cand = nullptr;
        if (hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) break;
        bool ok = true;
        for (int l = 0; l < have && ok; ++l) ok = streams_overlap(pool[dev][l], cand, stamps);
        if (ok) pool[dev][have++] = cand;
        else spare[n_spare++] = cand;
    }
    g_lane_pool_concurrent[dev] = have;
    // fewer than ML concurrent queues on this device / configuration: the remaining lanes share (correct, only slower)
    for (int i = 0; have < ML && i < n_spare; ++i) pool[dev][have++] = spare[i], spare[i] = nullptr;
    for (int i = 0; i < n_spare; ++i)
        if (spare[i]) (void)hipStreamDestroy(spare[i]);
    (void)hipFree(stamps);
    (void)hipGetLastError();
    if (have < ML) {                            // stream creation failed part-way: give back what was made, do not retry
        for (int i = 0; i < have; ++i) (void)hipStreamDestroy(pool[dev][i]), pool[dev][i] = nullptr;
        g_lane_pool_concurrent[dev] = 0;
        failed[dev] = true;
        (void)hipGetLastError();
        return nullptr;
    }
    ready[dev] = true;
    return pool[dev];
}

// one loop iteration of rows [r0, r0 + n) of the batch on stream st, stepping the lane's counter
int enqueue_step(t2s_sampler* s, float* x, const float* text, const float* noise, hipStream_t st, int lane, int r0,
                 int n) {
    const t2s_sample_config& c = s->cfg;
    int* step = s->step + 16 * lane;
    float* xl = x + (size_t)r0 * LAT;
    float* eu = s->eps_u + (size_t)r0 * LAT;
    float* ec = s->eps_c + (size_t)r0 * LAT;
    int rc = dit_forward_cfg_step(s->dit, xl, s->temb_table, step, text + (size_t)r0 * D, eu, ec, n, st, 2 * r0,
                                  s->mod_table, c.batch + 1, r0);
    if (rc != T2S_OK) return rc;
    // T2S_SKIP_UPDATE=1 (TIMING ONLY, results invalid: the state and the loop index never advance): the sampler without its
    // update launch -- the upper bound of what fusing the DDPM / RF update into the last row kernel could save (VERDICT r04
    // item 7; profiles/EXPERIMENTS.md section 1)
    static const bool skip_update = getenv("T2S_SKIP_UPDATE") && atoi(getenv("T2S_SKIP_UPDATE")) != 0;
    if (skip_update) return T2S_OK;
    if (c.mode == T2S_MODE_DDPM) {
        StepArgs a{};
        a.x = xl; a.eps_u = eu; a.eps_c = ec; a.noise = noise ? noise + (size_t)r0 * LAT : nullptr; a.coef = s->coef;
        a.step_ptr = step; a.steps = c.steps; a.cfg = c.cfg_scale; a.seed = c.seed; a.row0 = c.row0 + (uint32_t)r0;
        a.B = n; a.noise_rows = c.batch; a.advance = step;
        const int total = n * (LAT / 4), wgs = (total + 255) / 256;
        ddpm_step_kernel<<<wgs < STEP_MAX_WGS ? wgs : STEP_MAX_WGS, 256, 0, st>>>(a);
    } else {
        const int n4 = n * (LAT / 4), wgs = (n4 + 255) / 256;
        rf_step_kernel<<<wgs < STEP_MAX_WGS ? wgs : STEP_MAX_WGS, 256, 0, st>>>(xl, eu, ec, c.cfg_scale, 1.0f / (float)c.steps, n4, step);
    }
    T2S_LAUNCH_CHECK();   // (the update kernel's last workgroup advanced the lane's loop index)
    return T2S_OK;
}

// Whole-loop graphs by default?  Same-box A/B (tools/ab_loop_graph.sh, profiles/r04_loop_graph_ab.txt), series/s one-step /
// whole-loop: rectified flow 100 steps at B = 1024 637.4 / 636.3 (-0.2 %, noise), at B = 32 569.3 / 574.7 (+0.9 %: the host's
// 200 graph launches per run are on the critical path of a 56 ms run), DDPM 1000 steps at B = 256 63.70 / 63.75.  So: the
// whole loop as ONE graph per lane for the step counts the authors sample with (10 - 100, scripts/script.sh), where it is
// never slower and a capture is <= 2,560 nodes; the one-step graph beyond (a 1000-step loop would be 10,000 nodes per
// lane for nothing).
inline int loop_graph_default(int steps) { return steps <= 256; }

void drop_graph(t2s_sampler* s) {
    for (int l = 0; l < t2s_sampler::MAX_LANES; ++l) {
        if (s->exec[l]) (void)hipGraphExecDestroy(s->exec[l]);
        if (s->graph[l]) (void)hipGraphDestroy(s->graph[l]);
        s->exec[l] = nullptr;
        s->graph[l] = nullptr;
    }
    s->lanes_cap = 0;
}

// lanes of this run: as asked for, or automatically EQUAL shares of whole 32-row groups: two for a batch that is a
// multiple of 64 (and 16 + 16 for 32 series), three for 96.  Measured with the round-3 kernels (tools/strong_probe.py,
// series/s with one lane / the automatic choice): 256: 60.1 / 64.3, 192: 59.9 / 63.5, 128: 58.7 / 62.8, 96: 58.3 / 62.3,
// 64: 57.4 / 61.5, 32: 52.1 / 58.4 -- at the small sizes a launch is mostly fixed cost, which the other lanes' kernels
// cover.  Unequal lanes go either way (224 = 128 + 96: 60.1 / 62.6; 160 = 96 + 64: 59.4 / 57.7; 96 = 64 + 32: 55.6 when the
// two lanes run DIFFERENT attention kernels -- a persistent workgroup fills a CU's registers and the other lane's packed
// workgroups wait for it) and splits without whole groups lose (48: 47.3 / 39.8, 16: 44.1 / 37.7), so those stay on one
// lane; three lanes 61.6 and four 60.4 at 256, three 61.2 at 192.
int pick_lanes(const t2s_sampler* s, bool trace) {
    if (trace || s->cfg.batch < 2) return 1;
    const int B = s->cfg.batch;
    int lanes = (B % 64 == 0 || B == 32) ? 2 : (B == 96 ? 3 : 1);
    if (const char* e = getenv("T2S_SAMPLER_LANES")) lanes = atoi(e);
    if (s->lanes_req) lanes = s->lanes_req;
    lanes = lanes < 1 ? 1 : (lanes > t2s_sampler::MAX_LANES ? t2s_sampler::MAX_LANES : lanes);
    return lanes < s->cfg.batch ? lanes : s->cfg.batch;
}

}  // namespace

extern "C" int t2s_dit_max_seqs(const t2s_dit* h);

extern "C" int t2s_sampler_create(t2s_dit* dit, t2s_vae* vae, const t2s_sample_config* cfg, t2s_sampler** out) {
    T2S_REQUIRE(dit && cfg && out, "t2s_sampler_create: NULL argument");
    T2S_REQUIRE(cfg->mode == T2S_MODE_DDPM || cfg->mode == T2S_MODE_RF, "t2s_sampler_create: mode=%d", cfg->mode);
    T2S_REQUIRE(cfg->steps > 0 && cfg->steps <= 100000, "t2s_sampler_create: steps=%d", cfg->steps);
    T2S_REQUIRE(cfg->batch > 0 && 2 * cfg->batch <= t2s_dit_max_seqs(dit),
                "t2s_sampler_create: batch=%d needs 2*batch <= dit max_seqs=%d", cfg->batch, t2s_dit_max_seqs(dit));
    T2S_REQUIRE(cfg->t_values, "t2s_sampler_create: t_values is NULL");
    T2S_REQUIRE(cfg->mode != T2S_MODE_DDPM || cfg->ddpm_coef, "t2s_sampler_create: DDPM needs ddpm_coef");
    T2S_REQUIRE(!vae || (cfg->length >= 4 && cfg->length % 4 == 0 && cfg->length <= (1 << 20)),
                "t2s_sampler_create: length=%d unsupported", cfg->length);
    // allocations, synchronous copies and a stream synchronisation follow: not while another thread's run has a capture open
    // on the pool streams (same lock as t2s_sampler_run; see include/t2s.h "Threads")
    int cur_dev = 0;
    T2S_HIP_CHECK(hipGetDevice(&cur_dev));
    T2S_REQUIRE(cur_dev >= 0 && cur_dev < 16, "t2s_sampler_create: device %d", cur_dev);
#ifndef T2S_DIAG_UNSERIALISED   // (diagnosis build of tools/stress_threads.py --unserialised: round 4's locking)
    std::lock_guard<std::recursive_mutex> pool_lock(g_pool_use[cur_dev]);
#endif
    t2s_sampler* s = new t2s_sampler();
    s->dit = dit; s->vae = vae; s->cfg = *cfg;
    s->cfg.ddpm_coef = nullptr; s->cfg.t_values = nullptr;  // host pointers are not retained
    const size_t B = (size_t)cfg->batch, T = (size_t)cfg->steps;
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    alloc((void**)&s->temb_table, T * D * sizeof(float));
    alloc((void**)&s->coef, T * 3 * sizeof(float));
    alloc((void**)&s->eps_u, B * LAT * sizeof(float));
    alloc((void**)&s->eps_c, B * LAT * sizeof(float));
    alloc((void**)&s->tvals, T * sizeof(float));
    alloc((void**)&s->step, t2s_sampler::MAX_LANES * 16 * sizeof(int));   // one counter per lane, 64 B apart
    // whole-run adaLN table (dit_adaln_table; 3.2 GB at 256 series x 1000 steps for ~2 % of a step): only while it is a
    // small part of what the device has FREE right now -- at most 1/8 of it and 16 GB -- so that a process holding several
    // samplers, or sharing the GPU with torch's allocator or a training job, does not run dry later for an optimisation;
    // beyond that the per-step kernel stays in the loop (same bits).  T2S_ADALN_TABLE=0 switches the table off.
    const size_t table_bytes = T * (B + 1) * (size_t)MODROW * sizeof(float);
    const char* table_env = getenv("T2S_ADALN_TABLE");   // =0: keep the per-step adaLN kernel (A/B, and the test of that path)
    size_t mem_free = 0, mem_total = 0;
    if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) mem_free = 0, (void)hipGetLastError();
    if (e == hipSuccess && !(table_env && atoi(table_env) == 0) && table_bytes <= ((size_t)16 << 30) && table_bytes <= mem_free / 8 &&
        hipMalloc((void**)&s->mod_table, table_bytes) != hipSuccess) {
        s->mod_table = nullptr;       // an optimisation only: a device too full for it keeps the per-step kernel
        (void)hipGetLastError();
    }
    // Uploads and the table kernel run on a NON-BLOCKING stream of the library's own, never on the legacy default stream: an
    // operation on the legacy stream implicitly joins every blocking stream of the device, and HIP refuses it
    // (hipErrorStreamCaptureImplicit, "operation would make the legacy stream depend on a capturing blocking stream") while
    // ANY thread has a capture open -- invalidating that capture.  That was the round-4 two-thread failure (reproduced in
    // round 5 with the serialisation taken away, tools/stress_threads.py --unserialised).
    hipStream_t setup = setup_stream(cur_dev);
    if (e == hipSuccess && !setup) e = hipErrorUnknown;
    if (e == hipSuccess) e = hipMemcpyAsync(s->tvals, cfg->t_values, T * sizeof(float), hipMemcpyHostToDevice, setup);
    if (e == hipSuccess && cfg->mode == T2S_MODE_DDPM)
        e = hipMemcpyAsync(s->coef, cfg->ddpm_coef, T * 3 * sizeof(float), hipMemcpyHostToDevice, setup);
    if (e == hipSuccess) e = hipStreamSynchronize(setup);      // the host tables may go away when the call returns
    if (e != hipSuccess) {
        set_error("t2s_sampler_create: allocation/upload failed: %s", hipGetErrorString(e));
        t2s_sampler_destroy(s);
        return T2S_E_HIP;
    }
    // time-embedding table for every loop index (transformer.py:30-40 applied to t_values)
    int rc = t2s_time_embedding(dit, s->tvals, s->temb_table, cfg->steps, setup);
    if (rc == T2S_OK && hipStreamSynchronize(setup) != hipSuccess) {
        set_error("t2s_sampler_create: time-embedding table failed");
        rc = T2S_E_HIP;
    }
    if (rc != T2S_OK) {
        t2s_sampler_destroy(s);
        return rc;
    }
    // calibrate the lane-stream pool HERE (create synchronises anyway): t2s_sampler_run then never allocates, copies or
    // synchronises for it -- legal while the calling thread has a capture open, and its timing test runs on an idle stream
    (void)lane_streams();
    *out = s;
    return T2S_OK;
}

extern "C" int t2s_sampler_set_lanes(t2s_sampler* s, int lanes) {
    T2S_REQUIRE(s && lanes >= 0 && lanes <= t2s_sampler::MAX_LANES, "t2s_sampler_set_lanes: lanes=%d (0 = automatic, 1 .. %d)", lanes,
                t2s_sampler::MAX_LANES);
    s->lanes_req = lanes;
    return T2S_OK;
}

extern "C" int t2s_sampler_set_loop_graph(t2s_sampler* s, int whole_loop) {
    T2S_REQUIRE(s && (whole_loop == 0 || whole_loop == 1 || whole_loop == -1), "t2s_sampler_set_loop_graph: %d (1 whole loop, 0 one step, -1 default)", whole_loop);
    s->whole_req = whole_loop;
    return T2S_OK;
}

extern "C" int t2s_sampler_set_row0(t2s_sampler* s, uint32_t row0) {
    T2S_REQUIRE(s, "t2s_sampler_set_row0: NULL sampler");
    s->cfg.row0 = row0;   // uploaded next to the step counters at the start of every run: no re-capture
    return T2S_OK;
}

extern "C" int t2s_sampler_lane_pool(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
    return g_lane_pool_concurrent[dev];
}

extern "C" int t2s_sampler_graph_lanes(const t2s_sampler* s) { return (s && s->exec[0]) ? s->lanes_cap : 0; }

extern "C" void t2s_sampler_destroy(t2s_sampler* s) {
    if (!s) return;
    int cur_dev = 0;
    std::unique_lock<std::recursive_mutex> pool_lock;  // hipFree synchronises the device: not inside another thread's capture
#ifndef T2S_DIAG_UNSERIALISED
    if (hipGetDevice(&cur_dev) == hipSuccess && cur_dev >= 0 && cur_dev < 16) pool_lock = std::unique_lock<std::recursive_mutex>(g_pool_use[cur_dev]);
#endif
    drop_graph(s);
    for (int l = 0; l < t2s_sampler::MAX_LANES; ++l)
        if (s->ev_join[l]) (void)hipEventDestroy(s->ev_join[l]);
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    if (s->ev_in) (void)hipEventDestroy(s->ev_in);
    if (s->ev_out) (void)hipEventDestroy(s->ev_out);
    void* bufs[] = {s->temb_table, s->coef, s->eps_u, s->eps_c, s->tvals, s->step, s->mod_table};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    delete s;
}

extern "C" int t2s_vae_decode(t2s_vae* h, const float* z, float* recon, float* after, int B, int L, void* stream);

extern "C" int t2s_sampler_run(t2s_sampler* s, float* x, const float* text, const float* noise, float* series,
                               float* trace0, void* stream) {
    T2S_REQUIRE(s && x && text, "t2s_sampler_run: NULL argument");
    T2S_REQUIRE(!series || s->vae, "t2s_sampler_run: series requested but the sampler has no VAE");
    T2S_REQUIRE(!trace0 || s->vae, "t2s_sampler_run: trace requested but the sampler has no VAE");
    hipStream_t st = (hipStream_t)stream;
    const t2s_sample_config& c = s->cfg;
    int rc;
    int lanes = pick_lanes(s, trace0 != nullptr);
    const bool graph_ok = c.use_graph && !trace0;
    // Held to the end of the call by EVERY run that opens a stream capture or touches the pool streams -- a single-lane run
    // capturing on the caller's own stream included: another thread's t2s_sampler_create / _destroy (allocations, synchronous
    // copies, synchronisation) and the pool's calibration are then never concurrent with an open capture of this library.
    std::unique_lock<std::recursive_mutex> pool_lock;
    if (graph_ok || lanes > 1) {
        int dev = 0;
        T2S_HIP_CHECK(hipGetDevice(&dev));
        T2S_REQUIRE(dev >= 0 && dev < 16, "t2s_sampler_run: device %d", dev);
        pool_lock = std::unique_lock<std::recursive_mutex>(g_pool_use[dev]);
    }
    if (lanes > 1 && !lane_streams()) lanes = 1;      // no stream pool on this device: one chain, same results
    // the default stream cannot be captured (never a silent eager run), and several lanes run on streams created TOGETHER
    // (distinct hardware queues): the caller's stream then only carries the fork and the join
    hipStream_t const caller = st;
    const bool via_own = (graph_ok && st == nullptr) || lanes > 1;
    if (via_own) {
        if (!s->own) {
            hipStream_t* pool = lane_streams();
            T2S_REQUIRE(pool, "t2s_sampler_run: cannot create the lane streams");
            s->own = pool[0];
            T2S_HIP_CHECK(hipEventCreateWithFlags(&s->ev_in, hipEventDisableTiming));
            T2S_HIP_CHECK(hipEventCreateWithFlags(&s->ev_out, hipEventDisableTiming));
        }
        T2S_HIP_CHECK(hipEventRecord(s->ev_in, caller));
        T2S_HIP_CHECK(hipStreamWaitEvent(s->own, s->ev_in, 0));
        st = s->own;
    }
    if (lanes > 1 && !s->ev_fork) T2S_HIP_CHECK(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
    for (int l = 1; l < lanes; ++l)
        if (!s->side[l]) {
            hipStream_t* pool = lane_streams();
            T2S_REQUIRE(pool, "t2s_sampler_run: cannot create the lane streams");
            s->side[l] = pool[l];
            T2S_HIP_CHECK(hipEventCreateWithFlags(&s->ev_join[l], hipEventDisableTiming));
        }
    // lane l steps rows [r0[l], r0[l] + nr[l]) on lst[l]
    // split point: half the batch, rounded to a multiple of 32 rows when both lanes keep >= 32 -- the persistent
    // attention kernel walks 8 * rows / n_cu (sequence, head) items per CU, a whole number only in steps of 32 rows
    // (a 144 + 112 split of 256 measured 12 % SLOWER than one lane, 128 + 128 or 160 + 96 4 % faster)
    // (more lanes: equal shares, in whole 32-row groups when every lane keeps at least one)
    constexpr int ML = t2s_sampler::MAX_LANES;
    int r0[ML + 1] = {}, nr[ML] = {};
    {
        const int share = (c.batch + lanes - 1) / lanes, share32 = (share + 31) / 32 * 32;
        const int step = (c.batch - (lanes - 1) * share32 >= 32) ? share32 : share;
        for (int l = 0; l <= lanes; ++l) r0[l] = l * step < c.batch ? l * step : c.batch;
        r0[lanes] = c.batch;
        for (int l = 0; l < lanes; ++l) nr[l] = r0[l + 1] - r0[l];
        while (lanes > 1 && nr[lanes - 1] == 0) --lanes;      // a batch too small for its last lane
    }
    hipStream_t lst[ML] = {st, s->side[1], s->side[2], s->side[3]};
    // One graph per lane holds either ONE step (replayed `steps` times from the host) or the WHOLE loop (steps x 10 kernel
    // nodes, launched once): every node reads its loop index from the lane's device counter, so the two are the same
    // kernels in the same order.  Default: see loop_graph_default().
    int whole = s->whole_req;
    if (whole < 0) {
        const char* e = getenv("T2S_SAMPLER_LOOP_GRAPH");
        whole = e ? (atoi(e) != 0) : loop_graph_default(c.steps);
    }
    if (graph_ok && (!s->exec[0] || s->lanes_cap != lanes || s->whole_cap != whole || s->g_x != x || s->g_text != text ||
                     s->g_noise != noise)) {
        drop_graph(s);
        for (int l = 0; l < lanes; ++l) {
            T2S_HIP_CHECK(hipStreamBeginCapture(lst[l], hipStreamCaptureModeThreadLocal));
            rc = T2S_OK;
            for (int j = 0; j < (whole ? c.steps : 1) && rc == T2S_OK; ++j) rc = enqueue_step(s, x, text, noise, lst[l], l, r0[l], nr[l]);
            hipError_t e = hipStreamEndCapture(lst[l], &s->graph[l]);
            if (rc != T2S_OK) {
                drop_graph(s);
                return rc;
            }
            if (e != hipSuccess) {
                set_error("t2s_sampler_run: hipStreamEndCapture failed: %s", hipGetErrorString(e));
                drop_graph(s);
                return T2S_E_HIP;
            }
            T2S_HIP_CHECK(hipGraphInstantiate(&s->exec[l], s->graph[l], nullptr, nullptr, 0));
        }
        s->lanes_cap = lanes;
        s->whole_cap = whole;
        s->g_x = x; s->g_text = text; s->g_noise = noise;
    }
    // the adaLN modulation of every step for this run's text (state-independent: off the loop's critical path)
    if (s->mod_table && (rc = dit_adaln_table(s->dit, s->temb_table, c.steps, text, c.batch, s->mod_table, st)) != T2S_OK) return rc;
    if (lanes > 1) {    // fork: the other lanes start after everything already queued on the caller's stream
        T2S_HIP_CHECK(hipEventRecord(s->ev_fork, st));
        for (int l = 1; l < lanes; ++l) T2S_HIP_CHECK(hipStreamWaitEvent(s->side[l], s->ev_fork, 0));
    }
    for (int l = 0; l < lanes; ++l) {
        set_step_kernel<<<1, 64, 0, lst[l]>>>(s->step + 16 * l, 0, c.row0 + (uint32_t)r0[l]);
        T2S_LAUNCH_CHECK();
    }
    if (graph_ok && whole)
        for (int l = 0; l < lanes; ++l) T2S_HIP_CHECK(hipGraphLaunch(s->exec[l], lst[l]));
    for (int j = 0; j < c.steps && !(graph_ok && whole); ++j) {
        for (int l = 0; l < lanes; ++l) {
            if (graph_ok) {
                T2S_HIP_CHECK(hipGraphLaunch(s->exec[l], lst[l]));
            } else if ((rc = enqueue_step(s, x, text, noise, lst[l], l, r0[l], nr[l])) != T2S_OK) {
                return rc;
            }
        }
        if (trace0) {
            // infer.py:90-93: decode row 0 of the first batch after every step
            if ((rc = t2s_vae_decode(s->vae, x, trace0 + (size_t)j * c.length, nullptr, 1, c.length, st)) != T2S_OK)
                return rc;
        }
    }
    for (int l = 1; l < lanes; ++l) {   // join before the decode (and before anything the caller queues next)
        T2S_HIP_CHECK(hipEventRecord(s->ev_join[l], s->side[l]));
        T2S_HIP_CHECK(hipStreamWaitEvent(st, s->ev_join[l], 0));
    }
    if (series) {
        if ((rc = t2s_vae_decode(s->vae, x, series, nullptr, c.batch, c.length, st)) != T2S_OK) return rc;
    }
    if (via_own) {      // whatever the caller queues on its stream next sees the results
        T2S_HIP_CHECK(hipEventRecord(s->ev_out, st));
        T2S_HIP_CHECK(hipStreamWaitEvent(caller, s->ev_out, 0));
    }
    return T2S_OK;
}
