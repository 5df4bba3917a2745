// Attention for the TRAINING path (plain layouts, fp32 MFMA): forward that also emits the row
// log-sum-exp, and the flash-style backward (timm 1.0.11 Attention core; reference call site
// model/denoiser/transformer.py:116 under autograd, train.py:123-125).
//
// q, k, v: (BH, 480, 32) with bh = seq*4 + head.  o, do: token rows (S*480, 128), head h at
// columns 32h..32h+31.  dqkv: token rows (S*480, 384) = [dq | dk | dv] x heads, i.e. the gradient
// of the qkv linear's output.  lse: (BH, 480) in the log2 domain: m + log2(sum 2^(s - m)) with
// s = q.k * 32^-0.5 * log2(e).
//
// Backward, with P recomputed from lse (no N x N tensor is stored):
//   D_i   = sum_d dO[i][d] O[i][d]
//   dS    = P o (dP - D_i),  dP = dO V^T
//   dQ    = scale * dS K          (kernel A: queries on the lanes, transposed tiles as in forward)
//   dK    = scale * dS^T Q,  dV = P^T dO   (kernel B: keys on the lanes)
// Each kernel keeps two (480 x 32) operands of its (sequence, head) in LDS (row stride 36) and the
// other two in registers / streams them; the f32 accumulator layout is reused directly as the next
// MFMA's B operand (one register per lane), exactly as in the forward kernels.
#include "t2s_common.h"

namespace t2s {

namespace {
constexpr int NKB = NTOK / 32;     // 15
constexpr int STR = 36;            // padded LDS row stride (floats)
constexpr int LDS2 = 2 * NTOK * STR * 4 + 2 * NTOK * 4;   // two operands + two per-row vectors = 142,080 B
constexpr float SCALE = 0.17677669529663687f;
constexpr float QS = SCALE * 1.4426950408889634f;

__device__ __forceinline__ float pair_max_f(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float pair_sum_f(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// stage a (480,32) row-major matrix into LDS with row stride STR
__device__ __forceinline__ void stage_rows(float* dst, const float* src, int src_row_stride, int tid, int nthreads) {
    for (int idx = tid; idx < NTOK * 8; idx += nthreads) {
        const int row = idx >> 3, c4 = idx & 7;
        *reinterpret_cast<f32x4*>(dst + row * STR + c4 * 4) =
            *reinterpret_cast<const f32x4*>(src + (size_t)row * src_row_stride + c4 * 4);
    }
}
}  // namespace

// ------------------------------------------------------------------ forward with lse
__global__ __launch_bounds__(512) void attn_train_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ v, float* __restrict__ o_rows,
                                                             float* __restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + NTOK * STR;
    const int bh = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, i = lane & 31;
    const int seq = bh / NH, head = bh % NH;
    stage_rows(Ks, k + (size_t)bh * NTOK * DH, DH, tid, 512);
    stage_rows(Vs, v + (size_t)bh * NTOK * DH, DH, tid, 512);
    __syncthreads();
    const float* qg = q + (size_t)bh * NTOK * DH;
    for (int qt = wave; qt < NKB; qt += 8) {
        f32x4 qf[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            qf[g] = *reinterpret_cast<const f32x4*>(qg + (size_t)(qt * 32 + i) * DH + 8 * g + 4 * half) * QS;
        f32x16 ot;
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[r] = 0.f;
        float m_run = -INFINITY, l_lane = 0.f;
        for (int jb = 0; jb < NKB; ++jb) {
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
            const float* krow = Ks + (jb * 32 + i) * STR + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(krow + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) st = mfma32(kf[e], qf[g][e], st);
            }
            float mloc = st[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, st[r]);
            mloc = pair_max_f(mloc);
            const float m_new = fmaxf(m_run, mloc);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[r] = __builtin_amdgcn_exp2f(st[r] - m_new);
                ps += st[r];
                ot[r] *= alpha;
            }
            l_lane = l_lane * alpha + ps;
            const float* vrow = Vs + (jb * 32 + 4 * half) * STR + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) ot = mfma32(vrow[((r & 3) + 8 * (r >> 2)) * STR], st[r], ot);
        }
        const float l_tot = pair_sum_f(l_lane);
        const float inv = 1.0f / l_tot;
        const int tok = qt * 32 + i;
        float* orow = o_rows + ((size_t)seq * NTOK + tok) * D + head * DH + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 w = {ot[4 * g] * inv, ot[4 * g + 1] * inv, ot[4 * g + 2] * inv, ot[4 * g + 3] * inv};
            *reinterpret_cast<f32x4*>(orow + 8 * g) = w;
        }
        if (half == 0) lse[(size_t)bh * NTOK + tok] = m_run + __builtin_amdgcn_logf(l_tot);   // v_log_f32 = log2
    }
}

// ------------------------------------------------------------------ D_i = sum_d dO[i][d] * O[i][d]
__global__ __launch_bounds__(256) void attn_dsum_kernel(const float* __restrict__ o_rows, const float* __restrict__ do_rows,
                                                        float* __restrict__ dsum, int M) {
    // one thread per (token row, head): 32 contiguous floats
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * NH) return;
    const int row = idx >> 2, head = idx & 3;
    const f32x4* a = reinterpret_cast<const f32x4*>(o_rows + (size_t)row * D + head * DH);
    const f32x4* b = reinterpret_cast<const f32x4*>(do_rows + (size_t)row * D + head * DH);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f32x4 x = a[c], y = b[c];
        s += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
    }
    const int seq = row / NTOK, tok = row - seq * NTOK;
    dsum[((size_t)seq * NH + head) * NTOK + tok] = s;
}

// ------------------------------------------------------------------ kernel A: dQ (queries on lanes)
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                          const float* __restrict__ v, const float* __restrict__ do_rows,
                                                          const float* __restrict__ lse, const float* __restrict__ dsum,
                                                          float* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + NTOK * STR;
    const int bh = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, i = lane & 31;
    const int seq = bh / NH, head = bh % NH;
    stage_rows(Ks, k + (size_t)bh * NTOK * DH, DH, tid, 512);
    stage_rows(Vs, v + (size_t)bh * NTOK * DH, DH, tid, 512);
    __syncthreads();
    const float* qg = q + (size_t)bh * NTOK * DH;
    for (int qt = wave; qt < NKB; qt += 8) {
        const int tok = qt * 32 + i;
        f32x4 qf[4], dof[4];   // B operands: Q^T (pre-scaled, log2 domain) and dO^T of this lane's query
        const float* dorow = do_rows + ((size_t)seq * NTOK + tok) * D + head * DH + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            qf[g] = *reinterpret_cast<const f32x4*>(qg + (size_t)tok * DH + 8 * g + 4 * half) * QS;
            dof[g] = *reinterpret_cast<const f32x4*>(dorow + 8 * g);
        }
        const float lse_i = lse[(size_t)bh * NTOK + tok];
        const float d_i = dsum[(size_t)bh * NTOK + tok];
        f32x16 dq;
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[r] = 0.f;
        for (int jb = 0; jb < NKB; ++jb) {
            f32x16 st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = dp[r] = 0.f;
            const float* krow = Ks + (jb * 32 + i) * STR + 4 * half;
            const float* vrw = Vs + (jb * 32 + i) * STR + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(krow + 8 * g);
                const f32x4 vf = *reinterpret_cast<const f32x4*>(vrw + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st = mfma32(kf[e], qf[g][e], st);     // S^T[key][query]  (log2 domain)
                    dp = mfma32(vf[e], dof[g][e], dp);    // dP^T[key][query] = V dO^T
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(st[r] - lse_i);
                st[r] = p * (dp[r] - d_i);                // dS^T
            }
            // dQ^T[d][query] += K^T[d][key] dS^T[key][query]: A = K column d over the key pair of step r
            const float* kcol = Ks + (jb * 32 + 4 * half) * STR + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) dq = mfma32(kcol[((r & 3) + 8 * (r >> 2)) * STR], st[r], dq);
        }
        float* dst = dqkv + ((size_t)seq * NTOK + tok) * (3 * D) + head * DH + 4 * half;   // dq block
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 w = {dq[4 * g] * SCALE, dq[4 * g + 1] * SCALE, dq[4 * g + 2] * SCALE, dq[4 * g + 3] * SCALE};
            *reinterpret_cast<f32x4*>(dst + 8 * g) = w;
        }
    }
}

// ------------------------------------------------------------------ kernel B: dK, dV (keys on lanes)
__global__ __launch_bounds__(512) void attn_bwd_dkv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const float* __restrict__ do_rows,
                                                           const float* __restrict__ lse, const float* __restrict__ dsum,
                                                           float* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;                       // (480, STR) unscaled Q
    float* Os = smem + NTOK * STR;          // (480, STR) dO of this head
    float* Ls = smem + 2 * NTOK * STR;      // lse (log2 domain)
    float* Ds = Ls + NTOK;                  // D_i
    const int bh = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    const int seq = bh / NH, head = bh % NH;
    stage_rows(Qs, q + (size_t)bh * NTOK * DH, DH, tid, 512);
    stage_rows(Os, do_rows + (size_t)seq * NTOK * D + head * DH, D, tid, 512);
    for (int t = tid; t < NTOK; t += 512) {
        Ls[t] = lse[(size_t)bh * NTOK + t];
        Ds[t] = dsum[(size_t)bh * NTOK + t];
    }
    __syncthreads();
    const float* kg = k + (size_t)bh * NTOK * DH;
    const float* vg = v + (size_t)bh * NTOK * DH;
    for (int kb = wave; kb < NKB; kb += 8) {
        const int key = kb * 32 + j;
        f32x4 kf[4], vf[4];   // B operands: K^T (pre-scaled to the log2 domain) and V^T of this lane's key
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            kf[g] = *reinterpret_cast<const f32x4*>(kg + (size_t)key * DH + 8 * g + 4 * half) * QS;
            vf[g] = *reinterpret_cast<const f32x4*>(vg + (size_t)key * DH + 8 * g + 4 * half);
        }
        f32x16 dk, dv;
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[r] = dv[r] = 0.f;
        for (int qb = 0; qb < NKB; ++qb) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
            const float* qrow = Qs + (qb * 32 + j) * STR + 4 * half;    // A operand row = query (lane index j)
            const float* orow = Os + (qb * 32 + j) * STR + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 qa = *reinterpret_cast<const f32x4*>(qrow + 8 * g);
                const f32x4 oa = *reinterpret_cast<const f32x4*>(orow + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s = mfma32(qa[e], kf[g][e], s);      // S[query][key]   (registers = queries, lane = key)
                    dp = mfma32(oa[e], vf[g][e], dp);    // dP[query][key] = dO V^T
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qi = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;   // query of register r in this half
                const float p = __builtin_amdgcn_exp2f(s[r] - Ls[qi]);
                dp[r] = p * (dp[r] - Ds[qi]);   // dS[query][key]
                s[r] = p;                       // P[query][key]
            }
            // dV^T[d][key] += dO^T[d][query] P[query][key] ; dK^T[d][key] += Q^T[d][query] dS[query][key]
            const float* ocol = Os + (qb * 32 + 4 * half) * STR + j;    // lane index j = feature d here
            const float* qcol = Qs + (qb * 32 + 4 * half) * STR + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = ((r & 3) + 8 * (r >> 2)) * STR;
                dv = mfma32(ocol[ro], s[r], dv);
                dk = mfma32(qcol[ro], dp[r], dk);
            }
        }
        float* dst = dqkv + ((size_t)seq * NTOK + key) * (3 * D) + head * DH + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 wk = {dk[4 * g] * SCALE, dk[4 * g + 1] * SCALE, dk[4 * g + 2] * SCALE, dk[4 * g + 3] * SCALE};
            const f32x4 wv = {dv[4 * g], dv[4 * g + 1], dv[4 * g + 2], dv[4 * g + 3]};
            *reinterpret_cast<f32x4*>(dst + D + 8 * g) = wk;
            *reinterpret_cast<f32x4*>(dst + 2 * D + 8 * g) = wv;
        }
    }
}

static int attn_train_init() {
    static bool done = false;
    if (!done) {
        T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_train_fwd_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, LDS2));
        T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, LDS2));
        T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, LDS2));
        done = true;
    }
    return T2S_OK;
}

int attn_plain_train_fwd(const float* q, const float* k, const float* v, float* o_rows, float* lse, int BH,
                         hipStream_t st) {
    if (int rc = attn_train_init()) return rc;
    attn_train_fwd_kernel<<<BH, 512, LDS2, st>>>(q, k, v, o_rows, lse);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

int attn_bwd(const float* q, const float* k, const float* v, const float* o_rows, const float* do_rows,
             const float* lse, float* dsum, float* dqkv_rows, int BH, hipStream_t st) {
    if (int rc = attn_train_init()) return rc;
    const int M = (BH / NH) * NTOK;
    attn_dsum_kernel<<<(M * NH + 255) / 256, 256, 0, st>>>(o_rows, do_rows, dsum, M);
    T2S_LAUNCH_CHECK();
    attn_bwd_dq_kernel<<<BH, 512, LDS2, st>>>(q, k, v, do_rows, lse, dsum, dqkv_rows);
    T2S_LAUNCH_CHECK();
    attn_bwd_dkv_kernel<<<BH, 512, LDS2, st>>>(q, k, v, do_rows, lse, dsum, dqkv_rows);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

}  // namespace t2s
