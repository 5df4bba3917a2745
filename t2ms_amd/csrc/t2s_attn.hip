// Fused attention for the DiT: softmax(q k^T / sqrt(32)) v, N = 480 keys, head_dim 32,
// exact fp32 on v_mfma_f32_32x32x2_f32.  (timm 1.0.11 Attention.forward core; reference
// call site model/denoiser/transformer.py:104,116.)
//
// One workgroup (8 waves) per (sequence, head).  K (480x32, row stride 36 floats) and
// V (480x32) of that head live in LDS for the whole workgroup (130.5 KB); each wave owns
// 32-query tiles {w, w+8}.  Per 32-key block the wave computes the TRANSPOSED score tile
//   S^T[key][query] = K Q^T          (A operand = K rows from LDS, B operand = Q^T in regs)
// so that a query's scores sit in one lane pair (l, l^32): the online-softmax row max / sum
// are register reductions plus one cross-half exchange, and -- because an f32 MFMA operand
// is one register per lane -- the exponentiated tile P^T is ALREADY the B operand of
//   O^T[d][query] += V^T P^T         (A operand = V rows from LDS)
// with MFMA step r contracting the key pair {klo(r), klo(r)+4} that register r holds in the
// two lane halves.  No LDS round trip for P, no transposes.
//
// FRAG = true: q/k/v/o are in the library's fragment-major layout (t2s_common.h: frag_index;
// q/k/v per head with C = 32, o per sequence with C = 128) so Q loads, K/V staging reads and
// O stores are contiguous 1 KiB per wave instruction.  FRAG = false: plain (BH,480,32).
#include "t2s_common.h"

namespace t2s {

constexpr int KSTR = 36;  // padded K row stride (floats): conflict-free ds_read_b128 across 16 rows
constexpr int ATTN_LDS_BYTES = (NTOK * KSTR + NTOK * DH) * 4;  // 130,560

template <bool FRAG>
__global__ __launch_bounds__(512) void attn_fwd_kernel(const float* __restrict__ q,
                                                       const float* __restrict__ k,
                                                       const float* __restrict__ v,
                                                       float* __restrict__ o) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + NTOK * KSTR;

    const int bh = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int i = lane & 31;

    const float* kg = k + (size_t)bh * NTOK * DH;
    const float* vg = v + (size_t)bh * NTOK * DH;
    const float* qg = q + (size_t)bh * NTOK * DH;

    // ---- stage K (padded rows) and V into LDS: 3840 float4 each, 512 threads ----
    for (int idx = tid; idx < NTOK * DH / 4; idx += 512) {
        const f32x4 kv = reinterpret_cast<const f32x4*>(kg)[idx];
        const f32x4 vv = reinterpret_cast<const f32x4*>(vg)[idx];
        int row, col;
        if constexpr (FRAG) {  // idx = (tile*4 + g)*64 + l
            const int l = idx & 63, g = (idx >> 6) & 3, tile = idx >> 8;
            row = tile * 32 + (l & 31);
            col = 8 * g + 4 * (l >> 5);
        } else {
            row = idx >> 3;
            col = (idx & 7) * 4;
        }
        *reinterpret_cast<f32x4*>(Ks + row * KSTR + col) = kv;
        *reinterpret_cast<f32x4*>(Vs + row * DH + col) = vv;
    }
    __syncthreads();

    // softmax in the log2 domain: p = 2^(s*log2e*scale - m)
    const float qscale = 0.17677669529663687f * 1.4426950408889634f;

    for (int qt = wave; qt < NTOK / 32; qt += 8) {
        // Q^T fragment: lane (i,half) holds Q[qt*32+i][8g + 4*half + e]
        f32x4 qf[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if constexpr (FRAG)
                qf[g] = reinterpret_cast<const f32x4*>(qg)[(qt * 4 + g) * 64 + lane];
            else
                qf[g] = *reinterpret_cast<const f32x4*>(qg + (size_t)(qt * 32 + i) * DH + 8 * g + 4 * half);
            qf[g] *= qscale;
        }
        f32x16 ot;
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[r] = 0.f;
        float m_run = -INFINITY;
        float l_lane = 0.f;

        for (int jb = 0; jb < NTOK / 32; ++jb) {
            // ---- S^T = K Q^T for 32 keys x 32 queries ----
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
            const float* krow = Ks + (jb * 32 + i) * KSTR + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(krow + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) st = mfma32(kf[e], qf[g][e], st);
            }
            // ---- online softmax over this lane pair's 32 keys ----
            float mloc = st[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, st[r]);
            mloc = fmaxf(mloc, xhalf(mloc));
            const float m_new = fmaxf(m_run, mloc);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // first block: exp2(-inf) = 0
            m_run = m_new;
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[r] = __builtin_amdgcn_exp2f(st[r] - m_new);
                psum += st[r];
            }
            l_lane = l_lane * alpha + psum;
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[r] *= alpha;
            // ---- O^T += V^T P^T : step r contracts keys {klo(r), klo(r)+4} ----
            const float* vrow = Vs + (jb * 32 + 4 * half) * DH + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int klo = (r & 3) + 8 * (r >> 2);
                ot = mfma32(vrow[klo * DH], st[r], ot);
            }
        }
        // ---- normalise and store O[query i][d]: lane holds d = 8g + 4*half + (0..3) ----
        const float l_tot = l_lane + xhalf(l_lane);
        const float inv = 1.0f / l_tot;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 w = {ot[4 * g + 0] * inv, ot[4 * g + 1] * inv, ot[4 * g + 2] * inv,
                             ot[4 * g + 3] * inv};
            if constexpr (FRAG) {
                // o is (S*480,128) fragment-major: tile = seq*15 + qt, G = head*4 + g
                const int seq = bh / NH, head = bh % NH;
                reinterpret_cast<f32x4*>(o)[(((size_t)seq * (NTOK / 32) + qt) * 16 + head * 4 + g) * 64 + lane] = w;
            } else {
                *reinterpret_cast<f32x4*>(o + ((size_t)bh * NTOK + qt * 32 + i) * DH + 8 * g + 4 * half) = w;
            }
        }
    }
}

int attn_init() {  // once, outside any stream capture
    static bool attr_set = false;
    if (!attr_set) {
        T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, ATTN_LDS_BYTES));
        T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, ATTN_LDS_BYTES));
        attr_set = true;
    }
    return T2S_OK;
}

// DiT-internal launch: fragment-major q/k/v (per head) and o (per sequence)
int launch_attn_frag(const float* q, const float* k, const float* v, float* o, int BH, hipStream_t st) {
    attn_fwd_kernel<true><<<BH, 512, ATTN_LDS_BYTES, st>>>(q, k, v, o);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

}  // namespace t2s

extern "C" int t2s_attn_fwd(const float* q, const float* k, const float* v, float* o, int BH,
                            void* stream) {
    T2S_REQUIRE(q && k && v && o, "t2s_attn_fwd: NULL pointer");
    T2S_REQUIRE(BH > 0, "t2s_attn_fwd: BH=%d must be > 0", BH);
    if (int rc = t2s::attn_init()) return rc;
    t2s::attn_fwd_kernel<false><<<BH, 512, t2s::ATTN_LDS_BYTES, (hipStream_t)stream>>>(q, k, v, o);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}
