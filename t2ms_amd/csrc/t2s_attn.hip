// Fused attention for the DiT: softmax(q k^T / sqrt(32)) v, N = 480 keys, head_dim 32,
// exact fp32 on v_mfma_f32_32x32x2_f32.  (timm 1.0.11 Attention.forward core; reference
// call site model/denoiser/transformer.py:104,116.)
//
// Transposed formulation.  Per 32-key block a wave computes
//   S^T[key][query] = K Q^T          (A operand = K fragment, B operand = Q^T in registers)
// so a query's scores sit in one lane pair (l, l^32): the online-softmax max / sum are register
// reductions plus one cross-half exchange, and -- an f32 MFMA operand being one register per
// lane -- the exponentiated tile P^T is ALREADY the B operand of
//   O^T[d][query] += V^T P^T         (A operand = V^T fragment)
// with MFMA step r contracting the key pair {klo(r), klo(r)+4} held by the two lane halves.
// No LDS round trip for P, no transposes.
//
// attn_fwd_packed_kernel (the DiT's kernel): q, k are fragment-major per head, v is stored
// TRANSPOSED fragment-major (written that way by t2s_rows.h), so a 32-key block is 4 K + 4 V^T
// fragments of 1 KiB.  A workgroup (4 waves, 2 query tiles each = 256 queries) streams the 15
// key blocks through a 4-slot LDS ring filled by LDS-DMA two blocks ahead (counted vmcnt + raw
// s_barrier, so the DMA stays in flight across barriers); every LDS read is a lane-linear,
// conflict-free ds_read_b128 and feeds 2 MFMAs (both query tiles).  Two workgroups cover the
// 15 query tiles of a (sequence, head); LDS is 32 KiB, so several workgroups share a CU and
// one's softmax (VALU) overlaps another's MFMAs.
//
// attn_fwd_plain_kernel: same math on plain (BH,480,32) tensors for the standalone C-ABI entry.
#include "t2s_common.h"

namespace t2s {

// ------------------------------------------------------------------ shared math
struct SoftmaxState {
    float m_run;   // running max (log2 domain)
    float l_lane;  // this lane's partial row sum
};

// online-softmax update of one 32x32 transposed score tile held in st (in: scores, out: P^T)
__device__ __forceinline__ void softmax_block(f32x16& st, f32x16& ot, SoftmaxState& s) {
    float mloc = st[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, st[r]);
    mloc = fmaxf(mloc, xhalf(mloc));
    const float m_new = fmaxf(s.m_run, mloc);
    const float alpha = __builtin_amdgcn_exp2f(s.m_run - m_new);  // first block: exp2(-inf) = 0
    s.m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        st[r] = __builtin_amdgcn_exp2f(st[r] - m_new);
        psum += st[r];
    }
    s.l_lane = s.l_lane * alpha + psum;
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[r] *= alpha;
}

constexpr float QSCALE = 0.17677669529663687f * 1.4426950408889634f;  // 32^-0.5 * log2(e)

// ------------------------------------------------------------------ packed (DiT) kernel
constexpr int ATT_SLOTS = 4;
constexpr int ATT_SLOT_F4 = 512;                        // 4 K + 4 V^T fragments of 64 float4
constexpr int ATT_LDS_BYTES = ATT_SLOTS * ATT_SLOT_F4 * 16;  // 32 KiB
constexpr int NKB = NTOK / 32;                          // 15 key blocks / query tiles

__global__ __launch_bounds__(256, 2) void attn_fwd_packed_kernel(const float* __restrict__ q,
                                                                 const float* __restrict__ k,
                                                                 const float* __restrict__ vT,
                                                                 float* __restrict__ o) {
    extern __shared__ __attribute__((aligned(16))) f32x4 ring[];
    const int bh = blockIdx.x >> 1;
    const int part = blockIdx.x & 1;          // query tiles [8*part, 8*part+8)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const f32x4* qg = reinterpret_cast<const f32x4*>(q) + (size_t)bh * NKB * 256;
    const f32x4* kg = reinterpret_cast<const f32x4*>(k) + (size_t)bh * NKB * 256;
    const f32x4* vg = reinterpret_cast<const f32x4*>(vT) + (size_t)bh * NKB * 256;

    // this wave DMAs K fragments {wave} and V^T fragment {wave} of every block (8 pieces / 4 waves)
    auto issue = [&](int jb) {
        f32x4* slot = ring + (jb & (ATT_SLOTS - 1)) * ATT_SLOT_F4;
        glds16(kg + (jb * 4 + wave) * 64 + lane, slot + wave * 64);
        glds16(vg + (jb * 4 + wave) * 64 + lane, slot + 256 + wave * 64);
    };
    issue(0);
    issue(1);

    const int t0 = part * 8 + wave * 2;       // query tiles t0, t0+1 (tile 15 does not exist)
    const int t1 = (t0 + 1 < NKB) ? t0 + 1 : NKB - 1;
    const bool t1_valid = t0 + 1 < NKB;
    f32x4 qa[4], qb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        qa[g] = qg[(t0 * 4 + g) * 64 + lane] * QSCALE;
        qb[g] = qg[(t1 * 4 + g) * 64 + lane] * QSCALE;
    }
    f32x16 oa, ob;
#pragma unroll
    for (int r = 0; r < 16; ++r) oa[r] = ob[r] = 0.f;
    SoftmaxState sa{-INFINITY, 0.f}, sb{-INFINITY, 0.f};

#pragma unroll 1
    for (int jb = 0; jb < NKB; ++jb) {
        // block jb landed in every wave's view: own DMA retired (all but the youngest pair),
        // then the workgroup barrier
        if (jb < NKB - 1)
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (jb + 2 < NKB) issue(jb + 2);   // slot (jb+2)&3 was last read in iteration jb-2
        const f32x4* slot = ring + (jb & (ATT_SLOTS - 1)) * ATT_SLOT_F4 + lane;

        f32x16 sta, stb;
#pragma unroll
        for (int r = 0; r < 16; ++r) sta[r] = stb[r] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 kf = slot[g * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sta = mfma32(kf[e], qa[g][e], sta);
                stb = mfma32(kf[e], qb[g][e], stb);
            }
        }
        softmax_block(sta, oa, sa);
        softmax_block(stb, ob, sb);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 vf = slot[256 + g * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                oa = mfma32(vf[e], sta[4 * g + e], oa);
                ob = mfma32(vf[e], stb[4 * g + e], ob);
            }
        }
    }

    // ---- normalise and store O[query][d]: lane (query i, half) holds d = 8g + 4*half + e ----
    const int seq = bh / NH, head = bh % NH;
    f32x4* og = reinterpret_cast<f32x4*>(o);
    {
        const float inv = 1.0f / (sa.l_lane + xhalf(sa.l_lane));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 w = {oa[4 * g] * inv, oa[4 * g + 1] * inv, oa[4 * g + 2] * inv, oa[4 * g + 3] * inv};
            og[(((size_t)seq * NKB + t0) * 16 + head * 4 + g) * 64 + lane] = w;
        }
    }
    if (t1_valid) {
        const float inv = 1.0f / (sb.l_lane + xhalf(sb.l_lane));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 w = {ob[4 * g] * inv, ob[4 * g + 1] * inv, ob[4 * g + 2] * inv, ob[4 * g + 3] * inv};
            og[(((size_t)seq * NKB + t1) * 16 + head * 4 + g) * 64 + lane] = w;
        }
    }
}

int launch_attn_packed(const float* q, const float* k, const float* vT, float* o, int BH, hipStream_t st) {
    attn_fwd_packed_kernel<<<BH * 2, 256, ATT_LDS_BYTES, st>>>(q, k, vT, o);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

// ------------------------------------------------------------------ plain-layout kernel
// One workgroup (8 waves) per (sequence, head); K (row stride 36 floats: conflict-free
// ds_read_b128) and V of that head live in LDS (130.5 KB); each wave owns query tiles {w, w+8}.
constexpr int KSTR = 36;
constexpr int PLAIN_LDS_BYTES = (NTOK * KSTR + NTOK * DH) * 4;  // 130,560

__global__ __launch_bounds__(512) void attn_fwd_plain_kernel(const float* __restrict__ q,
                                                             const float* __restrict__ k,
                                                             const float* __restrict__ v,
                                                             float* __restrict__ o) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + NTOK * KSTR;
    const int bh = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, i = lane & 31;
    const float* kg = k + (size_t)bh * NTOK * DH;
    const float* vg = v + (size_t)bh * NTOK * DH;
    const float* qg = q + (size_t)bh * NTOK * DH;
    for (int idx = tid; idx < NTOK * DH / 4; idx += 512) {
        const f32x4 kv = reinterpret_cast<const f32x4*>(kg)[idx];
        const f32x4 vv = reinterpret_cast<const f32x4*>(vg)[idx];
        const int row = idx >> 3, col = (idx & 7) * 4;
        *reinterpret_cast<f32x4*>(Ks + row * KSTR + col) = kv;
        *reinterpret_cast<f32x4*>(Vs + row * DH + col) = vv;
    }
    __syncthreads();
    for (int qt = wave; qt < NKB; qt += 8) {
        f32x4 qf[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            qf[g] = *reinterpret_cast<const f32x4*>(qg + (size_t)(qt * 32 + i) * DH + 8 * g + 4 * half) * QSCALE;
        f32x16 ot;
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[r] = 0.f;
        SoftmaxState s{-INFINITY, 0.f};
        for (int jb = 0; jb < NKB; ++jb) {
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
            softmax_block(st, ot, s);
            const float* vrow = Vs + (jb * 32 + 4 * half) * DH + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int klo = (r & 3) + 8 * (r >> 2);
                ot = mfma32(vrow[klo * DH], st[r], ot);
            }
        }
        const float inv = 1.0f / (s.l_lane + xhalf(s.l_lane));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 w = {ot[4 * g] * inv, ot[4 * g + 1] * inv, ot[4 * g + 2] * inv, ot[4 * g + 3] * inv};
            *reinterpret_cast<f32x4*>(o + ((size_t)bh * NTOK + qt * 32 + i) * DH + 8 * g + 4 * half) = w;
        }
    }
}

int attn_init() {  // once, outside any stream capture
    static bool attr_set = false;
    if (!attr_set) {
        T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_plain_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, PLAIN_LDS_BYTES));
        attr_set = true;
    }
    return T2S_OK;
}

}  // namespace t2s

extern "C" int t2s_attn_fwd(const float* q, const float* k, const float* v, float* o, int BH,
                            void* stream) {
    T2S_REQUIRE(q && k && v && o, "t2s_attn_fwd: NULL pointer");
    T2S_REQUIRE(BH > 0, "t2s_attn_fwd: BH=%d must be > 0", BH);
    if (int rc = t2s::attn_init()) return rc;
    t2s::attn_fwd_plain_kernel<<<BH, 512, t2s::PLAIN_LDS_BYTES, (hipStream_t)stream>>>(q, k, v, o);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

extern "C" int t2s_attn_fwd_packed(const float* q, const float* k, const float* vT, float* o, int n_seq,
                                   void* stream) {
    T2S_REQUIRE(q && k && vT && o, "t2s_attn_fwd_packed: NULL pointer");
    T2S_REQUIRE(n_seq > 0, "t2s_attn_fwd_packed: n_seq=%d must be > 0", n_seq);
    return t2s::launch_attn_packed(q, k, vT, o, n_seq * t2s::NH, (hipStream_t)stream);
}
