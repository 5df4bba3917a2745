// Fused attention of the DiT forward in "bf16x3" arithmetic (t2s_x3.h): fp32-accurate QK^T and PV
// on v_mfma_f32_32x32x16_bf16.  Same algorithm and skeleton as attn_fwd_persistent_kernel
// (t2s_attn.hip; timm 1.0.11 Attention core, reference call site model/denoiser/transformer.py:116):
// transposed tiles (a query's scores live in one lane pair), sticky-reference softmax with the
// reference folded into the MFMA C operand, ONE persistent 8-wave workgroup per CU walking the heads
// with the K / V^T operands streamed through a 4-slot LDS ring by LDS-DMA three blocks ahead.
//
// Operands: q fp32 fragment-major (as the fp32 kernel; scaled and split in registers once per head);
// k, v^T pre-split bf16 planes written by the row-chain kernel (t2s_x3.h: X3_TILE_UNITS); the
// exponentiated tile P^T is split in registers.  Per 32-key block and query tile: 12 + 12 MFMAs of
// 32 cycles (768) instead of 16 + 16 of 64 (2048).
#include <stdlib.h>
#include "t2s_x3.h"

namespace t2s {

namespace {
constexpr int NKB = NTOK / 32;                                   // 15 key blocks / query tiles
constexpr float QSCALE = 0.17677669529663687f * 1.4426950408889634f;   // 32^-0.5 * log2(e)
constexpr float SM_BIG = 1.152921504606847e18f;                  // 2^60
constexpr int X3_SLOTS = 4;
constexpr int X3_SLOT_UNITS = 2 * X3_TILE_UNITS;                 // K planes + V^T planes = 12 KiB
constexpr int X3_LDS_BYTES = X3_SLOTS * X3_SLOT_UNITS * 16;      // 48 KiB
constexpr int X3_THREADS = 512;

__device__ __forceinline__ float pair_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float pair_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}


struct TileState {
    f32x16 negm;   // -m_ref in all 16 registers (C operand of the first QK MFMA)
    float m_ref;
    float l_lane;
};

__device__ __forceinline__ float exp_sum(f32x16& st) {
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = __builtin_amdgcn_exp2f(st[r]);
    return ((st[0] + st[1]) + (st[2] + st[3])) + ((st[4] + st[5]) + (st[6] + st[7])) +
           (((st[8] + st[9]) + (st[10] + st[11])) + ((st[12] + st[13]) + (st[14] + st[15])));
}

__device__ __forceinline__ void glds16u(const bf16x8* gsrc_lane, bf16x8* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// S^T tile (+ c) = K Q^T over the head dimension (two 16-deep steps), fp32-accurate
__device__ __forceinline__ f32x16 scores(const Split3 (&kf)[2], const Split3 (&q)[2], f32x16 c) {
    c = mfma_x3(kf[0], q[0], c);
    return mfma_x3(kf[1], q[1], c);
}

// the same with a C operand that must survive (the sticky reference): the first MFMA writes registers of its own
// instead of hipcc's copy + tied accumulate (see mfma32_from, t2s_common.h); same order of additions as scores()
__device__ __forceinline__ f32x16 scores_from(const Split3 (&kf)[2], const Split3 (&q)[2], const f32x16& c) {
    f32x16 acc;
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(kf[0].m), "v"(q[0].m), "v"(c));
    acc = mfma16(kf[0].l, q[0].h, acc);
    acc = mfma16(kf[0].h, q[0].l, acc);
    acc = mfma16(kf[0].m, q[0].h, acc);
    acc = mfma16(kf[0].h, q[0].m, acc);
    acc = mfma16(kf[0].h, q[0].h, acc);
    return mfma_x3(kf[1], q[1], acc);
}

// rare path: raw scores (C = 0) -> new reference, rescale the running sum / output, P^T in st
__device__ __forceinline__ float rereference(const Split3 (&kf)[2], const Split3 (&q)[2], f32x16& st, f32x16& ot,
                                             TileState& t) {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    st = scores(kf, q, z);
    float mloc = st[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, st[r]);
    mloc = pair_max(mloc);
    const float m_new = fmaxf(t.m_ref, mloc);
    const float alpha = __builtin_amdgcn_exp2f(t.m_ref - m_new);   // 0 on the first block
    t.m_ref = m_new;
    t.l_lane *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        ot[r] *= alpha;
        t.negm[r] = -m_new;
        st[r] -= m_new;
    }
    return exp_sum(st);
}

// scaled + split Q^T operand of one query tile from its four fp32 fragments (k-step s = fragments 2s, 2s+1)
__device__ __forceinline__ void load_q(const f32x4 (&raw)[4], Split3 (&q)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const f32x4 a = raw[2 * s] * QSCALE, b = raw[2 * s + 1] * QSCALE;
        const f32x8 v = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        q[s] = split3(v);
    }
}

// NT = query tiles of this wave (2; 1 for the wave that holds tile 14 alone).
// STAG: the two waves of a SIMD (waves w and w+4) run the same loop with the per-block barrier at
// DIFFERENT points -- after the softmax (STAG = 0) or right after QK^T (STAG = 1) -- so that they
// meet half a period apart: one is in its MFMA phases (PV, QK^T) while the other exponentiates and
// splits P on the VALU.  bf16 MFMAs occupy the vector issue port for 8 of their 32 cycles, so the
// two phases of different waves overlap; with the barrier at the same point both waves would run
// their MFMA phases together and their VALU phases together, leaving the matrix pipe idle half the time.
//
// Ring protocol (4 slots, block gb in slot gb & 3).  Barrier j (the one of key block j) guarantees
//   * block j+1 has landed (every DMA-issuing wave waited until only its youngest block is in flight),
//   * every wave is done with block j-1 (both variants finish PV(j-1) before QK(j)),
// so after it a wave may read K(j+1) and refill slot (j+3) & 3 == (j-1) & 3 with block j+3.
template <int NT, int STAG>
__device__ __forceinline__ void attn_x3_body(bf16x8* ring, const f32x4* qall, const bf16x8* kall, const bf16x8* vall,
                                             f32x4* og, int BH, int lane, int wave) {
    const int stride = gridDim.x;
    const int t0 = wave * 2;
    // 12 pieces of 1 KiB per key block (6 K planes/steps, 6 V^T)
    auto issue_piece = [&](int bh, int jb, int gslot, int p) {
        const bf16x8* src = (p < 6 ? kall : vall) + ((size_t)(bh * NKB + jb) * 6 + (p < 6 ? p : p - 6)) * 64 + lane;
        glds16u(src, ring + (gslot & (X3_SLOTS - 1)) * X3_SLOT_UNITS + p * 64);
    };
    // The one-tile wave (NT == 1: wave 7, whose SIMD carries 3 tiles where the others carry 4) issues all 12 pieces of a block,
    // the two-tile waves none (round 5, as t2s_attn.hip: attention 350 against 360 us per launch, sampler +0.9 %,
    // profiles/r05_x3_dma_w7_ab.txt).  `jb` may run past 14 into the following heads of this workgroup.
    auto issue_block = [&](int bh, int jb, int gslot) {
        if constexpr (NT == 1) {
            if (jb >= NKB) { jb -= NKB; bh += stride; }
            if (bh >= BH) { bh -= stride; jb = NKB - 1; }   // past the end: harmless re-fetch
#pragma unroll
            for (int p = 0; p < 12; ++p) issue_piece(bh, jb, gslot, p);
        }
    };
    // counted waits of the issuing wave: everything but the 12-piece DMAs of the youngest one / two blocks has landed
    auto wait_but = [&](int blocks) {
        if constexpr (NT == 1) {
            if (blocks == 2) asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        }
    };
    auto load_k = [&](Split3 (&kf)[2], int gslot) {
        const bf16x8* slot = ring + (gslot & (X3_SLOTS - 1)) * X3_SLOT_UNITS + lane;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            kf[s].h = slot[(0 + s) * 64];
            kf[s].m = slot[(2 + s) * 64];
            kf[s].l = slot[(4 + s) * 64];
        }
    };
    int bh = blockIdx.x;
    int gb = 0;                                      // global block counter -> ring slot
    issue_block(bh, 0, 0);
    issue_block(bh, 1, 1);
    issue_block(bh, 2, 2);

    Split3 qa[2], qb[2];
    f32x4 qna[4], qnb[4];
    {
        const f32x4* qg = qall + (size_t)bh * NKB * 256;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            qna[g] = qg[(t0 * 4 + g) * 64 + lane];
            qnb[g] = (NT == 2) ? qg[((t0 + 1) * 4 + g) * 64 + lane] : qna[g];
        }
        load_q(qna, qa);
        load_q(qnb, qb);
    }
    wait_but(2);                                     // block 0 landed
    __builtin_amdgcn_s_barrier();
    Split3 kf[2];
    load_k(kf, 0);

#pragma unroll 1
    for (; bh < BH; bh += stride) {
        f32x16 oa, ob;
        TileState ta, tb;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            oa[r] = ob[r] = 0.f;
            ta.negm[r] = tb.negm[r] = INFINITY;
        }
        ta.m_ref = tb.m_ref = -INFINITY;
        ta.l_lane = tb.l_lane = 0.f;

#pragma unroll 1
        for (int jb = 0; jb < NKB; ++jb, ++gb) {
            // ---- QK^T (MFMA)
            // block 0 of a head has no reference yet: straight to the reference-setting path (one QK pass, not two)
            const bool first = jb == 0;                           // scalar
            f32x16 sta = ta.negm, stb = tb.negm;
            if (!first) {
                sta = scores_from(kf, qa, ta.negm);
                if (NT == 2) stb = scores_from(kf, qb, tb.negm);
            }
            if (STAG) {
                wait_but(1);
                __builtin_amdgcn_s_barrier();
                issue_block(bh, jb + 3, gb + 3);
            }
            const bf16x8* slot = ring + (gb & (X3_SLOTS - 1)) * X3_SLOT_UNITS + lane;
            Split3 vf[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                vf[s].h = slot[(6 + 0 + s) * 64];
                vf[s].m = slot[(6 + 2 + s) * 64];
                vf[s].l = slot[(6 + 4 + s) * 64];
            }
            // ---- softmax + split of P^T (VALU)
            float psa = 0.f, psb = 0.f;
            bool redo = first;
            if (!first) {
                psa = exp_sum(sta);
                if (NT == 2) psb = exp_sum(stb);
                const bool stale = !(psa < SM_BIG) || !(psb < SM_BIG);
                redo = __builtin_amdgcn_ballot_w64(stale) != 0;   // wave-uniform, rare
            }
            if (redo) {
                psa = rereference(kf, qa, sta, oa, ta);
                if (NT == 2) psb = rereference(kf, qb, stb, ob, tb);
            }
            ta.l_lane += psa;
            tb.l_lane += psb;
            const Split3 pa0 = split3_acc(sta, 0), pa1 = split3_acc(sta, 1);
            Split3 pb0 = pa0, pb1 = pa1;
            if (NT == 2) { pb0 = split3_acc(stb, 0); pb1 = split3_acc(stb, 1); }
            if (!STAG) {
                wait_but(1);
                __builtin_amdgcn_s_barrier();
                issue_block(bh, jb + 3, gb + 3);
            }
            // K fragments of the next block (landed: barrier above), read behind the PV MFMAs
            load_k(kf, gb + 1);
            // ---- PV (MFMA): O^T += V^T P^T
            oa = mfma_x3(vf[0], pa0, oa);
            oa = mfma_x3(vf[1], pa1, oa);
            if (NT == 2) {
                ob = mfma_x3(vf[0], pb0, ob);
                ob = mfma_x3(vf[1], pb1, ob);
            }
            // prefetch the next head's Q fragments (a whole block old by the next counted wait)
            if (jb == NKB - 4 && bh + stride < BH) {
                const f32x4* qg = qall + (size_t)(bh + stride) * NKB * 256;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    qna[g] = qg[(t0 * 4 + g) * 64 + lane];
                    if (NT == 2) qnb[g] = qg[((t0 + 1) * 4 + g) * 64 + lane];
                }
            }
        }
        // ---- normalise and store O of this head; swap in the prefetched Q ----
        const int seq = bh / NH, head = bh % NH;
        {
            const float inv = 1.0f / pair_sum(ta.l_lane);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 w = {oa[4 * g] * inv, oa[4 * g + 1] * inv, oa[4 * g + 2] * inv, oa[4 * g + 3] * inv};
                og[(((size_t)seq * NKB + t0) * 16 + head * 4 + g) * 64 + lane] = w;
            }
        }
        if (NT == 2) {
            const float inv = 1.0f / pair_sum(tb.l_lane);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 w = {ob[4 * g] * inv, ob[4 * g + 1] * inv, ob[4 * g + 2] * inv, ob[4 * g + 3] * inv};
                og[(((size_t)seq * NKB + t0 + 1) * 16 + head * 4 + g) * 64 + lane] = w;
            }
        }
        load_q(qna, qa);
        if (NT == 2) load_q(qnb, qb);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the trailing (unused) DMAs
}

// k3 / vT3: split planes (BH*15 tiles x 6 KiB each); q, o: fp32 fragment-major as in t2s_attn_fwd_packed
__global__ __launch_bounds__(X3_THREADS, 2) T2S_X3_KERNEL void attn_fwd_x3_kernel(const float* __restrict__ q,
                                                                    const __bf16* __restrict__ k3,
                                                                    const __bf16* __restrict__ vT3,
                                                                    float* __restrict__ o, int BH) {
    extern __shared__ __attribute__((aligned(16))) bf16x8 ring3[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const f32x4* qg = reinterpret_cast<const f32x4*>(q);
    const bf16x8* kg = reinterpret_cast<const bf16x8*>(k3);
    const bf16x8* vg = reinterpret_cast<const bf16x8*>(vT3);
    f32x4* og = reinterpret_cast<f32x4*>(o);
    // (measured: without the stagger 324 us, with 313 us; s_setprio 1 for waves 4-7 on top: 320 us)
    constexpr int SG = 1;
    if (wave < 4)
        attn_x3_body<2, 0>(ring3, qg, kg, vg, og, BH, lane, wave);
    else if (wave < 7)
        attn_x3_body<2, SG>(ring3, qg, kg, vg, og, BH, lane, wave);
    else
        attn_x3_body<1, SG>(ring3, qg, kg, vg, og, BH, lane, wave);   // tile 14 only (15 is void)
}

// plain (BH,480,32) fp32 k or v -> split planes: k as the A operand with the key on the lane
// (transpose = 0), v as V^T with the feature on the lane (transpose = 1)
__global__ void pack_x3_kernel(const float* __restrict__ src, bf16x8* __restrict__ dst, int BH, int transpose) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // one (bh, tile, s, lane)
    if (idx >= BH * NKB * 2 * 64) return;
    const int lane = idx & 63, s = (idx >> 6) & 1, tile = (idx >> 7) % NKB, bh = idx / (NKB * 128);
    const int i = lane & 31, h = lane >> 5;
    f32x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int kk = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);     // permuted k order of the fragment
        v[j] = transpose ? src[((size_t)bh * NTOK + tile * 32 + kk) * DH + i]     // V^T[d = i][key = kk]
                         : src[((size_t)bh * NTOK + tile * 32 + i) * DH + kk];    // K[key = i][d = kk]
    }
    const Split3 sp = split3(v);
    bf16x8* d = dst + ((size_t)(bh * NKB + tile) * 6 + s) * 64 + lane;
    d[0] = sp.h;
    d[2 * 64] = sp.m;
    d[4 * 64] = sp.l;
}
}  // namespace

int attn_x3_init() {   // once, outside any stream capture
    static bool done = false;
    if (!done) {
        T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_x3_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS_BYTES));
        done = true;
    }
    return T2S_OK;
}

int launch_attn_x3(const float* q, const __bf16* k3, const __bf16* vT3, float* o, int BH, hipStream_t st) {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n_cu = prop.multiProcessorCount;
        else
            n_cu = 256;
    }
    static int wg_per_cu = 0;
    if (wg_per_cu == 0) {
        const char* e = getenv("T2S_X3_WGS_PER_CU");
        wg_per_cu = e ? atoi(e) : 1;
        if (wg_per_cu < 1 || wg_per_cu > 2) wg_per_cu = 1;
    }
    const int slots = n_cu * wg_per_cu;
    const int grid = BH < slots ? BH : slots;
    attn_fwd_x3_kernel<<<grid, X3_THREADS, X3_LDS_BYTES, st>>>(q, k3, vT3, o, BH);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

int pack_x3(const float* src, __bf16* dst, int BH, int transpose, hipStream_t st) {
    const int n = BH * NKB * 2 * 64;
    pack_x3_kernel<<<(n + 255) / 256, 256, 0, st>>>(src, reinterpret_cast<bf16x8*>(dst), BH, transpose);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

}  // namespace t2s

// ------------------------------------------------------------------ C ABI: stand-alone entry on plain tensors
namespace {
using namespace t2s;
// q (BH,480,32) -> fp32 fragment-major [(bh*15 + tile)*4 + g][lane][e] = Q[32 tile + i][8g + 4h + e]
__global__ void q_to_frag_kernel(const float* __restrict__ q, f32x4* __restrict__ qf, int BH) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= BH * 15 * 4 * 64) return;
    const int lane = idx & 63, g = (idx >> 6) & 3, tile = (idx >> 8) % 15, bh = idx / (15 * 256);
    qf[idx] = *reinterpret_cast<const f32x4*>(q + ((size_t)bh * NTOK + tile * 32 + (lane & 31)) * DH + 8 * g + 4 * (lane >> 5));
}
// o fragment-major [((seq*15 + tile)*16 + head*4 + g)][lane][e] -> (BH,480,32)
__global__ void o_from_frag_kernel(const f32x4* __restrict__ of, float* __restrict__ o, int BH) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= BH * 15 * 4 * 64) return;
    const int lane = idx & 63, g = (idx >> 6) & 3, tile = (idx >> 8) % 15, bh = idx / (15 * 256);
    const int seq = bh / NH, head = bh % NH;
    *reinterpret_cast<f32x4*>(o + ((size_t)bh * NTOK + tile * 32 + (lane & 31)) * DH + 8 * g + 4 * (lane >> 5)) =
        of[(((size_t)seq * 15 + tile) * 16 + head * 4 + g) * 64 + lane];
}
}  // namespace

extern "C" int t2s_attn_fwd_x3(const float* q, const float* k, const float* v, float* o, int BH, void* stream) {
    T2S_REQUIRE(q && k && v && o && BH > 0 && BH % NH == 0, "t2s_attn_fwd_x3: bad argument (BH=%d must be a multiple of 4)", BH);
    hipStream_t st = (hipStream_t)stream;
    if (int rc = attn_x3_init()) return rc;
    const size_t n = (size_t)BH * NTOK * DH;
    float* f32buf = nullptr;
    __bf16* planes = nullptr;
    T2S_HIP_CHECK(hipMalloc(&f32buf, 2 * n * sizeof(float)));
    if (hipMalloc(&planes, 2 * 3 * n * sizeof(__bf16)) != hipSuccess) {
        (void)hipFree(f32buf);
        set_error("t2s_attn_fwd_x3: hipMalloc failed");
        return T2S_E_HIP;
    }
    const int nf = BH * 15 * 4 * 64;
    q_to_frag_kernel<<<(nf + 255) / 256, 256, 0, st>>>(q, reinterpret_cast<f32x4*>(f32buf), BH);
    int rc = pack_x3(k, planes, BH, 0, st);
    if (rc == T2S_OK) rc = pack_x3(v, planes + 3 * n, BH, 1, st);
    if (rc == T2S_OK) rc = launch_attn_x3(f32buf, planes, planes + 3 * n, f32buf + n, BH, st);
    if (rc == T2S_OK) {
        o_from_frag_kernel<<<(nf + 255) / 256, 256, 0, st>>>(reinterpret_cast<const f32x4*>(f32buf + n), o, BH);
        if (hipGetLastError() != hipSuccess) rc = T2S_E_HIP;
    }
    (void)hipStreamSynchronize(st);
    (void)hipFree(f32buf);
    (void)hipFree(planes);
    return rc;
}

