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
#include <stdlib.h>

#include "t2s_common.h"

namespace t2s {

// ------------------------------------------------------------------ shared math
struct SoftmaxState {
    float m_run;   // running max (log2 domain)
    float l_lane;  // this lane's partial row sum
};

// max / sum over a lane pair (l, l^32) with v_permlane32_swap (VALU; no LDS crossbar round trip):
// swap(v, v) returns {lo-half of v | lo-half of v} and {hi-half | hi-half}, so op(r0, r1) is the
// pair-wise result in both halves.
__device__ __forceinline__ float pair_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float pair_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// online-softmax update of one 32x32 transposed score tile held in st (in: scores, out: P^T)
__device__ __forceinline__ void softmax_block(f32x16& st, f32x16& ot, SoftmaxState& s) {
    float mloc = st[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, st[r]);
    mloc = pair_max(mloc);
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

// ---- softmax with a sticky reference (the DiT kernel's scheme) ------------------------------
// On gfx950 the f32 MFMA executes on the vector lanes (measured: MFMA-busy + VALU-busy cycles add
// up, SQ_VALU_MFMA_COEXEC = 0), so softmax VALU instructions are paid 1:1 in matrix time.  The
// common path therefore does the minimum: the score accumulator is INITIALISED with -m_ref (a
// 16-register copy kept per tile: the MFMA C operand may differ from D, so the subtraction is
// free), then 16 v_exp + a 16-term sum; no per-block max, no output rescale.  m_ref is the true
// running max as of the last re-reference; a block is re-referenced (classic online-softmax
// step, recomputed from raw scores) only when its exponentials show that m_ref is stale by
// more than 2^60 -- detected a posteriori from the row sum, wave-uniformly.  Block 0 has no reference
// (m_ref = -inf): it enters the reference-setting path directly, skipping the common path.
constexpr float SM_BIG = 1.152921504606847e18f;   // 2^60


struct TileState {
    f32x16 negm;   // -m_ref in all 16 registers (C operand of the first QK MFMA)
    float m_ref;
    float l_lane;
};

// in place: st <- 2^st, returns the lane's 16-term sum
__device__ __forceinline__ float exp_sum(f32x16& st) {
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = __builtin_amdgcn_exp2f(st[r]);
    return ((st[0] + st[1]) + (st[2] + st[3])) + ((st[4] + st[5]) + (st[6] + st[7])) +
           (((st[8] + st[9]) + (st[10] + st[11])) + ((st[12] + st[13]) + (st[14] + st[15])));
}

// rare path: raw scores (C = 0) -> new reference, rescale the running sum / output, P^T in st
__device__ __forceinline__ float rereference(const f32x4 (&kf)[4], const f32x4 (&q)[4], f32x16& st,
                                             f32x16& ot, TileState& t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) st = mfma32(kf[g][e], q[g][e], st);
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

// NT = number of query tiles this wave really owns (2, or 1 for the wave holding tile 14 alone).
// Order per key block jb, K fragments of jb already in registers:
//   ds_read V(jb) | QK_A, QK_B | exp+sum (A, B), overflow check | PV_A |
//   wait+barrier(jb+1), DMA(jb+3), ds_read K(jb+1) | PV_B
// so the LDS latencies and the barrier hide behind matrix work.
template <int NT>
__device__ __forceinline__ void attn_packed_body(f32x4* ring, const f32x4* qg, const f32x4* kg,
                                                 const f32x4* vg, f32x4* og, int bh, int t0, int lane,
                                                 int wave) {
    f32x4 qa[4], qb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        qa[g] = qg[(t0 * 4 + g) * 64 + lane] * QSCALE;
        if (NT == 2) qb[g] = qg[((t0 + 1) * 4 + g) * 64 + lane] * QSCALE;
    }
    f32x16 oa, ob;
    TileState ta, tb;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        oa[r] = ob[r] = 0.f;
        ta.negm[r] = tb.negm[r] = INFINITY;
    }
    ta.m_ref = tb.m_ref = -INFINITY;
    ta.l_lane = tb.l_lane = 0.f;

    // Blocks 0..2 were issued by the caller.  The DMA is kept exactly three blocks (= 6 glds per
    // wave) ahead for the WHOLE loop -- past the end it re-fetches the last block into a slot nobody
    // reads -- so every wait is the same counted `vmcnt(4)`.
    auto issue_clamped = [&](int jb) {
        const int src = jb < NKB ? jb : NKB - 1;
        f32x4* slot = ring + (jb & (ATT_SLOTS - 1)) * ATT_SLOT_F4;
        glds16_asm(kg + (src * 4 + wave) * 64 + lane, slot + wave * 64);
        glds16_asm(vg + (src * 4 + wave) * 64 + lane, slot + 256 + wave * 64);
    };
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue_clamped(3);
    f32x4 kf[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) kf[g] = ring[g * 64 + lane];

#pragma unroll 1
    for (int jb = 0; jb < NKB; ++jb) {
        const f32x4* slot = ring + (jb & (ATT_SLOTS - 1)) * ATT_SLOT_F4 + lane;
        f32x4 vf[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) vf[g] = slot[256 + g * 64];

        // scores relative to the sticky reference (C operand of the first MFMA, kept in its own registers).
        // Block 0 has no reference yet: it goes straight to the reference-setting path (one QK pass, not two).
        f32x16 sta = ta.negm, stb = tb.negm;
        float psa = 0.f, psb = 0.f;
        bool redo = jb == 0;                                  // scalar
        if (!redo) {
            sta = mfma32_from(kf[0][0], qa[0][0], ta.negm);
            if (NT == 2) stb = mfma32_from(kf[0][0], qb[0][0], tb.negm);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = (g == 0 ? 1 : 0); e < 4; ++e) {
                    sta = mfma32(kf[g][e], qa[g][e], sta);
                    if (NT == 2) stb = mfma32(kf[g][e], qb[g][e], stb);
                }
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
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) oa = mfma32(vf[g][e], sta[4 * g + e], oa);      // PV_A
        // ---- make block jb+1 visible, keep the DMA two blocks ahead, fetch its K fragments ----
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue_clamped(jb + 4);   // slot (jb+4)&3 == jb&3: every wave read K(jb), V(jb) before this barrier
        {
            const f32x4* nslot = ring + ((jb + 1) & (ATT_SLOTS - 1)) * ATT_SLOT_F4 + lane;
#pragma unroll
            for (int g = 0; g < 4; ++g) kf[g] = nslot[g * 64];
        }
        if (NT == 2) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) ob = mfma32(vf[g][e], stb[4 * g + e], ob);  // PV_B
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the trailing (unused) DMAs

    // ---- normalise and store O[query][d]: lane (query i, half) holds d = 8g + 4*half + e ----
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
}

// PARTS = 2: two workgroups per (sequence, head), two query tiles per wave.  PARTS = 4 (launches too small to fill the chip
// otherwise): four workgroups per head, ONE query tile per wave -- twice the waves per SIMD for the same matrix work, K / V
// streamed four times per head from L2.  Tile 15 does not exist: the wave that would own it repeats tile 14 (same bits,
// stored twice).
template <int PARTS>
__global__ __launch_bounds__(256, 2) void attn_fwd_packed_kernel(const float* __restrict__ q,
                                                                 const float* __restrict__ k,
                                                                 const float* __restrict__ vT,
                                                                 float* __restrict__ o, int BH) {
    extern __shared__ __attribute__((aligned(16))) f32x4 ring[];
    // The workgroups of a (sequence, head) stream the same K/V: give them ids r, r+8, ... of a
    // (8 PARTS)-id group so that (round-robin XCD placement) they share an L2.  Speed only.
    const int grp = blockIdx.x / (8 * PARTS), rr = blockIdx.x % (8 * PARTS);
    const int bh = grp * 8 + (rr & 7);
    const int part = rr >> 3;                 // query tiles [8*part, 8*part+8) (PARTS = 2) / [4*part, 4*part+4) (PARTS = 4)
    if (bh >= BH) return;                     // whole workgroup (grid is padded to whole groups)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const f32x4* qg = reinterpret_cast<const f32x4*>(q) + (size_t)bh * NKB * 256;
    const f32x4* kg = reinterpret_cast<const f32x4*>(k) + (size_t)bh * NKB * 256;
    const f32x4* vg = reinterpret_cast<const f32x4*>(vT) + (size_t)bh * NKB * 256;
    f32x4* og = reinterpret_cast<f32x4*>(o);

    // this wave DMAs K fragment {wave} and V^T fragment {wave} of every block (8 pieces / 4 waves)
    {
        f32x4* slot0 = ring;
        f32x4* slot1 = ring + ATT_SLOT_F4;
        glds16_asm(kg + (0 * 4 + wave) * 64 + lane, slot0 + wave * 64);
        glds16_asm(vg + (0 * 4 + wave) * 64 + lane, slot0 + 256 + wave * 64);
        glds16_asm(kg + (1 * 4 + wave) * 64 + lane, slot1 + wave * 64);
        glds16_asm(vg + (1 * 4 + wave) * 64 + lane, slot1 + 256 + wave * 64);
        f32x4* slot2 = ring + 2 * ATT_SLOT_F4;
        glds16_asm(kg + (2 * 4 + wave) * 64 + lane, slot2 + wave * 64);
        glds16_asm(vg + (2 * 4 + wave) * 64 + lane, slot2 + 256 + wave * 64);
    }
    if constexpr (PARTS == 2) {
        const int t0 = part * 8 + wave * 2;       // query tiles t0, t0+1 (tile 15 does not exist)
        if (t0 + 1 < NKB)
            attn_packed_body<2>(ring, qg, kg, vg, og, bh, t0, lane, wave);
        else
            attn_packed_body<1>(ring, qg, kg, vg, og, bh, t0, lane, wave);
    } else {
        const int t = part * 4 + wave;
        attn_packed_body<1>(ring, qg, kg, vg, og, bh, t < NKB ? t : NKB - 1, lane, wave);
    }
}

// ------------------------------------------------------------------ persistent variant
// Measured with in-kernel stamps (tools/probe_attn.py, probe_clock.py): the key-block loop runs at
// ~96 % of the MFMA roof while two waves share a SIMD, but with two independent 4-wave workgroups
// per CU the hardware's age-priority arbitration lets the older workgroup run at its stand-alone
// speed and starves the younger one, which then finishes alone at ~62 % (bimodal durations,
// 331 vs 496 us).  So here ONE 8-wave workgroup per CU stays resident and walks the heads: both
// waves of every SIMD belong to the same workgroup and meet at the per-block barrier, which
// enforces fair progress; a head's 15 query tiles map to the 8 waves x 2 tiles (the 16th slot is
// skipped), K/V are streamed once per head, and the DMA ring / block loop run CONTINUOUSLY across
// heads (next head's first blocks already in flight, its Q fragments prefetched).
constexpr int PERSIST_THREADS = 512;

template <int NT>
__device__ __forceinline__ void attn_persistent_body(f32x4* ring, const f32x4* qall, const f32x4* kall,
                                                     const f32x4* vall, f32x4* og, int BH, int lane, int wave) {
    const int stride = gridDim.x;
    const int t0 = wave * 2;
    // `jb` may run past 14 into the following heads of this workgroup.
    // The wave that owns only ONE query tile (wave 7, the only NT == 1 instance: its SIMD carries 3 tiles where the others
    // carry 4) issues ALL eight fragments of a block; the two-tile waves issue none.  An LDS-DMA instruction costs ~100 issue
    // cycles, and in this kernel vector issue does not overlap the f32 MFMAs: one per block and wave on the critical SIMDs was
    // 2 % of the attention time (round 5: 500 against 509 us per launch, headline +0.7 %, profiles/r05_attn_dma_w7_ab.txt).
    // Counted wait of wave 7: blocks j+2 and j+3 (2 x 8 instructions) may stay in flight.
    auto issue_block = [&](int bh, int jb, int gslot) {
        if constexpr (NT == 1) {
            if (jb >= NKB) { jb -= NKB; bh += stride; }
            if (bh >= BH) { bh -= stride; jb = NKB - 1; }   // past the end: harmless re-fetch
            const size_t off = (size_t)bh * NKB * 256 + jb * 4 * 64 + lane;
            f32x4* dst = ring + (gslot & (ATT_SLOTS - 1)) * ATT_SLOT_F4;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                glds16_asm(kall + off + f * 64, dst + f * 64);
                glds16_asm(vall + off + f * 64, dst + 256 + f * 64);
            }
        }
    };
#define ATT_PERSIST_WAIT() if constexpr (NT == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory")
    int bh = blockIdx.x;
    int gb = 0;                                      // global block counter -> ring slot
    issue_block(bh, 0, 0);
    issue_block(bh, 1, 1);
    issue_block(bh, 2, 2);

    f32x4 qa[4], qb[4], qna[4], qnb[4];
    {
        const f32x4* qg = qall + (size_t)bh * NKB * 256;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            qa[g] = qg[(t0 * 4 + g) * 64 + lane] * QSCALE;
            if (NT == 2) qb[g] = qg[((t0 + 1) * 4 + g) * 64 + lane] * QSCALE;
            qna[g] = qa[g];
            qnb[g] = qb[g];
        }
    }
    ATT_PERSIST_WAIT();
    __builtin_amdgcn_s_barrier();
    issue_block(bh, 3, 3);
    f32x4 kf[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) kf[g] = ring[g * 64 + lane];

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
            const f32x4* slot = ring + (gb & (ATT_SLOTS - 1)) * ATT_SLOT_F4 + lane;
            f32x4 vf[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) vf[g] = slot[256 + g * 64];
            // block 0 of a head has no reference yet: straight to the reference-setting path (one QK pass, not two)
            f32x16 sta = ta.negm, stb = tb.negm;
            float psa = 0.f, psb = 0.f;
            bool redo = jb == 0;                                  // scalar
            if (!redo) {
                sta = mfma32_from(kf[0][0], qa[0][0], ta.negm);
                if (NT == 2) stb = mfma32_from(kf[0][0], qb[0][0], tb.negm);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = (g == 0 ? 1 : 0); e < 4; ++e) {
                        sta = mfma32(kf[g][e], qa[g][e], sta);
                        if (NT == 2) stb = mfma32(kf[g][e], qb[g][e], stb);
                    }
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
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) oa = mfma32(vf[g][e], sta[4 * g + e], oa);      // PV_A
            // prefetch the next head's Q fragments (a whole block old by the next counted wait, so
            // the `vmcnt(2)` below never really waits on them)
            if (jb == NKB - 4 && bh + stride < BH) {
                const f32x4* qg = qall + (size_t)(bh + stride) * NKB * 256;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    qna[g] = qg[(t0 * 4 + g) * 64 + lane];
                    if (NT == 2) qnb[g] = qg[((t0 + 1) * 4 + g) * 64 + lane];
                }
            }
            // ---- make block gb+1 visible, keep the DMA three blocks ahead, fetch its K fragments
            ATT_PERSIST_WAIT();
            __builtin_amdgcn_s_barrier();
            issue_block(bh, jb + 4, gb + 4);     // slot (gb+4)&3 == gb&3: all its reads are done
            {
                const f32x4* nslot = ring + ((gb + 1) & (ATT_SLOTS - 1)) * ATT_SLOT_F4 + lane;
#pragma unroll
                for (int g = 0; g < 4; ++g) kf[g] = nslot[g * 64];
            }
            if (NT == 2) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) ob = mfma32(vf[g][e], stb[4 * g + e], ob);  // PV_B
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
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            qa[g] = qna[g] * QSCALE;
            if (NT == 2) qb[g] = qnb[g] * QSCALE;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the trailing (unused) DMAs
}

__global__ __launch_bounds__(PERSIST_THREADS, 2) void attn_fwd_persistent_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ vT,
    float* __restrict__ o, int BH) {
    extern __shared__ __attribute__((aligned(16))) f32x4 ring[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const f32x4* qg = reinterpret_cast<const f32x4*>(q);
    const f32x4* kg = reinterpret_cast<const f32x4*>(k);
    const f32x4* vg = reinterpret_cast<const f32x4*>(vT);
    f32x4* og = reinterpret_cast<f32x4*>(o);
    if (wave < 7)
        attn_persistent_body<2>(ring, qg, kg, vg, og, BH, lane, wave);
    else
        attn_persistent_body<1>(ring, qg, kg, vg, og, BH, lane, wave);   // tiles 14 only (15 is void)
}

int launch_attn_packed(const float* q, const float* k, const float* vT, float* o, int BH, hipStream_t st) {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n_cu = prop.multiProcessorCount;
        else
            n_cu = 256;
    }
    // persistent from THREE heads per CU on; below that the two-workgroups-per-head kernel spreads a head over two CUs and
    // shares a CU with the other sampler lane's kernels (a persistent workgroup fills the register file of its CU).
    // Series/s with two sampler lanes, packed / persistent (tools/strong_probe.py, T2S_ATTN_PERSIST_MIN): 256 heads per
    // launch (64 series) 61.5 / 59.8, 512 (128 series) 63.3-64.0 / 62.3-62.5, 768 (192) 62.3-62.6 / 63.2, 1024 (256)
    // 63.1-63.4 / 63.9-64.1.  (Alone on the chip the persistent kernel wins from one head per CU: 69 vs 75 us at 256 heads --
    // the switch sat there while the 32-series shard ran as one chain.)  T2S_ATTN_PERSIST_MIN=<heads> moves the switch.
    static const int persist_min = getenv("T2S_ATTN_PERSIST_MIN") ? atoi(getenv("T2S_ATTN_PERSIST_MIN")) : 0;
    if (BH >= (persist_min > 0 ? persist_min : 3 * n_cu)) {
        // persistent: one 8-wave workgroup per CU walks the heads.  Never more workgroups than heads: a workgroup without a head
        // of its own would start its K / V ring at a negative head index (the default switch keeps BH >= 3 n_cu; the A/B
        // switch T2S_ATTN_PERSIST_MIN can put fewer heads than CUs here -- found by tests/test_hip_contracts.py in round 5)
        attn_fwd_persistent_kernel<<<BH < n_cu ? BH : n_cu, PERSIST_THREADS, ATT_LDS_BYTES, st>>>(q, k, vT, o, BH);
    } else {
        // four workgroups per head while two per head would leave CUs without a workgroup of this launch (fewer than 128 heads:
        // 16 sequences, i.e. 8 series with CFG): series/s at 8 series 27.5 against 22.8 (+20 %); from 128 heads on two per head
        // are as good or better (32 series: 57.9 against 56.3) -- profiles/r05_attn_parts_ab.txt.  T2S_ATTN_PARTS=2 / 4 fixes it (A/B).
        static const int parts_env = getenv("T2S_ATTN_PARTS") ? atoi(getenv("T2S_ATTN_PARTS")) : 0;
        const int parts = parts_env == 2 || parts_env == 4 ? parts_env : (((BH + 7) / 8) * 16 < n_cu ? 4 : 2);
        if (parts == 4)
            attn_fwd_packed_kernel<4><<<((BH + 7) / 8) * 32, 256, ATT_LDS_BYTES, st>>>(q, k, vT, o, BH);
        else
            attn_fwd_packed_kernel<2><<<((BH + 7) / 8) * 16, 256, ATT_LDS_BYTES, st>>>(q, k, vT, o, BH);
    }
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
        const float inv = 1.0f / pair_sum(s.l_lane);
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
