// Attention of the bf16 TRAINING path: forward with log-sum-exp and the flash-style backward, all
// contractions on v_mfma_f32_32x32x16_bf16 (fp32 accumulate), softmax statistics in fp32.
// Same mathematics as t2s_attn_bwd.hip (timm 1.0.11 Attention core; reference call site
// model/denoiser/transformer.py:116 under autograd, train.py:123-125).
//
// q, k, v: bf16 (BH, 480, 32), bh = seq*4 + head; q is PRE-SCALED by log2(e)/sqrt(32) (ATT_QS, applied by the qkv GEMM's
// epilogue), so q.k is the score in the log2 domain and the softmax reference / log-sum-exp / D_i are subtracted for free
// as the C operand of the score MFMAs (mfma16_from).  o, do: bf16 token rows (S*480, 128), head h at
// columns 32h..32h+31.  dqkv: bf16 token rows (S*480, 384) = [dq | dk | dv] x heads.  lse: fp32
// (BH, 480) in the log2 domain of the scaled scores.
//
// One workgroup (8 waves) per (sequence, head); two whole (480 x 32) operands live in LDS as bf16
// "images" with 64-byte rows whose 16-byte chunks are XOR-swizzled by (row>>2)&3.  That one image
// serves both kinds of read without bank conflicts:
//   row read  (ds_read_b128)        -> MFMA operand with the image ROW on the lane     (K, Q, dO, V)
//   col read  (ds_read_b64_tr_b16)  -> MFMA operand with the image COLUMN on the lane  (K^T, V^T, Q^T, dO^T)
// The 32x32 score tile never leaves registers: the fp32 accumulator, converted pairwise to bf16,
// is the next MFMA's operand (acc_frag; its permuted k order is matched by the column reads).
#include "t2s_x3.h"   // T2S_X3_KERNEL (no packed fp32: it serialises with bf16 MFMAs), wg_sync; includes t2s_bf16.h

namespace t2s {

namespace {
constexpr int NKB = NTOK / 32;                 // 15 tiles of 32 tokens
constexpr int IMG = NTOK * 64;                 // bytes of one (480 x 32) bf16 image
constexpr float SCALE = 0.17677669529663687f;  // 32^-0.5
constexpr float LN2 = 0.6931471805599453f;     // SCALE / ATT_QS: dK = dS^T q SCALE = dS^T (q ATT_QS) ln 2

__device__ __forceinline__ int img_off(int row, int chunk) { return row * 64 + 16 * (chunk ^ ((row >> 2) & 3)); }

__device__ __forceinline__ float pair_max_f(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float pair_sum_f(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Store the 32 features of this lane pair's row (accumulator layout: the lane of half h holds features 8g + 4h + e) as
// bf16 in TWO 16-byte pieces per lane: the halves first trade pieces with v_permlane32_swap so that half 0 owns features
// 0..15 and half 1 features 16..31 -- a contiguous 64 bytes per row instead of eight scattered 8-byte pieces (which reach
// HBM as partial lines at less than half the store rate).  `row32` points at feature 0 of the row.
__device__ __forceinline__ void store_row32(__bf16* row32, const f32x16& acc, float scale, int half) {
    uint32_t d[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 w = {acc[4 * g] * scale, acc[4 * g + 1] * scale, acc[4 * g + 2] * scale, acc[4 * g + 3] * scale};
        const bf16x4 pk = pack4(w);
        d[g][0] = __builtin_bit_cast(uint2, pk).x;
        d[g][1] = __builtin_bit_cast(uint2, pk).y;
    }
    uint4 lo, hi;   // after the swaps: [features 8G..8G+3 | 8G+4..8G+7] for G = 2 half (lo) and 2 half + 1 (hi)
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        const auto a = __builtin_amdgcn_permlane32_swap(d[0][w], d[2][w], false, false);   // pieces g = 0 and 2
        const auto b = __builtin_amdgcn_permlane32_swap(d[1][w], d[3][w], false, false);   // pieces g = 1 and 3
        (&lo.x)[w] = a[0];
        (&lo.x)[2 + w] = a[1];
        (&hi.x)[w] = b[0];
        (&hi.x)[2 + w] = b[1];
    }
    uint4* dst = reinterpret_cast<uint4*>(row32 + 16 * half);
    dst[0] = lo;
    dst[1] = hi;
}

// stage 480 rows of 32 bf16 (64 B, src row stride in elements) into a swizzled image
__device__ __forceinline__ void stage_img(char* img, const __bf16* src, int src_stride, int tid, int nthreads) {
    for (int idx = tid; idx < NTOK * 4; idx += nthreads) {
        const int row = idx >> 2, c = idx & 3;
        *reinterpret_cast<bf16x8*>(img + img_off(row, c)) = *reinterpret_cast<const bf16x8*>(src + (size_t)row * src_stride + c * 8);
    }
}

// The same image filled by LDS-DMA (no registers, asynchronous): piece p of 30 = rows [16p, 16p+16) = 1 KiB; lane l writes
// LDS bytes [16 l, 16 l + 16) of the piece = row 16p + (l>>2), swizzled chunk l&3, so it FETCHES chunk (l&3) ^ ((row>>2)&3).
// hipcc compiled stage_img's loop as load -> s_waitcnt vmcnt(0) -> ds_write per iteration: eight serialized HBM round trips
// per workgroup before its first MFMA (a forward kernel run with ONE key block instead of 15 still took 63 % of the time).
__device__ __forceinline__ void dma_img_piece(char* img, const __bf16* src, int src_stride, int p, int lane) {
    const int row = 16 * p + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    glds16(reinterpret_cast<const f32x4*>(src + (size_t)row * src_stride + chunk * 8), reinterpret_cast<f32x4*>(img + p * 1024));
}
// (With the staging by DMA the forward kernel is, per launch, ~145 us of "skeleton" -- K / V / Q in, O out: 566 MB = 113 us at
// 5 TB/s -- plus ~90 us of key-block compute that the two workgroups a CU holds overlap only partly.  Prefetching the K / V
// of the head that takes the slot next into the caches while this one computes: no gain, 0.91 vs 0.89 ms per step.)
// two images (60 pieces) over the 8 waves of a workgroup; complete after s_waitcnt vmcnt(0) + barrier
__device__ __forceinline__ void dma_two_images(char* img_a, const __bf16* src_a, int stride_a, char* img_b, const __bf16* src_b,
                                               int stride_b, int wave, int lane) {
    for (int p = wave; p < 60; p += 8) {
        if (p < 30) dma_img_piece(img_a, src_a, stride_a, p, lane);
        else dma_img_piece(img_b, src_b, stride_b, p - 30, lane);
    }
}

// operand with the image row (token base + lane&31) on the lane: k = feature 16s + 8h + 0..7
__device__ __forceinline__ bf16x8 row_frag(const char* img, int base, int lane, int s) {
    return *reinterpret_cast<const bf16x8*>(img + img_off(base + (lane & 31), 2 * s + (lane >> 5)));
}

// operand with the image column (feature lane&31) on the lane: k = token base + 16s + 8(j>>2) + 4h + (j&3)
__device__ __forceinline__ bf16x8 col_frag(const char* img, int base, int lane, int s) {
    const int grp = (lane >> 4) & 3, dhalf = grp & 1, h = grp >> 1, q = (lane & 15) >> 2, p = lane & 3;
    const int row = base + 16 * s + 4 * h + q;                 // (row >> 2) & 3 is the same for the 4 rows of the block
    const int chunk = 2 * dhalf + (p >> 1);
    const char* a0 = img + img_off(row, chunk) + 8 * (p & 1);
    const char* a1 = img + img_off(row + 8, chunk) + 8 * (p & 1);
    return join_tr(lds_tr16(a0), lds_tr16(a1));
}
}  // namespace

// ------------------------------------------------------------------ forward with lse
__global__ __launch_bounds__(512) T2S_X3_KERNEL void attn16_fwd_kernel(const __bf16* __restrict__ q, const __bf16* __restrict__ k,
                                                         const __bf16* __restrict__ v, __bf16* __restrict__ o_rows,
                                                         float* __restrict__ lse, int reverse) {
    __shared__ __attribute__((aligned(16))) char smem[2 * IMG];
    char* Ks = smem;
    char* Vs = smem + IMG;
    const int bh = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, half = lane >> 5, i = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int seq = bh / NH, head = bh % NH;
    dma_two_images(Ks, k + (size_t)bh * NTOK * DH, DH, Vs, v + (size_t)bh * NTOK * DH, DH, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_sync();
    const __bf16* qg = q + (size_t)bh * NTOK * DH;
    for (int qt = wave; qt < NKB; qt += 8) {
        const int tok = qt * 32 + i;
        bf16x8 qf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qg + (size_t)tok * DH + 16 * s + 8 * half);
        f32x16 ot;
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[r] = 0.f;
        // Sticky-reference softmax: the reference m_ref is the row max of the FIRST key block; later blocks
        // are exponentiated against it without a max or an output rescale (the kernel is VALU-bound:
        // that is a third of its vector work).  A block whose row sum shows the reference is stale by
        // more than 2^40 re-references the running state (classic online-softmax step), wave-uniformly.
        // (Round 2 measured hand-pipelined LDS reads here -- K fragments of block jb+1 fetched behind the score MFMAs of
        // block jb, V^T fragments before the exponentials, four independent row-sum chains: no gain, 0.96 vs 0.94 ms per
        // step in a same-box A/B; the other waves of the SIMD already cover those round trips.  PMC view of this kernel
        // (tools/pmc_train.sh): VALU busy 56 %, matrix pipe 24 % (half of it co-executing), 31 % of the cycles neither,
        // 2.7 of 4 wave slots occupied on average -- the per-workgroup K / V staging and the 15-tiles-over-8-waves tail.
        // A persistent variant was built to remove exactly that (one 16-wave workgroup per CU walking 18 heads, the next
        // head's K / V images arriving by LDS-DMA into a second buffer, Q fetched a head ahead, counted vmcnt so the
        // output stores stay in flight across the per-head barrier): 1.01 vs 0.93 ms per step -- SLOWER; the barrier puts the
        // 16 waves back into lockstep at every head (offsetting them with s_sleep changed nothing).  Dropped.
        // Also without effect: a one-time half-block s_sleep offset between the two waves a SIMD holds of a workgroup, and a
        // software pipeline that issues the score MFMAs of block jb+1 before the exponentials of block jb (two alternating
        // accumulators; 0.91 vs 0.91 ms), and a staged arrival of the images (four 16 KiB stages, the first query tile gated per
        // stage with counted vmcnt + s_barrier so that it starts on the first 128 keys: 0.98 vs 0.89 ms -- every extra barrier
        // re-aligns the eight waves, whose drift apart is what overlaps their exp and MFMA phases).  What is left is the exponential itself: 16 v_exp per 4 MFMAs at head_dim 32 --
        // the kernel does 4.7 T exp/s, the chip's v_exp issue rate is ~20 T/s only if nothing else used the port.)
        float m_run = 0.f, l_lane = 0.f;    // m_run: the reference, in the log2 domain of the (pre-scaled) scores
        f32x16 negm;                        // -m_run in all 16 registers: the C operand of every score MFMA
        for (int jb = 0; jb < NKB; ++jb) {
            if (jb == 0) {                  // reference = row max of the first key block (two extra MFMAs per tile)
                f32x16 raw;
#pragma unroll
                for (int r = 0; r < 16; ++r) raw[r] = 0.f;
#pragma unroll
                for (int s = 0; s < 2; ++s) raw = mfma16(row_frag(Ks, 0, lane, s), qf[s], raw);
                float mloc = raw[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, raw[r]);
                m_run = pair_max_f(mloc);
#pragma unroll
                for (int r = 0; r < 16; ++r) negm[r] = -m_run;
            }
            f32x16 st = mfma16_from(row_frag(Ks, jb * 32, lane, 0), qf[0], negm);   // S^T[key][query] - m_run
            st = mfma16(row_frag(Ks, jb * 32, lane, 1), qf[1], st);
            f32x16 pt;
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pt[r] = __builtin_amdgcn_exp2f(st[r]);
                ps += pt[r];
            }
            if (__builtin_amdgcn_ballot_w64(!(ps < 1.0995116e12f)) != 0) {      // 2^40; also catches inf / NaN
                float mloc = st[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, st[r]);
                const float up = fmaxf(0.f, pair_max_f(mloc));                  // the new reference is m_run + up
                const float alpha = __builtin_amdgcn_exp2f(-up);
                m_run += up;
                ps = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    negm[r] = -m_run;
                    pt[r] = __builtin_amdgcn_exp2f(st[r] - up);
                    ps += pt[r];
                    ot[r] *= alpha;
                }
                l_lane *= alpha;
            }
            l_lane += ps;
#pragma unroll
            for (int s = 0; s < 2; ++s) ot = mfma16(col_frag(Vs, jb * 32, lane, s), acc_frag(pt, s), ot);   // O^T += V^T P^T
        }
        const float l_tot = pair_sum_f(l_lane);
        const float inv = 1.0f / l_tot;
        store_row32(o_rows + ((size_t)seq * NTOK + tok) * D + head * DH, ot, inv, half);
        if (half == 0) lse[(size_t)bh * NTOK + tok] = m_run + __builtin_amdgcn_logf(l_tot);   // v_log_f32 = log2
    }
}

// ------------------------------------------------------------------ kernel A: dQ (queries on lanes)
__global__ __launch_bounds__(512) T2S_X3_KERNEL void attn16_bwd_dq_kernel(const __bf16* __restrict__ q, const __bf16* __restrict__ k,
                                                            const __bf16* __restrict__ v, const __bf16* __restrict__ o_rows,
                                                            const __bf16* __restrict__ do_rows,
                                                            const float* __restrict__ lse, float* __restrict__ dsum,
                                                            __bf16* __restrict__ dqkv, int reverse) {
    __shared__ __attribute__((aligned(16))) char smem[2 * IMG];
    char* Ks = smem;
    char* Vs = smem + IMG;
    const int bh = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, half = lane >> 5, i = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int seq = bh / NH, head = bh % NH;
    dma_two_images(Ks, k + (size_t)bh * NTOK * DH, DH, Vs, v + (size_t)bh * NTOK * DH, DH, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_sync();
    const __bf16* qg = q + (size_t)bh * NTOK * DH;
    for (int qt = wave; qt < NKB; qt += 8) {
        const int tok = qt * 32 + i;
        const __bf16* dorow = do_rows + ((size_t)seq * NTOK + tok) * D + head * DH + 8 * half;
        bf16x8 qf[2], dof[2];   // B operands: Q^T and dO^T of this lane's query
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qf[s] = *reinterpret_cast<const bf16x8*>(qg + (size_t)tok * DH + 16 * s + 8 * half);
            dof[s] = *reinterpret_cast<const bf16x8*>(dorow + 16 * s);
        }
        const float lse_i = lse[(size_t)bh * NTOK + tok];
        // D_i = sum_d dO[i][d] O[i][d]: each lane half holds 16 of the query's 32 features; published for the dK/dV
        // kernel (which runs next on the same stream) instead of a separate pass over o and do
        float d_i = 0.f;
        {
            const __bf16* orow = o_rows + ((size_t)seq * NTOK + tok) * D + head * DH + 8 * half;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f32x8 ov = unpack8(*reinterpret_cast<const bf16x8*>(orow + 16 * s)), dv8 = unpack8(dof[s]);
#pragma unroll
                for (int e = 0; e < 8; ++e) d_i += ov[e] * dv8[e];
            }
            d_i = pair_sum_f(d_i);
            if (half == 0) dsum[(size_t)bh * NTOK + tok] = d_i;
        }
        f32x16 dq, nl, nd;     // nl / nd: -lse_i / -D_i in all 16 registers, the C operands of the score / dP MFMAs
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dq[r] = 0.f;
            nl[r] = -lse_i;
            nd[r] = -d_i;
        }
        for (int jb = 0; jb < NKB; ++jb) {
            f32x16 st = mfma16_from(row_frag(Ks, jb * 32, lane, 0), qf[0], nl);    // S^T[key][query] - lse (log2 domain)
            f32x16 dp = mfma16_from(row_frag(Vs, jb * 32, lane, 0), dof[0], nd);   // dP^T[key][query] - D = V dO^T - D
            st = mfma16(row_frag(Ks, jb * 32, lane, 1), qf[1], st);
            dp = mfma16(row_frag(Vs, jb * 32, lane, 1), dof[1], dp);
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = __builtin_amdgcn_exp2f(st[r]) * dp[r];   // dS^T = P (dP - D)
#pragma unroll
            for (int s = 0; s < 2; ++s) dq = mfma16(col_frag(Ks, jb * 32, lane, s), acc_frag(st, s), dq);   // dQ^T += K^T dS^T
        }
        store_row32(dqkv + ((size_t)seq * NTOK + tok) * (3 * D) + head * DH, dq, SCALE, half);   // dq block
    }
}

// ------------------------------------------------------------------ kernel B: dK, dV (keys on lanes)
__global__ __launch_bounds__(512) T2S_X3_KERNEL void attn16_bwd_dkv_kernel(const __bf16* __restrict__ q, const __bf16* __restrict__ k,
                                                             const __bf16* __restrict__ v, const __bf16* __restrict__ do_rows,
                                                             const float* __restrict__ lse, const float* __restrict__ dsum,
                                                             __bf16* __restrict__ dqkv, int reverse) {
    __shared__ __attribute__((aligned(16))) char smem[2 * IMG + 2 * NTOK * 4];
    char* Qs = smem;                                              // Q image
    char* Os = smem + IMG;                                        // dO image (this head)
    float* Ls = reinterpret_cast<float*>(smem + 2 * IMG);         // lse (log2 domain)
    float* Ds = Ls + NTOK;                                        // D_i
    const int bh = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, half = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int seq = bh / NH, head = bh % NH;
    dma_two_images(Qs, q + (size_t)bh * NTOK * DH, DH, Os, do_rows + (size_t)seq * NTOK * D + head * DH, D, wave, lane);
    for (int t = tid; t < NTOK; t += 512) {
        Ls[t] = -lse[(size_t)bh * NTOK + t];      // negated: they initialise the accumulators (C operands) below
        Ds[t] = -dsum[(size_t)bh * NTOK + t];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_sync();
    const __bf16* kg = k + (size_t)bh * NTOK * DH;
    const __bf16* vg = v + (size_t)bh * NTOK * DH;
    for (int kb = wave; kb < NKB; kb += 8) {
        const int key = kb * 32 + j;
        bf16x8 kf[2], vf[2];   // B operands: K^T and V^T of this lane's key
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            kf[s] = *reinterpret_cast<const bf16x8*>(kg + (size_t)key * DH + 16 * s + 8 * half);
            vf[s] = *reinterpret_cast<const bf16x8*>(vg + (size_t)key * DH + 16 * s + 8 * half);
        }
        f32x16 dk, dv;
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[r] = dv[r] = 0.f;
        for (int qb = 0; qb < NKB; ++qb) {
            // accumulators start at the per-register query statistics (queries 8g + 4 half + 0..3 of this block):
            // sc = S - lse, dp = dP - D come straight out of the MFMAs
            f32x16 sc, dp;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(Ls + qb * 32 + 8 * g + 4 * half);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(Ds + qb * 32 + 8 * g + 4 * half);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sc[4 * g + e] = l4[e];
                    dp[4 * g + e] = d4[e];
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                sc = mfma16(row_frag(Qs, qb * 32, lane, s), kf[s], sc);    // S[query][key] - lse (registers = queries)
                dp = mfma16(row_frag(Os, qb * 32, lane, s), vf[s], dp);    // dP[query][key] - D = dO V^T - D
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(sc[r]);
                dp[r] = p * dp[r];             // dS[query][key]
                sc[r] = p;                     // P[query][key]
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                dv = mfma16(col_frag(Os, qb * 32, lane, s), acc_frag(sc, s), dv);   // dV^T += dO^T P
                dk = mfma16(col_frag(Qs, qb * 32, lane, s), acc_frag(dp, s), dk);   // dK^T += Q^T dS
            }
        }
        __bf16* dst = dqkv + ((size_t)seq * NTOK + key) * (3 * D) + head * DH;
        store_row32(dst + D, dk, LN2, half);        // Qs holds q ATT_QS: dK = dS^T (q ATT_QS) ln 2
        store_row32(dst + 2 * D, dv, 1.0f, half);
    }
}

// ------------------------------------------------------------------ fused backward: dQ, dK and dV in one pass over the scores
// The two kernels above each recompute S, exp and dP (the exponentials are what bounds them) and between them read q, k, v,
// do twice.  Here a PERSISTENT 8-wave workgroup per CU walks its heads; within a head wave w OWNS key blocks 2w and 2w + 1
// (their K^T / V^T operand fragments and dK^T / dV^T accumulators stay in its registers) and the 15 query blocks are taken in
// two PHASES of 8 (the second: 7 + a skipped one).  In a phase the waves visit the 8 blocks in ROTATED order (step s: block
// (w + s) mod 8), so at any step they are on 8 different blocks of the Q / dO half-images in LDS.  Per step and key block:
// S, dP (registers = queries, as in kernel B), P = exp2, dS = P (dP - D), dV^T += dO^T P, dK^T += Q^T dS; then dS -- whose
// contraction index for dQ, the key, sits on the lanes -- crosses a private 2 KiB LDS tile as bf16 (written as rows, read
// back column-wise by ds_read_b64_tr_b16) and  dQ^T(block) += K^T dS^T  takes two more MFMAs; the two key blocks of a wave
// add up in registers.  Those partials are summed over the waves in an fp32 LDS accumulator [block][register quad][lane]
// (16-byte pieces, conflict-free) in a FIXED order: block b takes the contribution of step s as its s-th, each wave
// waiting for flag[b] == s before its read-add-write and publishing s + 1 after -- a ring of neighbour waits, not a
// barrier per step: bit-reproducible, and a wave only ever waits for an EARLIER step of another wave (no cycle).
// HBM never waits for compute: the next phase's Q / dO / O half-images and log-sum-exps arrive by untracked LDS-DMA into the
// other buffer while this phase computes (D_i = sum dO.O is formed from the images between the two barriers of a phase
// boundary and never leaves LDS); dq / dk / dv leave as fire-and-forget stores.
// 7 score-sized MFMA pairs and 2 x 16 exponentials per (query, key) tile become 5 and 16; q, k, v, do cross HBM once.
// History (same box, ms per step for the 4 launches; the two kernels: 2.17): one head per 15-wave workgroup, one key block
// per wave, ds_add_f32 into the accumulator 21.7 (the LDS atomic unit); read-add-write 2.6; two key blocks per wave 2.53 --
// of which 1.07 is a workgroup's loads and stores with nothing to overlap them (one workgroup per CU: timing build with one
// step instead of 15), 0.81 the score / dK / dV part, 0.29 the dS tile + dQ MFMAs, 0.36 the ordered accumulation.
constexpr int FP_IMG = 256 * 64;                    // half image: 8 query blocks of one (head, operand)
constexpr int FP_QS = 0;                            // Q half-images [2]
constexpr int FP_OS = 2 * FP_IMG;                   // dO half-images [2]
constexpr int FP_OB = 4 * FP_IMG;                   // O half-image (consumed at the phase boundary)
constexpr int FP_ACC = 5 * FP_IMG;                  // dQ^T partial sums, 8 blocks x 4 KiB
constexpr int FP_TILE = FP_ACC + 8 * 4096;          // 16 tiles of 2 KiB
constexpr int FP_LS = FP_TILE + 16 * 2048;          // -lse [2][256]
constexpr int FP_DS = FP_LS + 2 * 256 * 4;          // -D_i [2][256]
constexpr int FP_LRAW = FP_DS + 2 * 256 * 4;        // lse as fetched [256]
constexpr int FP_FLAG = FP_LRAW + 256 * 4;          // [8]
constexpr int FP_DUMP = FP_FLAG + 64;              // 1 KiB nobody reads (cache warming by DMA)
constexpr int FP_LDS = FP_DUMP + 1024;

__device__ __forceinline__ f32x16 mfma16_zero(bf16x8 a, bf16x8 b) {   // C = 0 as an inline constant: no 16 v_mov
    f32x16 d;
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
    return d;
}

// the 16 dS registers (as the two bf16 operand fragments) into a tile: row = key (this lane's), columns = queries 8g + 4 half + 0..3
__device__ __forceinline__ void tile_put(char* srow, int sw, bf16x8 f0, bf16x8 f1) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 w0 = __builtin_bit_cast(u32x4, f0), w1 = __builtin_bit_cast(u32x4, f1);
    *reinterpret_cast<u32x2*>(srow + 16 * (0 ^ sw)) = u32x2{w0[0], w0[1]};
    *reinterpret_cast<u32x2*>(srow + 16 * (1 ^ sw)) = u32x2{w0[2], w0[3]};
    *reinterpret_cast<u32x2*>(srow + 16 * (2 ^ sw)) = u32x2{w1[0], w1[1]};
    *reinterpret_cast<u32x2*>(srow + 16 * (3 ^ sw)) = u32x2{w1[2], w1[3]};
}

// dma_img_piece with the untracked DMA (the caller counts vmcnt and synchronises)
__device__ __forceinline__ void dma_img_piece_asm(char* img, const __bf16* src, int src_stride, int p, int lane) {
    const int row = 16 * p + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    glds16_asm_at(reinterpret_cast<const f32x4*>(src + (size_t)row * src_stride + chunk * 8), lds_addr_of(img) + p * 1024);
}

template <int ABL>
__global__ __launch_bounds__(512) T2S_X3_KERNEL void attn16_bwd_fused_kernel(const __bf16* __restrict__ q, const __bf16* __restrict__ k,
                                                               const __bf16* __restrict__ v, const __bf16* __restrict__ o_rows,
                                                               const __bf16* __restrict__ do_rows,
                                                               const float* __restrict__ lse, __bf16* __restrict__ dqkv,
                                                               int BH, int reverse) {
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    f32x4* acc = reinterpret_cast<f32x4*>(fsm + FP_ACC);
    float* lraw = reinterpret_cast<float*>(fsm + FP_LRAW);
    int* flag = reinterpret_cast<int*>(fsm + FP_FLAG);
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_it = (BH - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // heads of this workgroup (>= 1)
    const int n_ph = 2 * n_it;
    auto head_of = [&](int it) __attribute__((always_inline)) {
        const int idx = blockIdx.x + it * gridDim.x;
        return reverse ? BH - 1 - idx : idx;
    };
    // phase ph = (head iteration, query half): its Q / dO / O half-images and lse by LDS-DMA, 49 (43) pieces over the waves
    auto prefetch = [&](int ph) __attribute__((always_inline)) {
        const int bh = head_of(ph >> 1), hf = ph & 1, seq = bh / NH, head = bh % NH;
        const int npc = hf ? 14 : 16;
        const __bf16* qsrc = q + ((size_t)bh * NTOK + 256 * hf) * DH;
        const size_t rowoff = ((size_t)seq * NTOK + 256 * hf) * D + head * DH;
        char* Qd = fsm + FP_QS + (ph & 1) * FP_IMG;
        char* Od = fsm + FP_OS + (ph & 1) * FP_IMG;
        for (int i = wave; i <= 3 * npc; i += 8) {
            if (i < npc) dma_img_piece_asm(Qd, qsrc, DH, i, lane);
            else if (i < 2 * npc) dma_img_piece_asm(Od, do_rows + rowoff, D, i - npc, lane);
            else if (i < 3 * npc) dma_img_piece_asm(fsm + FP_OB, o_rows + rowoff, D, i - 2 * npc, lane);
            else if (hf == 0 || lane < 56)
                glds16_asm_at(reinterpret_cast<const f32x4*>(lse + (size_t)bh * NTOK + 256 * hf) + lane, lds_addr_of(fsm + FP_LRAW));
        }
    };
    // -D_i = -sum_d dO.O and -lse of phase ph from the landed images (two threads per query); rows past the head's end are
    // never read
    auto stats = [&](int ph) __attribute__((always_inline)) {
        const int ql = tid >> 1, hh = tid & 1;
        const char* Od = fsm + FP_OS + (ph & 1) * FP_IMG;
        float d_i = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int off = img_off(ql, 2 * hh + c);
            const f32x8 ov = unpack8(*reinterpret_cast<const bf16x8*>(fsm + FP_OB + off)), dv8 = unpack8(*reinterpret_cast<const bf16x8*>(Od + off));
#pragma unroll
            for (int e = 0; e < 8; ++e) d_i += ov[e] * dv8[e];
        }
        d_i += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 1) << 2, __builtin_bit_cast(int, d_i)));
        if (hh == 0) {
            reinterpret_cast<float*>(fsm + FP_DS)[(ph & 1) * 256 + ql] = -d_i;
            reinterpret_cast<float*>(fsm + FP_LS)[(ph & 1) * 256 + ql] = -lraw[ql];
        }
    };
    char* scr0 = fsm + FP_TILE + (2 * wave) * 2048;
    char* scr1 = scr0 + 2048;
    const int sw = (j >> 2) & 3;
    char* srow0 = scr0 + j * 64 + 8 * half;           // this lane's key row of the dS tiles
    char* srow1 = scr1 + j * 64 + 8 * half;
    // Wave 7 has one key block: its second chain repeats block 14 with a zero dQ operand and its dK / dV are dropped.
    const bool second = wave < 7;
    const int kb0 = 2 * wave, kb1 = second ? 2 * wave + 1 : NKB - 1;
    bf16x8 kf0[2], vf0[2], kc0[2], kf1[2], vf1[2], kc1[2];
    // a head's key blocks: B operands K^T / V^T straight from HBM; the A operand of the dQ product (K with the FEATURE on the
    // lane, k = keys in the column-read order) through the tiles
    // a head's key blocks: the B operands K^T / V^T straight from HBM into registers; the A operand of the dQ product (K with
    // the FEATURE on the lane, k = keys in the column-read order) through the wave's tiles, filled by LDS-DMA
    auto kv_tiles = [&](int bh) __attribute__((always_inline)) {
        const __bf16* kg = k + (size_t)bh * NTOK * DH;
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            dma_img_piece_asm(scr0, kg + (size_t)kb0 * 32 * DH, DH, pc, lane);
            dma_img_piece_asm(scr1, kg + (size_t)kb1 * 32 * DH, DH, pc, lane);
        }
    };
    auto kv_issue = [&](int bh) __attribute__((always_inline)) {
        const __bf16* kg = k + (size_t)bh * NTOK * DH;
        const __bf16* vg = v + (size_t)bh * NTOK * DH;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            kf0[t] = *reinterpret_cast<const bf16x8*>(kg + (size_t)(kb0 * 32 + j) * DH + 16 * t + 8 * half);
            vf0[t] = *reinterpret_cast<const bf16x8*>(vg + (size_t)(kb0 * 32 + j) * DH + 16 * t + 8 * half);
            kf1[t] = *reinterpret_cast<const bf16x8*>(kg + (size_t)(kb1 * 32 + j) * DH + 16 * t + 8 * half);
            vf1[t] = *reinterpret_cast<const bf16x8*>(vg + (size_t)(kb1 * 32 + j) * DH + 16 * t + 8 * half);
        }
    };
    constexpr int KV_LOADS = 8;                       // global loads kv_issue puts behind the DMA pieces in the vmcnt order
    auto kv_finish = [&]() __attribute__((always_inline)) {      // after the tile pieces have landed
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            kc0[t] = col_frag(scr0, 0, lane, t);
            kc1[t] = col_frag(scr1, 0, lane, t);
            if (!second) {
#pragma unroll
                for (int e = 0; e < 8; ++e) kc1[t][e] = (__bf16)0.f;
            }
        }
    };
    // the next head's K and V pulled into L2 while this head computes (LDS-DMA into a dump: no registers)
    auto warm_kv = [&](int bh) __attribute__((always_inline)) {
        const f32x4* kg = reinterpret_cast<const f32x4*>(k + (size_t)bh * NTOK * DH);
        const f32x4* vg = reinterpret_cast<const f32x4*>(v + (size_t)bh * NTOK * DH);
        for (int i = wave; i < 60; i += 8)
            glds16_asm_at((i < 30 ? kg + i * 64 : vg + (i - 30) * 64) + lane, lds_addr_of(fsm + FP_DUMP));
    };
    prefetch(0);
    kv_tiles(head_of(0));
    kv_issue(head_of(0));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    kv_finish();
    wg_sync();
    stats(0);
    wg_sync();
    f32x16 dk0, dv0, dk1, dv1;
#pragma unroll
    for (int r = 0; r < 16; ++r) dk0[r] = dv0[r] = dk1[r] = dv1[r] = 0.f;
    for (int ph = 0; ph < n_ph; ++ph) {
        const int hf = ph & 1, nblk = hf ? 7 : 8;
        const char* Qs = fsm + FP_QS + hf * FP_IMG;
        const char* Os = fsm + FP_OS + hf * FP_IMG;
        const float* Ls = reinterpret_cast<const float*>(fsm + FP_LS) + hf * 256;
        const float* Ds = reinterpret_cast<const float*>(fsm + FP_DS) + hf * 256;
        const bool next_head = hf && ph + 1 < n_ph && !(ABL & 4);
        if (!(ABL & 8) && hf && ph + 1 < n_ph) warm_kv(head_of((ph + 1) >> 1));
        if (!(ABL & 2) && ph + 1 < n_ph) prefetch(ph + 1);          // the other buffers: every wave is past its reads of them (barrier below)
        for (int s = 0; s < 8; ++s) {
            const int qb = (wave + s) & 7;
            if (qb >= nblk || (ABL & 1)) {
                if (s == 7 && next_head) kv_tiles(head_of((ph + 1) >> 1));
                continue;
            }
            // the accumulators start at -lse / -D of the block's queries (register = query); read once per chain: a shared
            // copy as an untied C operand cost 32 registers and spilled
            auto stat16 = [&](const float* src) __attribute__((always_inline)) {
                f32x16 r16;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(src + qb * 32 + 8 * g + 4 * half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) r16[4 * g + e] = t4[e];
                }
                return r16;
            };
            const bf16x8 qr0 = row_frag(Qs, qb * 32, lane, 0), qr1 = row_frag(Qs, qb * 32, lane, 1);
            const bf16x8 or0 = row_frag(Os, qb * 32, lane, 0), or1 = row_frag(Os, qb * 32, lane, 1);
            f32x16 sc0 = stat16(Ls), dp0 = stat16(Ds);
            asm volatile("" ::: "memory");
            f32x16 sc1 = stat16(Ls), dp1 = stat16(Ds);
            sc0 = mfma16(qr0, kf0[0], sc0);               // S[query][key] - lse
            dp0 = mfma16(or0, vf0[0], dp0);               // dP[query][key] - D
            sc1 = mfma16(qr0, kf1[0], sc1);
            dp1 = mfma16(or0, vf1[0], dp1);
            sc0 = mfma16(qr1, kf0[1], sc0);
            dp0 = mfma16(or1, vf0[1], dp0);
            sc1 = mfma16(qr1, kf1[1], sc1);
            dp1 = mfma16(or1, vf1[1], dp1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(sc0[r]);
                dp0[r] = p * dp0[r];           // dS[query][key]
                sc0[r] = p;                    // P[query][key]
            }
            const bf16x8 ds00 = acc_frag(dp0, 0), ds01 = acc_frag(dp0, 1);
            tile_put(srow0, sw, ds00, ds01);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(sc1[r]);
                dp1[r] = p * dp1[r];
                sc1[r] = p;
            }
            const bf16x8 ds10 = acc_frag(dp1, 0), ds11 = acc_frag(dp1, 1);
            tile_put(srow1, sw, ds10, ds11);
            {
                const bf16x8 oc0 = col_frag(Os, qb * 32, lane, 0), oc1 = col_frag(Os, qb * 32, lane, 1);
                const bf16x8 qc0 = col_frag(Qs, qb * 32, lane, 0), qc1 = col_frag(Qs, qb * 32, lane, 1);
                dv0 = mfma16(oc0, acc_frag(sc0, 0), dv0);      // dV^T += dO^T P
                dk0 = mfma16(qc0, ds00, dk0);                  // dK^T += Q^T dS
                dv1 = mfma16(oc0, acc_frag(sc1, 0), dv1);
                dk1 = mfma16(qc0, ds10, dk1);
                dv0 = mfma16(oc1, acc_frag(sc0, 1), dv0);
                dk0 = mfma16(qc1, ds01, dk0);
                dv1 = mfma16(oc1, acc_frag(sc1, 1), dv1);
                dk1 = mfma16(qc1, ds11, dk1);
            }
            asm volatile("" ::: "memory");     // the tile reads below stay behind the tile writes above (one wave: LDS is in order)
            f32x16 dq = mfma16_zero(kc0[0], col_frag(scr0, 0, lane, 0));               // dQ^T[feature][query] of the two key blocks
            dq = mfma16(kc1[0], col_frag(scr1, 0, lane, 0), dq);
            dq = mfma16(kc0[1], col_frag(scr0, 0, lane, 1), dq);
            dq = mfma16(kc1[1], col_frag(scr1, 0, lane, 1), dq);
            if (s == 7 && next_head) {         // the tiles are free (their last column reads have been consumed): next head's K
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                kv_tiles(head_of((ph + 1) >> 1));
            }
            // ordered accumulation (exclusive between the two flag operations: plain read-add-write of 16-byte pieces)
            f32x4* a = acc + (size_t)qb * 256 + lane;
            if (s != 0) {
                while (__hip_atomic_load(flag + qb, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != s) __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 old = a[64 * g];
#pragma unroll
                    for (int e = 0; e < 4; ++e) dq[4 * g + e] += old[e];
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) a[64 * g] = f32x4{dq[4 * g], dq[4 * g + 1], dq[4 * g + 2], dq[4 * g + 3]};
            __hip_atomic_store(flag + qb, s + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // phase boundary: the next phase's images have landed (this wave's pieces; the barrier covers the others'), every
        // block of this phase has its 8 contributions.  At a head's end the next head's K / V loads go out first (their
        // registers are dead) and only the DMA pieces in front of them are waited for.
        if (next_head) {
            kv_issue(head_of((ph + 1) >> 1));
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KV_LOADS) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        wg_sync();
        const int bh = head_of(ph >> 1), seq = bh / NH, head = bh % NH;
        if (wave < nblk) {             // dQ of local block `wave`: query rows 256 hf + 32 wave + j
            f32x16 dq;
            const f32x4* a = acc + (size_t)wave * 256 + lane;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 t4 = a[64 * g];
#pragma unroll
                for (int e = 0; e < 4; ++e) dq[4 * g + e] = t4[e];
            }
            store_row32(dqkv + ((size_t)seq * NTOK + 256 * hf + 32 * wave + j) * (3 * D) + head * DH, dq, SCALE, half);
        }
        if (hf) {                      // the head is complete
            __bf16* dst0 = dqkv + ((size_t)seq * NTOK + kb0 * 32 + j) * (3 * D) + head * DH;
            store_row32(dst0 + D, dk0, LN2, half);
            store_row32(dst0 + 2 * D, dv0, 1.0f, half);
            if (second) {
                __bf16* dst1 = dst0 + (size_t)32 * (3 * D);
                store_row32(dst1 + D, dk1, LN2, half);
                store_row32(dst1 + 2 * D, dv1, 1.0f, half);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) dk0[r] = dv0[r] = dk1[r] = dv1[r] = 0.f;
            if (next_head) kv_finish();
        }
        if (ph + 1 < n_ph) stats(ph + 1);
        wg_sync();
    }
}

int attn16_train_fwd(const __bf16* q, const __bf16* k, const __bf16* v, __bf16* o_rows, float* lse, int BH, hipStream_t st) {
    attn16_fwd_kernel<<<BH, 512, 0, st>>>(q, k, v, o_rows, lse, next_tile_dir());
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

int attn16_bwd(const __bf16* q, const __bf16* k, const __bf16* v, const __bf16* o_rows, const __bf16* do_rows,
               const float* lse, float* dsum, __bf16* dqkv_rows, int BH, hipStream_t st) {
    // T2S_ATTN_BWD_FUSED=1: the one-pass kernel (read per call, so that a test can compare the two in one process)
    const char* fused = getenv("T2S_ATTN_BWD_FUSED");
    if (fused && atoi(fused)) {
        static const int n_cu = [] {
            int dev = 0, n = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
            return n;
        }();
        const int abl = getenv("T2S_FB_ABL") ? atoi(getenv("T2S_FB_ABL")) : 0;
        const int dir = next_tile_dir();
#define FB_LAUNCH(A)                                                                                                      \
    case A:                                                                                                               \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn16_bwd_fused_kernel<A>),                                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, FP_LDS) != hipSuccess) return T2S_E_HIP;     \
        attn16_bwd_fused_kernel<A><<<BH < n_cu ? BH : n_cu, 512, FP_LDS, st>>>(q, k, v, o_rows, do_rows, lse, dqkv_rows, BH, dir); \
        break;
        switch (abl) {
            FB_LAUNCH(0) FB_LAUNCH(1) FB_LAUNCH(2) FB_LAUNCH(4) FB_LAUNCH(6) FB_LAUNCH(8) FB_LAUNCH(10) FB_LAUNCH(12) FB_LAUNCH(14)
            default: return T2S_E_INVALID;
        }
        T2S_LAUNCH_CHECK();
        (void)next_tile_dir();   // keep the direction pattern of the launches that follow
        return T2S_OK;
    }
    attn16_bwd_dq_kernel<<<BH, 512, 0, st>>>(q, k, v, o_rows, do_rows, lse, dsum, dqkv_rows, next_tile_dir());   // also writes D_i -> dsum
    T2S_LAUNCH_CHECK();
    attn16_bwd_dkv_kernel<<<BH, 512, 0, st>>>(q, k, v, do_rows, lse, dsum, dqkv_rows, next_tile_dir());
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

}  // namespace t2s

// ------------------------------------------------------------------ C ABI: stand-alone bf16 attention forward
namespace {
__global__ void f32_to_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, size_t n4, float scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) reinterpret_cast<t2s::bf16x4*>(dst)[i] = t2s::pack4(reinterpret_cast<const t2s::f32x4*>(src)[i] * scale);
}
__global__ void bf16_to_f32_kernel(const __bf16* __restrict__ src, float* __restrict__ dst, size_t n4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) reinterpret_cast<t2s::f32x4*>(dst)[i] = t2s::unpack4(reinterpret_cast<const t2s::bf16x4*>(src)[i]);
}
}  // namespace

extern "C" int t2s_attn_fwd_bf16(const float* q, const float* k, const float* v, float* o_rows, float* lse, int n_seq,
                                 void* stream) {
    using namespace t2s;
    T2S_REQUIRE(q && k && v && o_rows && lse && n_seq > 0, "t2s_attn_fwd_bf16: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)n_seq * NH * NTOK * DH;      // elements of each of q, k, v, o
    __bf16* buf = nullptr;
    T2S_HIP_CHECK(hipMalloc(&buf, 4 * n * sizeof(__bf16)));
    const unsigned blocks = (unsigned)((n / 4 + 255) / 256);
    f32_to_bf16_kernel<<<blocks, 256, 0, st>>>(q, buf, n / 4, ATT_QS);   // the kernels take q pre-scaled (as the qkv GEMM stores it)
    f32_to_bf16_kernel<<<blocks, 256, 0, st>>>(k, buf + n, n / 4, 1.0f);
    f32_to_bf16_kernel<<<blocks, 256, 0, st>>>(v, buf + 2 * n, n / 4, 1.0f);
    int rc = attn16_train_fwd(buf, buf + n, buf + 2 * n, buf + 3 * n, lse, n_seq * NH, st);
    if (rc == T2S_OK) {
        bf16_to_f32_kernel<<<blocks, 256, 0, st>>>(buf + 3 * n, o_rows, n / 4);
        if (hipGetLastError() != hipSuccess) rc = T2S_E_HIP;
    }
    (void)hipStreamSynchronize(st);
    (void)hipFree(buf);
    return rc;
}

