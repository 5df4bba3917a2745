// The DiT's row-local chain on 16-TOKEN tiles (v_mfma_f32_16x16x4_f32) for SMALL launches.
//
// t2s_rows.h carries 32 tokens per wave.  A strong-scaling shard of 32 series (a 256-series job on 8 GPUs: 64 sequences
// in a CFG pass) is 960 such tiles for the chip's 1024 SIMDs: one wave per SIMD, nothing to overlap a wave's prologue
// (its rows from HBM), LayerNorm / GELU phases, launch and tail with.  Here a wave carries 16 tokens, so the same launch
// is 1920 waves -- two per SIMD, each with half the work -- and the residual stream (32 registers) never leaves the
// register file (the 32-token kernel parks it in HBM around the MLP loop).
//
// Layout ("16-layout").  16x16x4: lane l = 16 g + i supplies A[i][k = g] and B[k = g][j = i]; result register r of lane
// (g, j) is D[4 g + r][j].  A token's 128 features live on its four lanes (g = 0..3): lane (g, tok) holds, for every
// 16-feature block mt and r = 0..3, feature 16 mt + pi(g, r) with
//     pi(g, r) = 8 (r >> 1) + 2 (r & 1) + 4 (g & 1) + (g >> 1).
// Transposed products Y^T[n][tok] = W[n][:] . act[tok][:] with the weight rows of a 16-output block permuted the same
// way (A row 4 g + r <-> output 16 mo + pi(g, r)) hand the result back in this layout, i.e. as the B operand of the next
// product: MFMA step (mt, r) contracts the four features {16 mt + pi(g, r) : g = 0..3}.
// pi is chosen so that the k ORDER of every accumulation equals the 32-token kernel's (its MFMA (G, e) contracts the pair
// {8 G + e, 8 G + 4 + e}; two consecutive pairs are one 16x16x4 step: (0,4,1,5) (2,6,3,7) (8,12,9,13) (10,14,11,15) per
// block), and every elementwise formula / LayerNorm sum is the shared one of t2s_rows.h: a token gets THE SAME BITS from
// either kernel (tests/test_hip_parity.py: the 32-series shards equal the rows of the 256-series batch bitwise).
//
// HBM tensors keep the 32-token fragment-major layout (t2s_common.h: frag_index), so this kernel and the 32-token one,
// and the attention kernel between them, are interchangeable per launch.  A lane's features {q, q+2} of two float4
// quads are completed with its partner lane (l ^ 32, v_permlane32_swap) for 16-byte stores; loads fetch both quads.
//
// Weights: the same 16 KiB chunks (32 outputs x K = 128, or the 8 x 2 blocks one fc1 chunk feeds into fc2) in 16-layout
// fragment order, through the same 3-slot LDS ring / counted-vmcnt DMA as t2s_rows.h.  Two output blocks (or two fc2
// accumulators) are interleaved so that dependent MFMAs sit 64 cycles apart (16x16x4: 32-cycle issue, 40-cycle dependent
// latency).  All three instances exist: <qkv>, <proj + MLP + qkv>, <proj + MLP + fused final layer>.
#pragma once
#include "t2s_rows.h"

namespace t2s {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__host__ __device__ inline int pi16(int g, int r) { return 8 * (r >> 1) + 2 * (r & 1) + 4 * (g & 1) + (g >> 1); }
// position 4 g + r of offset o = pi16(g, r) inside its 16-feature block
__host__ __device__ inline int pos16(int o) {
    const int r = ((o >> 3) << 1) | ((o >> 1) & 1), g = ((o >> 2) & 1) | ((o & 1) << 1);
    return 4 * g + r;
}
// W (N, K = 128) -> chunk c = n / 32: fragments [mo_l = (n / 16) & 1][mt = k / 16], lane = 16 g_k + pos16(n % 16), element r_k
__host__ __device__ inline size_t packed16_index(int n, int k) {
    const int pk = pos16(k & 15);
    const int lane = (pk >> 2) * 16 + pos16(n & 15);
    return (((((size_t)(n >> 5) * 2 + ((n >> 4) & 1)) * 8 + (k >> 4)) * 64) + lane) * 4 + (pk & 3);
}
// fc2 W (128, K = 256) -> chunk c = k / 32 (the hidden units one fc1 chunk produces): fragments [mo2 = n / 16][j = (k / 16) & 1]
__host__ __device__ inline size_t packed16_fc2_index(int n, int k) {
    const int pk = pos16(k & 15);
    const int lane = (pk >> 2) * 16 + pos16(n & 15);
    return (((((size_t)(k >> 5) * 8 + (n >> 4)) * 2 + ((k >> 4) & 1)) * 64) + lane) * 4 + (pk & 3);
}

// (lo | hi) exchange with the partner lane l ^ 32: returns {a.lo | b.lo} and {a.hi | b.hi}
__device__ __forceinline__ void swap32(float a, float b, float& o0, float& o1) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    o0 = __uint_as_float(r[0]);
    o1 = __uint_as_float(r[1]);
}

// NB 16-feature blocks of this lane's token from a fragment-major tile (C = 16 NB columns): both float4 quads of a block
// are fetched, the lane keeps elements {q, q + 2}
template <int NB>
__device__ __forceinline__ void load_rows16(const f32x4* __restrict__ tile_base, int h, int q, int row, f32x4 (&x)[NB]) {
#pragma unroll
    for (int mt = 0; mt < NB; ++mt) {
        const f32x4 qa = tile_base[(2 * mt) * 64 + 32 * h + row];
        const f32x4 qb = tile_base[(2 * mt + 1) * 64 + 32 * h + row];
        x[mt] = q ? f32x4{qa[1], qa[3], qb[1], qb[3]} : f32x4{qa[0], qa[2], qb[0], qb[2]};
    }
}
// the same in two halves -- the 16 loads now, the selection when the values are first needed -- so that the loads can stay in
// flight across the kernel's first barrier (ROWS_SYNC_BUT16)
template <int NB>
__device__ __forceinline__ void load_rows16_raw(const f32x4* __restrict__ tile_base, int h, int row, f32x4 (&raw)[2 * NB]) {
#pragma unroll
    for (int mt = 0; mt < NB; ++mt) {
        raw[2 * mt] = tile_base[(2 * mt) * 64 + 32 * h + row];
        raw[2 * mt + 1] = tile_base[(2 * mt + 1) * 64 + 32 * h + row];
    }
}
template <int NB>
__device__ __forceinline__ void select_rows16(const f32x4 (&raw)[2 * NB], int q, f32x4 (&x)[NB]) {
#pragma unroll
    for (int mt = 0; mt < NB; ++mt) {
        const f32x4 qa = raw[2 * mt], qb = raw[2 * mt + 1];
        x[mt] = q ? f32x4{qa[1], qa[3], qb[1], qb[3]} : f32x4{qa[0], qa[2], qb[0], qb[2]};
    }
}
// the reverse: lanes q = 0 end up with the whole second quad of a block, lanes q = 1 with the first, one 16-byte store each
template <int NB>
__device__ __forceinline__ void store_rows16(f32x4* __restrict__ tile_base, int h, int q, int row, const f32x4 (&x)[NB]) {
#pragma unroll
    for (int mt = 0; mt < NB; ++mt) {
        float o0, o1, o2, o3;
        swap32(x[mt][2], x[mt][0], o0, o1);
        swap32(x[mt][3], x[mt][1], o2, o3);
        tile_base[(2 * mt + (q ? 0 : 1)) * 64 + 32 * h + row] = f32x4{o0, o1, o2, o3};
    }
}

// per-feature constants are staged in LDS in 16-layout order: element 16 mt + 4 g + r holds feature 16 mt + pi16(g, r)
__device__ __forceinline__ f32x4 ldc16(const float* __restrict__ vec16, int mt, int g) {
    return *reinterpret_cast<const f32x4*>(vec16 + 16 * mt + 4 * g);
}

// sum over the four lanes of a token: (P0 + P2) + (P1 + P3), the tree t2s_rows.h's ln_modulate follows
__device__ __forceinline__ float token_sum16(float p) {
    float a, b;
    swap32(p, p, a, b);             // {lo | lo}, {hi | hi}: a + b = p + p(l ^ 32) in every lane
    const float t = a + b;
    return t + __shfl_xor(t, 16, 64);
}

__device__ __forceinline__ void ln_mean_rstd16(const f32x4 (&x)[8], float eps, float& mean, float& rstd) {
    float p = 0.f;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) p += x[mt][r];
    mean = token_sum16(p) * (1.0f / 128.0f);
    float qq = 0.f;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = x[mt][r] - mean;
            qq = __builtin_fmaf(d, d, qq);
        }
    rstd = ln_rstd(token_sum16(qq), eps);
}

__device__ __forceinline__ void ln_modulate16(const f32x4 (&x)[8], f32x4 (&y)[8], const float* __restrict__ shift16,
                                              const float* __restrict__ scale16, int g, float eps) {
    float mean, rstd;
    ln_mean_rstd16(x, eps, mean, rstd);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const f32x4 sc = ldc16(scale16, mt, g);
        const f32x4 sh = ldc16(shift16, mt, g);
#pragma unroll
        for (int r = 0; r < 4; ++r) y[mt][r] = ln_y(x[mt][r], mean, rstd, sc[r], sh[r]);
    }
}

// the lane's four elements {q, q + 2} of two float4 quads of a natural-order vector: features 16 mt + 4 h + q + {0, 2, 8, 10}
__device__ __forceinline__ f32x4 gather16(const float* __restrict__ vec, int mt, int h, int q) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(vec + 16 * mt + 4 * h);
    const f32x4 c = *reinterpret_cast<const f32x4*>(vec + 16 * mt + 8 + 4 * h);
    return q ? f32x4{a[1], a[3], c[1], c[3]} : f32x4{a[0], a[2], c[0], c[2]};
}

// One chunk = two 16-output blocks x K = 128 (fragments [mo_l][mt] at wb[(8 mo_l + mt) * 64]) against the activation b
// (16-layout); the two accumulators alternate.  SWAP exchanges the MFMA operands: rows = tokens, columns = outputs (V^T).
template <bool SWAP>
__device__ __forceinline__ void kchunk16(const f32x4* __restrict__ wb, const f32x4 (&b)[8], f32x4& acc0, f32x4& acc1) {
    f32x4 w0 = wb[0], w1 = wb[8 * 64];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        f32x4 n0 = w0, n1 = w1;
        if (mt + 1 < 8) {
            n0 = wb[(mt + 1) * 64];
            n1 = wb[(8 + mt + 1) * 64];
        }
        T2S_SCHED_FENCE();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc0 = SWAP ? mfma16(b[mt][r], w0[r], acc0) : mfma16(w0[r], b[mt][r], acc0);
            acc1 = SWAP ? mfma16(b[mt][r], w1[r], acc1) : mfma16(w1[r], b[mt][r], acc1);
        }
        T2S_SCHED_FENCE();
        w0 = n0;
        w1 = n1;
    }
}

template <bool DO_MLP, bool DO_QKV>
__global__ __launch_bounds__(256, 2) void dit_rows16_kernel(const RowArgs a) {
    extern __shared__ __attribute__((aligned(16))) f32x4 wring[];  // [ROWS_SLOTS][1024]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, tok = lane & 15, h = g & 1, q = g >> 1;
    const int n16 = a.M >> 4;
    int t16 = blockIdx.x * 4 + wave;              // 16-token half tile of this wave
    const bool active = t16 < n16;                // tail waves compute on a clamped tile, store nothing
    if (!active) t16 = n16 - 1;
    const int tile = t16 >> 1, row = 16 * (t16 & 1) + tok;
    const int seq = (tile * 32) / NTOK;
    const int tile_in_seq = tile - seq * (NTOK / 32);
    const float* __restrict__ modrow = mod_row_of(a, seq);

    constexpr int N_CHUNKS = (DO_MLP ? 20 : 0) + (DO_QKV ? 12 : 0);
    auto chunk_src = [&](int ci) -> const f32x4* {
        if constexpr (DO_MLP) {
            if (ci < 4) return a.Wp + (size_t)ci * ROWS_CHUNK_F4;
            if (ci < 20) {
                const int j = ci - 4;
                return ((j & 1) ? a.W2c : a.W1) + (size_t)(j >> 1) * ROWS_CHUNK_F4;
            }
            ci -= 20;
        }
        return a.Wq + (size_t)ci * ROWS_CHUNK_F4;
    };
    auto fill = [&](int ci) {
        const f32x4* src = chunk_src(ci) + lane;
        f32x4* dst = wring + (ci % ROWS_SLOTS) * ROWS_CHUNK_F4;
#pragma unroll
        for (int p = 0; p < 4; ++p) glds16_asm(src + (wave + 4 * p) * 64, dst + (wave + 4 * p) * 64);
    };
    auto prefetch = [&](int ci) {
        if (ci + ROWS_DIST < N_CHUNKS) fill(ci + ROWS_DIST);
    };
    // end of chunk ci: chunk ci + 1 landed for every wave (counted wait: the DIST - 1 younger chunks of 4 pieces each
    // and, in the qkv phase, the 2 stores of each of the last DIST tiles stay in flight), then the workgroup barrier
    auto chunk_done = [&](int ci, bool own_stores) {
        if (ci + ROWS_DIST < N_CHUNKS) {
            if (own_stores)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ROWS_DIST - 1) * 4 + ROWS_DIST * 2) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ROWS_DIST - 1) * 4) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    };

#pragma unroll
    for (int c0 = 0; c0 < ROWS_DIST; ++c0) fill(c0);

    // ---- per-feature constants in LDS, 16-layout order (element (f & ~15) | pos16(f & 15) holds feature f) ----
    float* cb = reinterpret_cast<float*>(wring + ROWS_SLOTS * ROWS_CHUNK_F4);
    float* cm = cb + ROWS_CB_FLOATS + wave * ROWS_CMF;
    // Order of the prologue's vector-memory operations (round 5, as t2s_rows.h): the weight DMAs, the attention output (first:
    // its round trip runs under the constants'), the constants as ONE batch of loads then one batch of ds_writes, and LAST the
    // residual stream, which the first barrier leaves in flight (ROWS_SYNC_BUT16).  At the launch sizes this kernel serves a
    // launch is a single round of waves: the prologue's serial round trips are not hidden by anything.
    f32x4 araw[DO_MLP ? 16 : 1];
    if constexpr (DO_MLP) {
        load_rows16_raw<8>(reinterpret_cast<const f32x4*>(a.ao) + (size_t)tile * 16 * 64, h, row, araw);
        __builtin_amdgcn_sched_barrier(0);
    }
    {
        const int t0 = threadIdx.x, t1 = t0 + 256;
        float c0 = 0.f, c1 = 0.f, q0 = 0.f, q1 = 0.f;
        f32x4 m1 = {}, m2 = {}, mq = {};
        if constexpr (DO_MLP) {
            c0 = t0 < 128 ? a.bp[t0] : a.b1[t0 - 128];
            c1 = t1 < 384 ? a.b1[t1 - 128] : a.b2[t1 - 384];
            const float* src = modrow + a.blk * MODW;
            m1 = *reinterpret_cast<const f32x4*>(src + (64 + lane) * 4);
            m2 = *reinterpret_cast<const f32x4*>(src + (128 + lane) * 4);
        }
        if constexpr (DO_QKV) {
            q0 = a.bq[t0];
            if (t1 < 384) q1 = a.bq[t1];
            mq = *reinterpret_cast<const f32x4*>(modrow + a.qkv_blk * MODW + lane * 4);   // shift_msa | scale_msa
        }
        __builtin_amdgcn_sched_barrier(0);
        auto put4 = [&](float* base, int f0, const f32x4& v) {
#pragma unroll
            for (int e = 0; e < 4; ++e) base[((f0 + e) & ~15) | pos16((f0 + e) & 15)] = v[e];
        };
        if constexpr (DO_MLP) {
            cb[(t0 & ~15) | pos16(t0 & 15)] = c0;
            cb[(t1 & ~15) | pos16(t1 & 15)] = c1;
            put4(cm, (64 + lane) * 4, m1);
            put4(cm, (128 + lane) * 4, m2);
        }
        if constexpr (DO_QKV) {
            cb[512 + ((t0 & ~15) | pos16(t0 & 15))] = q0;
            if (t1 < 384) cb[512 + ((t1 & ~15) | pos16(t1 & 15))] = q1;
            put4(cm, lane * 4, mq);
        }
    }
    const float* c_bp = cb;
    const float* c_b1 = cb + 128;
    const float* c_b2 = cb + 384;
    const float* c_bq = cb + 512;

    // residual stream of this lane's token
    f32x4 x[8];
    const int tile_src = tile - (seq - seq % a.in_seqs) * (NTOK / 32);   // same tile of sequence seq % in_seqs
    bool generated = false;
    if constexpr (!DO_MLP) {
        if (a.p_lat != nullptr) {
            // patchify in the prologue (see t2s_rows.h): patch_emb weight in the unused proj / MLP bias slots, bias in the
            // wave's unused MLP adaLN slots, natural feature order
            for (int i = threadIdx.x; i < 512; i += 256) cb[i] = a.p_pw[i];
            *reinterpret_cast<f32x4*>(cm + 256 + (lane & 31) * 4) = *reinterpret_cast<const f32x4*>(a.p_pb + (lane & 31) * 4);
            __syncthreads();
            const int n = tile_in_seq * 32 + row;
            float cv[4];
            patch_conv(a, seq, n, cv);
            const float* posrow = a.p_pos + (size_t)n * D;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                // features 16 mt + pi16(g, r) = 16 mt + 4 h + q + {0, 2, 8, 10}
                const f32x4 pa = *reinterpret_cast<const f32x4*>(posrow + 16 * mt + 4 * h);
                const f32x4 pc = *reinterpret_cast<const f32x4*>(posrow + 16 * mt + 8 + 4 * h);
                const f32x4 ba = *reinterpret_cast<const f32x4*>(cm + 256 + 16 * mt + 4 * h);
                const f32x4 bc = *reinterpret_cast<const f32x4*>(cm + 256 + 16 * mt + 8 + 4 * h);
                const int d0 = 16 * mt + 4 * h + q;
                x[mt][0] = patch_feature(cv, *reinterpret_cast<const f32x4*>(cb + (d0 + 0) * 4), q ? ba[1] : ba[0], q ? pa[1] : pa[0]);
                x[mt][1] = patch_feature(cv, *reinterpret_cast<const f32x4*>(cb + (d0 + 2) * 4), q ? ba[3] : ba[2], q ? pa[3] : pa[2]);
                x[mt][2] = patch_feature(cv, *reinterpret_cast<const f32x4*>(cb + (d0 + 8) * 4), q ? bc[1] : bc[0], q ? pc[1] : pc[0]);
                x[mt][3] = patch_feature(cv, *reinterpret_cast<const f32x4*>(cb + (d0 + 10) * 4), q ? bc[3] : bc[2], q ? pc[3] : pc[2]);
            }
            // every lane takes part in the stores' lane exchange; only tokens of the distinct sequences are written
            if (active && seq < a.in_seqs)
                store_rows16<8>(const_cast<f32x4*>(reinterpret_cast<const f32x4*>(a.x_in)) + (size_t)tile_src * 16 * 64, h, q, row, x);
            generated = true;
        }
    }
    if (!DO_MLP && !generated) load_rows16<8>(reinterpret_cast<const f32x4*>(a.x_in) + (size_t)tile_src * 16 * 64, h, q, row, x);
    int ci = 0;

    if constexpr (DO_MLP) {
        const float* mb = cm;   // [shift_msa, scale_msa of the qkv block | gate_msa, shift_mlp, scale_mlp, gate_mlp of a.blk]
        // ---------------- x += gate_msa * (proj(ao) + b) ----------------
        {
            f32x4 bop[8], xraw[16];
            __builtin_amdgcn_sched_barrier(0);
            load_rows16_raw<8>(reinterpret_cast<const f32x4*>(a.x_in) + (size_t)tile_src * 16 * 64, h, row, xraw);
            __builtin_amdgcn_sched_barrier(0);
            select_rows16<8>(araw, q, bop);
            ROWS_SYNC_BUT16();  // chunk 0 landed, ao and the constants here; the residual stream (16 loads) may still be on its way
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                prefetch(ci);
                const f32x4* wb = wring + (ci % ROWS_SLOTS) * ROWS_CHUNK_F4 + lane;
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                kchunk16<false>(wb, bop, acc0, acc1);
                if (c == 0) select_rows16<8>(xraw, q, x);     // first use of the residual stream: it has landed behind the MFMAs
#pragma unroll
                for (int ml = 0; ml < 2; ++ml) {
                    const f32x4 bias = ldc16(c_bp, 2 * c + ml, g);
                    const f32x4 gate = ldc16(mb + 2 * D, 2 * c + ml, g);
                    const f32x4& acc = ml ? acc1 : acc0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[2 * c + ml][r] = res_gate(x[2 * c + ml][r], gate[r], acc[r], bias[r]);
                }
                chunk_done(ci, false);
                ++ci;
            }
        }
        // ---------------- x += gate_mlp * (fc2(gelu(fc1(mod(LN(x))))) + b2) ----------------
        {
            f32x4 xm[8];
            ln_modulate16(x, xm, mb + 3 * D, mb + 4 * D, g, 1e-6f);
            f32x4 acc[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int c = 0; c < 8; ++c) {  // 32 hidden units per chunk; ci = 4 + 2c here
                prefetch(ci);
                f32x4 hT[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                {
                    const f32x4* wb = wring + (ci % ROWS_SLOTS) * ROWS_CHUNK_F4 + lane;
                    kchunk16<false>(wb, xm, hT[0], hT[1]);
#pragma unroll
                    for (int ml = 0; ml < 2; ++ml) {
                        const f32x4 bias = *reinterpret_cast<const f32x4*>(c_b1 + 32 * c + 16 * ml + 4 * g);
#pragma unroll
                        for (int r = 0; r < 4; ++r) hT[ml][r] = gelu_tanh_f(hT[ml][r] + bias[r]);
                    }
                }
                chunk_done(ci, false);
                ++ci;
                prefetch(ci);
                {   // fc2 partial over the 32 hidden units of this chunk; fragments [mo2][j] at wb[(2 mo2 + j) * 64]
                    const f32x4* wb = wring + (ci % ROWS_SLOTS) * ROWS_CHUNK_F4 + lane;
                    f32x4 wa = wb[0], wc = wb[2 * 64];
#pragma unroll
                    for (int s = 0; s < 8; ++s) {      // s = 2 mp + j: output blocks 2 mp, 2 mp + 1, hidden block j
                        const int mp = s >> 1, j = s & 1;
                        f32x4 na = wa, nc = wc;
                        if (s + 1 < 8) {
                            const int mp1 = (s + 1) >> 1, j1 = (s + 1) & 1;
                            na = wb[(2 * (2 * mp1) + j1) * 64];
                            nc = wb[(2 * (2 * mp1 + 1) + j1) * 64];
                        }
                        T2S_SCHED_FENCE();
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            acc[2 * mp] = mfma16(wa[r], hT[j][r], acc[2 * mp]);
                            acc[2 * mp + 1] = mfma16(wc[r], hT[j][r], acc[2 * mp + 1]);
                        }
                        T2S_SCHED_FENCE();
                        wa = na;
                        wc = nc;
                    }
                }
                chunk_done(ci, false);
                ++ci;
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const f32x4 bias = ldc16(c_b2, m, g);
                const f32x4 gate = ldc16(mb + 5 * D, m, g);
#pragma unroll
                for (int r = 0; r < 4; ++r) x[m][r] = res_gate(x[m][r], gate[r], acc[m][r], bias[r]);
            }
            if (active && (DO_QKV || a.out0 == nullptr || a.keep_x))   // final residual stream of this block
                store_rows16<8>(reinterpret_cast<f32x4*>(a.x) + (size_t)tile * 16 * 64, h, q, row, x);
        }
    } else {
        ROWS_SYNC();  // chunk 0 landed, constants visible
    }

    // ---- fused final layer of the LAST block (transformer.py:182-191), the arithmetic of t2s_rows.h in the 16-token
    // layout: affine LayerNorm (eps 1e-5), Linear 128 -> 4, unpatchify; lane g of a token writes patch output p = g
    if constexpr (DO_MLP && !DO_QKV) {
        if (a.out0 != nullptr) {
            float mean, rstd;
            ln_mean_rstd16(x, 1e-5f, mean, rstd);
            float fp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                const f32x4 gam = gather16(a.f_lnw, mt, h, q), bet = gather16(a.f_lnb, mt, h, q);
                f32x4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = ln_affine(x[mt][r], mean, rstd, gam[r], bet[r]);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const f32x4 w = gather16(a.f_ow + p * D, mt, h, q);
#pragma unroll
                    for (int r = 0; r < 4; ++r) fp[p] = __builtin_fmaf(y[r], w[r], fp[p]);
                }
            }
            float ft[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) ft[p] = token_sum16(fp[p]);
            if (active) {
                const int n = tile_in_seq * 32 + row;
                const int hh = n >> 5, ww = n & 31;
                float* dst = (seq < a.split) ? a.out0 + (size_t)seq * LAT : a.out1 + (size_t)(seq - a.split) * LAT;
                const float v = g == 0 ? ft[0] : (g == 1 ? ft[1] : (g == 2 ? ft[2] : ft[3]));
                dst[(2 * ww + (g & 1)) * LATW + 2 * hh + (g >> 1)] = v + a.f_ob[g];
            }
        }
    }

    // ---------------- q, k, v of the next block ----------------
    if constexpr (DO_QKV) {
        f32x4 xm[8];
        ln_modulate16(x, xm, cm, cm + D, g, 1e-6f);
#pragma unroll 1
        for (int t = 0; t < 12; ++t) {  // output tile t = which*4 + head
            prefetch(ci);
            const f32x4* wb = wring + (ci % ROWS_SLOTS) * ROWS_CHUNK_F4 + lane;
            const int which = t >> 2, head = t & 3;
            float* base = which == 0 ? a.q : (which == 1 ? a.k : a.v);
            f32x4* dst = reinterpret_cast<f32x4*>(base) + (((size_t)seq * NH + head) * (NTOK / 32) + tile_in_seq) * 4 * 64;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            if (which < 2) {
                // q / k tile, transposed product: lane = token, registers = features (16-layout of the head's 32)
                kchunk16<false>(wb, xm, acc0, acc1);
                f32x4 o[2];
#pragma unroll
                for (int ml = 0; ml < 2; ++ml) {
                    const f32x4 bias = *reinterpret_cast<const f32x4*>(c_bq + 32 * t + 16 * ml + 4 * g);
                    const f32x4& acc = ml ? acc1 : acc0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[ml][r] = acc[r] + bias[r];
                }
                if (active) store_rows16<2>(dst, h, q, row, o);
            } else {
                // v tile with the MFMA operands swapped: lane (g, j) = output feature 16 ml + pi16(j >> 2, j & 3), registers
                // = tokens 4 g + r of this half tile: one float4 of the V^T fragment the attention kernel consumes
                kchunk16<true>(wb, xm, acc0, acc1);
                if (active) {
                    const int hb = row >> 4;     // which half of the 32-key block (wave-uniform)
#pragma unroll
                    for (int ml = 0; ml < 2; ++ml) {
                        const float bias = c_bq[32 * t + 16 * ml + tok];      // 16-layout position 4 (j >> 2) + (j & 3) = j
                        const f32x4& acc = ml ? acc1 : acc0;
                        const int d = 16 * ml + pi16(tok >> 2, tok & 3);
                        const f32x4 o = {acc[0] + bias, acc[1] + bias, acc[2] + bias, acc[3] + bias};
                        dst[(2 * hb + q) * 64 + 32 * h + d] = o;
                    }
                }
            }
            chunk_done(ci, active);
            ++ci;
        }
    }
}

template <bool DO_MLP, bool DO_QKV>
inline int launch_dit_rows16(const RowArgs& a, hipStream_t st) {
    if (a.M <= 0 || a.M % 32 != 0) {
        set_error("dit_rows16: M=%d must be a positive multiple of 32", a.M);
        return T2S_E_INVALID;
    }
    const int n16 = a.M / 16;
    dit_rows16_kernel<DO_MLP, DO_QKV><<<(n16 + 3) / 4, 256, ROWS_LDS_BYTES, st>>>(a);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

inline int dit_rows16_init() {
    T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dit_rows16_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ROWS_LDS_BYTES));
    T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dit_rows16_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ROWS_LDS_BYTES));
    T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dit_rows16_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, ROWS_LDS_BYTES));
    return T2S_OK;
}

}  // namespace t2s
