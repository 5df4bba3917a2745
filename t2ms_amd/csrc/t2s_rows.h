// The DiT's row-local chain as ONE register-resident kernel (MI355X / gfx950).
//
// Everything in a DiT block except attention is independent per token:
//   x += g1 * proj(attn)            (transformer.py:116, timm Attention.proj)
//   x += g2 * fc2(gelu(fc1(mod(LN(x)))))      (transformer.py:117, timm Mlp)
//   q,k,v = qkv(mod(LN(x)))  of the NEXT block (transformer.py:116, timm Attention.qkv)
// An f32 MFMA operand is one register per lane, and the 32x32 accumulator layout
// (lane = column, registers = rows {(r&3)+8(r>>2)+4*half}) is -- for TRANSPOSED products
// Y^T[n][tok] = W[n][:] . act[tok][:] -- exactly the B-operand layout of the next product:
// step (G,e) of a K-loop contracts the k-pair {8G+e, 8G+4+e} held by the two lane halves.
// So a wave puts 32 tokens on its lanes and carries them through the whole chain in
// registers: no intermediate HBM traffic, no LDS round trip for activations (the fc1
// activation, the post-attention residual and the LayerNorm outputs never leave the
// register file).
//
// Weights are the A operand.  A workgroup (4 waves = 128 tokens) streams them ONCE through an
// LDS ring of 16 KiB slots (three in the f32 kernel, see ROWS_SLOTS below) filled by LDS-DMA
// (global_load_lds_dwordx4) ahead of their use; a chunk = 16 fragments of 1 KiB = one 32-output tile over K = 128 (proj / fc1 / qkv)
// or the 4 x 4 (n-tile, k-group) fragments one fc1 chunk feeds into fc2.  Waves read
// fragments with conflict-free ds_read_b128 (lane-linear).  One barrier per chunk
// (= per 64 MFMAs per wave).
//
// Activations live in HBM in the fragment-major layout (t2s_common.h: frag_index), so every
// global access of this kernel is a contiguous 1 KiB per wave instruction.
//
// Per 32-token tile: 256 (proj) + 512 (fc1) + 512 (fc2) + 768 (qkv) MFMAs of 32x32x2.
#pragma once
#include "t2s_common.h"
#include <stdlib.h>

namespace t2s {

constexpr int ROWS_CHUNK_F4 = 1024;                       // float4 per chunk (16 KiB)
// LDS: [ROWS_SLOTS x 16 KiB weight ring][biases: bp 128 | b1 256 | b2 128 | bq 384][per wave: 6 adaLN vectors
// of the MLP block (768) | shift, scale of the qkv block (256)].  Every per-feature constant the
// chunk loops need is read from LDS (lgkmcnt); a global load inside those loops gets sunk next to
// its use by the compiler and waits with vmcnt(0) -- a full memory round trip, draining the
// weight DMA and all stores, once per 64 MFMAs (seen in the ISA; cost ~15 %).
constexpr int ROWS_CB_FLOATS = 896;
constexpr int ROWS_CM_FLOATS = 1024;                      // per wave (bf16x3 kernel, t2s_rows_x3.h)
#ifndef T2S_ROWS_NW
#define T2S_ROWS_NW 4
#endif
// f32 kernel: a THREE-slot ring, the DMA ROWS_DIST = 2 chunks ahead behind COUNTED vmcnt waits, and 768 adaLN floats
// per wave (the MLP phase never reads shift_msa / scale_msa of its block: those two slots hold the qkv block's shift /
// scale) -- 49,152 + 3,584 + 12,288 = 65,024 B.  With the two-slot ring every chunk ended in `vmcnt(0)`: a weight chunk
// had one chunk time (1.7 us) to arrive, which a lone wave per SIMD (the 32-series shard of an 8-GPU strong-scaling
// run: 960 tiles for 1024 SIMDs) cannot hide.  Same-box A/B (tools/ab_sample.sh, series/s at B = 256 / 128, two sampler
// lanes): two slots 61.6 / 61.0, three slots 62.2 / 61.9, FOUR slots (81,408 B) 61.4 / 58.2 -- the larger footprint
// keeps the other lane's kernels off the CU, so the ring stays as small as the distance allows.
#ifndef T2S_ROWS_SLOTS
#define T2S_ROWS_SLOTS 3
#endif
constexpr int ROWS_SLOTS = T2S_ROWS_SLOTS;
#ifndef T2S_ROWS_DIST
#define T2S_ROWS_DIST 2
#endif
constexpr int ROWS_DIST = T2S_ROWS_DIST;                  // chunks the DMA runs ahead (< ROWS_SLOTS)
static_assert(ROWS_DIST >= 1 && ROWS_DIST < ROWS_SLOTS, "the DMA may run at most ROWS_SLOTS - 1 chunks ahead");
constexpr int ROWS_CMF = 768;                             // per-wave adaLN floats of the f32 kernel
constexpr int ROWS_LDS_BYTES = ROWS_SLOTS * ROWS_CHUNK_F4 * 16 + (ROWS_CB_FLOATS + T2S_ROWS_NW * ROWS_CMF) * 4;

struct RowArgs {
    float* x;          // (M,128) residual stream, fragment-major, in place
    const float* x_in; // where the stream is READ at kernel entry: x itself, or (block 0 of a CFG pass) the patchified
    int in_seqs;       //   tokens of the in_seqs distinct sequences, sequence s reading slot s % in_seqs
    // last block only (DO_MLP && !DO_QKV): fused final layer when out0 != NULL; sequences [0,split) -> out0, rest -> out1;
    // keep_x = 0 skips the store of the final residual stream (only the t2s_dit_read_stream tap reads it)
    const float *f_lnw, *f_lnb, *f_ow, *f_ob;
    float *out0, *out1;
    int split, keep_x;
    // <qkv only> kernels (block 0's LayerNorm + qkv): when p_lat != NULL the tokens are not read from x_in but GENERATED
    // from the latent (patchify, transformer.py:166-172: conv 2x2 stride 2 -> Linear 4 -> 128, + pos_embed) and, for the
    // in_seqs distinct sequences, written to x_in's buffer for block 0's <proj + MLP> kernel -- the stand-alone patchify
    // launch (and one read of its output) leaves the sampling loop's critical path
    const float* p_lat;   // (p_B, 64, 30); sequence s reads latent row s % p_B
    int p_B;
    const float *p_cw, *p_cb, *p_pw, *p_pb, *p_pos;   // conv.weight (4,1,2,2) / bias (4), patch_emb.weight (128,4) / bias, pos_embed (480,128)
    const float* ao;   // (M,128) attention output (pre-proj), fragment-major
    const float* mod;  // (S,MODROW) adaLN modulation of this pass -- or, when mod_step != NULL, the sampler's table
    // (steps, mod_rows, MODROW) of the WHOLE run (t2s_sampler.hip: row 0 of a step = the text-free branch, row 1 + b = batch
    // row b): sequence s of this pass reads step *mod_step, row 0 if s < mod_uncond, else 1 + mod_row0 + (s - mod_uncond)
    const int* mod_step;
    int mod_rows, mod_uncond, mod_row0;
    int M;             // S*480 (multiple of 32)
    int blk;           // block whose proj+MLP run (ignored if !DO_MLP)
    int qkv_blk;       // block whose LN1/modulate/qkv run (ignored if !DO_QKV)
    const f32x4 *Wp, *W1, *W2c, *Wq;    // packed weights; W2c is fc2 in chunk order [c][nt][g]
    const float *bp, *b1, *b2, *bq;
    float *q, *k, *v;  // per head (S*4, 480, 32): q, k fragment-major; v TRANSPOSED fragment-major (V^T)
};

// patchify of one token (shared by the 32- and the 16-token kernel: the same expression, hence the same bits)
template <class Args>   // RowArgs or RowArgsX3 (t2s_rows_x3.h): the same p_* fields
__device__ __forceinline__ void patch_conv(const Args& a, int seq, int n, float (&cv)[4]) {
    const int hh = n >> 5, ww = n & 31;
    const float* xin = a.p_lat + (size_t)(seq % a.p_B) * LAT;
    float px[4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) px[i * 2 + j] = xin[(2 * ww + j) * LATW + 2 * hh + i];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float acc = a.p_cw[c * 4 + 0] * px[0];
        acc = __builtin_fmaf(a.p_cw[c * 4 + 1], px[1], acc);
        acc = __builtin_fmaf(a.p_cw[c * 4 + 2], px[2], acc);
        acc = __builtin_fmaf(a.p_cw[c * 4 + 3], px[3], acc);
        cv[c] = acc + a.p_cb[c];
    }
}
__device__ __forceinline__ float patch_feature(const float (&cv)[4], f32x4 w, float pb, float pos) {
    float acc = w.x * cv[0];
    acc = __builtin_fmaf(w.y, cv[1], acc);
    acc = __builtin_fmaf(w.z, cv[2], acc);
    acc = __builtin_fmaf(w.w, cv[3], acc);
    return (acc + pb) + pos;
}

__device__ __forceinline__ const float* mod_row_of(const RowArgs& a, int seq) {
    if (a.mod_step == nullptr) return a.mod + (size_t)seq * MODROW;
    const int r = seq < a.mod_uncond ? 0 : 1 + a.mod_row0 + (seq - a.mod_uncond);
    return a.mod + ((size_t)(*a.mod_step) * a.mod_rows + r) * MODROW;
}

// GELU(tanh): 0.5 x (1 + tanh(u)) == x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3).
// On gfx950 the f32 MFMA shares the VALU lanes, so every VALU instruction here is paid in matrix
// time: 7 instructions (x*x, fma, mul, v_exp, add, v_rcp, mul) instead of an IEEE division.
__device__ __forceinline__ float gelu_tanh_f(float x) {
    constexpr float C0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;   // -2 sqrt(2/pi) log2(e)
    constexpr float C1 = C0 * 0.044715f;
    const float t = x * x;
    const float a = x * __builtin_fmaf(t, C1, C0);          // -2u * log2(e)
    const float e = __builtin_amdgcn_exp2f(a);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// per-register constants: feature n = 32*nt + 8*g + 4*half + e  <->  register 4g+e of tile nt
__device__ __forceinline__ f32x4 ldc4(const float* __restrict__ vec, int nt, int g, int half) {
    return *reinterpret_cast<const f32x4*>(vec + 32 * nt + 8 * g + 4 * half);
}

// ---- arithmetic shared by the 32-token kernel below and the 16-token kernel (t2s_rows16.h) --------------------------------
// The two kernels must give a token the SAME BITS (a 32-series strong-scaling shard runs the 16-token kernel, the 256-series
// batch the 32-token one, and the shard's rows have to equal the batch's): every fused multiply-add is spelled out, and the
// LayerNorm sums use a reduction tree both lane layouts can follow -- four partial sums P0..P3 over the features a lane
// group of the 16-token layout owns (t2s_rows16.h: feature 16 mt + pi(g, r)), each added up in ascending (mt, r) order, then
// (P0 + P2) + (P1 + P3).  A lane half h of the 32-token layout owns P_h (its elements e = 0, 2) and P_{h+2} (e = 1, 3).
__device__ __forceinline__ float res_gate(float x, float gate, float acc, float bias) {      // x + gate (acc + bias)
    return __builtin_fmaf(gate, acc + bias, x);
}
__device__ __forceinline__ float ln_y(float x, float mean, float rstd, float sc, float sh) {   // modulate(LN(x))
    return __builtin_fmaf((x - mean) * rstd, 1.0f + sc, sh);
}
__device__ __forceinline__ float ln_rstd(float ss_total, float eps) { return rsqrtf(ss_total * (1.0f / 128.0f) + eps); }
__device__ __forceinline__ float ln_affine(float x, float mean, float rstd, float gam, float bet) {   // final LayerNorm (affine)
    return __builtin_fmaf((x - mean) * rstd, gam, bet);
}

// mean and 1/std over the 128 features a lane pair holds (32-token layout), in the shared reduction tree
__device__ __forceinline__ void ln_mean_rstd32(const f32x16 (&x)[4], float eps, float& mean, float& rstd) {
    float s0 = 0.f, s1 = 0.f;                       // P_half (e = 0, 2) and P_{half+2} (e = 1, 3)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            s0 += x[nt][4 * g + 0];
            s1 += x[nt][4 * g + 1];
            s0 += x[nt][4 * g + 2];
            s1 += x[nt][4 * g + 3];
        }
    float s = s0 + s1;
    s += xhalf(s);
    mean = s * (1.0f / 128.0f);
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float d0 = x[nt][4 * g + 0] - mean, d1 = x[nt][4 * g + 1] - mean;
            const float d2 = x[nt][4 * g + 2] - mean, d3 = x[nt][4 * g + 3] - mean;
            q0 = __builtin_fmaf(d0, d0, q0);
            q1 = __builtin_fmaf(d1, d1, q1);
            q0 = __builtin_fmaf(d2, d2, q0);
            q1 = __builtin_fmaf(d3, d3, q1);
        }
    float ss = q0 + q1;
    ss += xhalf(ss);
    rstd = ln_rstd(ss, eps);
}

// LayerNorm (no affine, eps) + modulate over the 128 features a lane pair holds
__device__ __forceinline__ void ln_modulate(const f32x16 (&x)[4], f32x16 (&y)[4],
                                            const float* __restrict__ shift,
                                            const float* __restrict__ scale, int half, float eps) {
    float mean, rstd;
    ln_mean_rstd32(x, eps, mean, rstd);
    // The second pass recomputes x - mean from x: hipcc otherwise keeps the 64 differences of the variance pass alive
    // next to x and y (192 registers) and spills some of them -- and every scratch reload waits vmcnt(0), draining
    // the weight DMA and the parking stores.  The empty asm makes `mean` opaque so the subtraction is not CSE'd.
    float mean2 = mean;
    asm volatile("" : "+v"(mean2));
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 sc = ldc4(scale, nt, g, half);
            const f32x4 sh = ldc4(shift, nt, g, half);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[nt][4 * g + e] = ln_y(x[nt][4 * g + e], mean2, rstd, sc[e], sh[e]);
        }
}


// One 32-output x K=128 weight tile (16 fragments at wb[G * 64]) against a B operand in registers, software-pipelined:
// fragment G+1 is fetched from LDS BEFORE the four MFMAs of fragment G are issued (hipcc otherwise emits
// `ds_read_b128 ; s_waitcnt lgkmcnt(0) ; 4 x v_mfma` with ONE fragment register).  sched_barrier pins the order; with the
// weight DMA hidden from the waitcnt pass (glds16_asm) the wait becomes the counted lgkmcnt(1).  Measured: the LDS round
// trip was already covered by the dependent 64-cycle MFMA chain -- timing-only builds without the fragment reads, without the
// DMA or without the chunk barriers are all within 3 % of the real kernel (profiles/r03_rows_ablations.txt); what a tile
// pays beyond its MFMA + VALU time is its prologue (x / attention-output rows from HBM), launch and tail.
#ifndef T2S_ROWS_PIPE
#define T2S_ROWS_PIPE 1
#endif
#define T2S_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
template <bool SWAP, typename BOP>
__device__ __forceinline__ void ktile_mfma(const f32x4* __restrict__ wb, BOP&& bop, f32x16& acc) {
#if T2S_ROWS_PIPE
    f32x4 w = wb[0];
#pragma unroll
    for (int G = 0; G < 16; ++G) {
        f32x4 wn = w;
        if (G + 1 < 16) wn = wb[(G + 1) * 64];
        T2S_SCHED_FENCE();
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = SWAP ? mfma32(bop(G, e), w[e], acc) : mfma32(w[e], bop(G, e), acc);
        T2S_SCHED_FENCE();
        w = wn;
    }
#else
#pragma unroll
    for (int G = 0; G < 16; ++G) {
        const f32x4 w = wb[G * 64];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = SWAP ? mfma32(bop(G, e), w[e], acc) : mfma32(w[e], bop(G, e), acc);
    }
#endif
}


// NW = waves per workgroup: 4 (two workgroups per CU; the default) or 8 (one; each wave then issues half the
// weight DMA pieces and the weights cross L2 -> LDS once per 256 tokens).  Measured: 8 is slower, 510 vs 480 us
// average per launch -- one barrier domain of 8 waves loses more than the halved DMA issue gains.
#ifndef T2S_ROWS_NW
#define T2S_ROWS_NW 4
#endif
constexpr int ROWS_NW = T2S_ROWS_NW;
// chunk barrier of the row kernels: the weight DMA is issued by inline asm (glds16_asm: invisible to hipcc's waitcnt
// pass, so that the ds_read -> MFMA pipelines keep their counted lgkmcnt waits), hence the explicit vmcnt(0) in front of
// __syncthreads(), which by itself only covers what the compiler tracks
#define ROWS_SYNC()                                          \
    do {                                                     \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     \
        __syncthreads();                                     \
    } while (0)

// the same, but the 16 YOUNGEST vector-memory operations of the wave may stay in flight (the tile's residual stream, issued last
// on purpose).  A raw barrier: __syncthreads()' release fence would make hipcc wait for vmcnt(0) again.
#define ROWS_SYNC_BUT16()                                                \
    do {                                                                 \
        asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_s_barrier();                                    \
    } while (0)

template <bool DO_MLP, bool DO_QKV>
__global__ __launch_bounds__(64 * ROWS_NW, 8 / ROWS_NW) void dit_rows_kernel(const RowArgs a) {
    extern __shared__ __attribute__((aligned(16))) f32x4 wring[];  // [ROWS_SLOTS][1024]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform: scalar addressing
    const int half = lane >> 5;
    const int n_tiles = a.M >> 5;
    int tile = blockIdx.x * ROWS_NW + wave;       // 32-token tile of this wave
    const bool active = tile < n_tiles;           // tail waves compute on a clamped tile, store nothing
    if (!active) tile = n_tiles - 1;
    const int seq = (tile * 32) / NTOK;           // 480 = 15*32: a tile never straddles sequences
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
    // each wave DMAs fragments {wave, wave+4, wave+8, wave+12} of the chunk
    auto fill = [&](int ci) {
        const f32x4* src = chunk_src(ci) + lane;
        f32x4* dst = wring + (ci % ROWS_SLOTS) * ROWS_CHUNK_F4;
#pragma unroll
        for (int p = 0; p < 16 / ROWS_NW; ++p) glds16_asm(src + (wave + ROWS_NW * p) * 64, dst + (wave + ROWS_NW * p) * 64);
    };
    // start of chunk ci: keep the DMA ROWS_DIST chunks ahead.  Slot (ci + DIST) % SLOTS held chunk ci + DIST - SLOTS,
    // whose reads every wave finished before the barrier that ended chunk ci - 1 (DIST < SLOTS).
    auto prefetch = [&](int ci) {
        if (ci + ROWS_DIST < N_CHUNKS) fill(ci + ROWS_DIST);
    };
    // end of chunk ci: chunk ci + 1 must have landed for every wave -> counted wait (the DIST - 1 younger chunks, 16 /
    // NW pieces each, stay in flight; any younger stores only make the wait stricter), then the workgroup barrier.
    // `own_stores` = VM ops this wave issued after the youngest DMA that may stay in flight too (the qkv epilogue's).
    auto chunk_done = [&](int ci, int own_stores) {
        constexpr int PER = 16 / ROWS_NW;
        if (ci + ROWS_DIST < N_CHUNKS) {
            if (own_stores > 0)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ROWS_DIST - 1) * PER + ROWS_DIST * 4) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ROWS_DIST - 1) * PER) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    };

#pragma unroll
    for (int c0 = 0; c0 < ROWS_DIST; ++c0) fill(c0);

    // ---- stage the per-feature constants in LDS (visible after the first barrier) ----
    float* cb = reinterpret_cast<float*>(wring + ROWS_SLOTS * ROWS_CHUNK_F4);
    // per wave: [0,256) shift_msa | scale_msa of the QKV block, [256,768) gate_msa | shift_mlp | scale_mlp | gate_mlp
    // of the MLP block (whose own shift_msa / scale_msa were consumed by the previous kernel)
    float* cm = cb + ROWS_CB_FLOATS + wave * ROWS_CMF;
    // Order of the prologue's vector-memory operations (round 5, as in t2s_rows_x3.h): the weight DMAs, the attention output
    // (requested FIRST: its round trip runs under the constants'), the constants as ONE batch of loads then one batch of
    // ds_writes (written as load -> store pairs hipcc waits vmcnt(0) in front of every store: three serial L2 round trips
    // before the tile's own loads were even issued), and LAST the residual stream, which the first barrier leaves in flight
    // (ROWS_SYNC_BUT16: it is first used after the four proj chunks).
    static_assert(ROWS_NW == 4, "the constants below are staged by 256 threads");
    f32x16 bop[DO_MLP ? 4 : 1];  // B operand of proj: ao[row][8G+4half+e] at bop[G>>2][4(G&3)+e]
    if constexpr (DO_MLP) {
        const f32x4* ar = reinterpret_cast<const f32x4*>(a.ao) + (size_t)tile * 16 * 64 + lane;
#pragma unroll
        for (int G = 0; G < 16; ++G) {
            const f32x4 t = ar[G * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) bop[G >> 2][4 * (G & 3) + e] = t[e];
        }
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
        if constexpr (DO_MLP) {
            cb[t0] = c0;
            cb[t1] = c1;
            *reinterpret_cast<f32x4*>(cm + (64 + lane) * 4) = m1;
            *reinterpret_cast<f32x4*>(cm + (128 + lane) * 4) = m2;
        }
        if constexpr (DO_QKV) {
            cb[512 + t0] = q0;
            if (t1 < 384) cb[512 + t1] = q1;
            *reinterpret_cast<f32x4*>(cm + lane * 4) = mq;
        }
    }
    const float* c_bp = cb;
    const float* c_b1 = cb + 128;
    const float* c_b2 = cb + 384;
    const float* c_bq = cb + 512;

    // residual stream of this lane's token, accumulator layout: x[nt][4g+e] = X[row][32nt+8g+4half+e]
    f32x16 x[4];
    const int tile_src = tile - (seq - seq % a.in_seqs) * (NTOK / 32);   // same tile of sequence seq % in_seqs
    bool generated = false;
    if constexpr (!DO_MLP) {
        if (a.p_lat != nullptr) {
            // patchify in the prologue: patch_emb weight in the (unused) proj / MLP bias slots of LDS, bias in the wave's
            // (unused) MLP adaLN slots; visible after this barrier (chunk 0 is waited for again further down)
            for (int i = threadIdx.x; i < 512; i += 64 * ROWS_NW) cb[i] = a.p_pw[i];
            *reinterpret_cast<f32x4*>(cm + 256 + (lane & 31) * 4) = *reinterpret_cast<const f32x4*>(a.p_pb + (lane & 31) * 4);
            __syncthreads();
            const int n = (tile - seq * (NTOK / 32)) * 32 + (lane & 31);
            float cv[4];
            patch_conv(a, seq, n, cv);
            const float* posrow = a.p_pos + (size_t)n * D;
#pragma unroll
            for (int G = 0; G < 16; ++G) {
                const int d0 = 8 * G + 4 * half;
                const f32x4 pos4 = *reinterpret_cast<const f32x4*>(posrow + d0);
                const f32x4 pb4 = *reinterpret_cast<const f32x4*>(cm + 256 + d0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    x[G >> 2][4 * (G & 3) + e] = patch_feature(cv, *reinterpret_cast<const f32x4*>(cb + (d0 + e) * 4), pb4[e], pos4[e]);
            }
            if (active && seq < a.in_seqs) {   // block 0's <proj + MLP> kernel reads the tokens of sequence s % in_seqs
                f32x4* xo = const_cast<f32x4*>(reinterpret_cast<const f32x4*>(a.x_in)) + (size_t)tile_src * 16 * 64 + lane;
#pragma unroll
                for (int G = 0; G < 16; ++G) {
                    f32x4 t;
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = x[G >> 2][4 * (G & 3) + e];
                    xo[G * 64] = t;
                }
            }
            generated = true;
        }
    }
    auto load_x = [&]() {
        const f32x4* xr = reinterpret_cast<const f32x4*>(a.x_in) + (size_t)tile_src * 16 * 64 + lane;
#pragma unroll
        for (int G = 0; G < 16; ++G) {
            const f32x4 t = xr[G * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) x[G >> 2][4 * (G & 3) + e] = t[e];
        }
    };
    if (!DO_MLP && !generated) load_x();
    int ci = 0;

    if constexpr (DO_MLP) {
        const float* mb = cm;   // [shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp] of a.blk
        // ---------------- x += gate_msa * (proj(ao) + b) ----------------
        {
            __builtin_amdgcn_sched_barrier(0);
            load_x();
            __builtin_amdgcn_sched_barrier(0);
            ROWS_SYNC_BUT16();  // chunk 0 landed, ao and the constants here; the residual stream (16 loads) may still be on its way
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                prefetch(ci);
                const f32x4* wb = wring + (ci % ROWS_SLOTS) * ROWS_CHUNK_F4 + lane;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                ktile_mfma<false>(wb, [&](int G, int e) { return bop[G >> 2][4 * (G & 3) + e]; }, acc);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bias = ldc4(c_bp, nt, g, half);
                    const f32x4 gate = ldc4(mb + 2 * D, nt, g, half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[nt][4 * g + e] = res_gate(x[nt][4 * g + e], gate[e], acc[4 * g + e], bias[e]);
                }
                chunk_done(ci, 0);
                ++ci;
            }
        }
        // ---------------- x += gate_mlp * (fc2(gelu(fc1(mod(LN(x))))) + b2) ----------------
        f32x4* xw = reinterpret_cast<f32x4*>(a.x) + (size_t)tile * 16 * 64 + lane;
        {
            f32x16 xm[4];
            ln_modulate(x, xm, mb + 3 * D, mb + 4 * D, half, 1e-6f);
            // Park the post-attention residual in its own HBM slot (each lane re-reads exactly
            // what it wrote) so the 1024-MFMA MLP loop does not carry 64 more live registers.
            if (active) {
#pragma unroll
                for (int G = 0; G < 16; ++G) {
                    f32x4 t;
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = x[G >> 2][4 * (G & 3) + e];
                    xw[G * 64] = t;
                }
            }
            f32x16 acc[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll 1
            for (int c = 0; c < 8; ++c) {  // 32 hidden units per chunk; ci = 4 + 2c (even) here
                prefetch(ci);
                f32x16 hT;
                {
                    const f32x4* wb = wring + (ci % ROWS_SLOTS) * ROWS_CHUNK_F4 + lane;
#pragma unroll
                    for (int r = 0; r < 16; ++r) hT[r] = 0.f;
                    ktile_mfma<false>(wb, [&](int G, int e) { return xm[G >> 2][4 * (G & 3) + e]; }, hT);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 bias = *reinterpret_cast<const f32x4*>(c_b1 + 32 * c + 8 * g + 4 * half);
#pragma unroll
                        for (int e = 0; e < 4; ++e) hT[4 * g + e] = gelu_tanh_f(hT[4 * g + e] + bias[e]);
                    }
                }
                chunk_done(ci, 0);
                ++ci;
                prefetch(ci);
                {   // fc2 partial over k-groups 4c..4c+3 of K=256; fragments ordered [nt][g]
                    const f32x4* wb = wring + (ci % ROWS_SLOTS) * ROWS_CHUNK_F4 + lane;
#if T2S_ROWS_PIPE
                    f32x4 w = wb[0];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {      // i = 4 g + nt: same order of additions as before
                        const int g = i >> 2, nt = i & 3;
                        f32x4 wn = w;
                        if (i + 1 < 16) wn = wb[(((i + 1) & 3) * 4 + ((i + 1) >> 2)) * 64];
                        T2S_SCHED_FENCE();
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[nt] = mfma32(w[e], hT[4 * g + e], acc[nt]);
                        T2S_SCHED_FENCE();
                        w = wn;
                    }
#else
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) {
                            const f32x4 w = wb[(nt * 4 + g) * 64];
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[nt] = mfma32(w[e], hT[4 * g + e], acc[nt]);
                        }
                    }
#endif
                }
                chunk_done(ci, 0);
                ++ci;
            }
            // the parked residual comes back in ONE batch of 16 loads; with the (wave-uniform) store condition inside
            // the element loop hipcc branched around every store and waited vmcnt(0) after every load: 16 dependent
            // L2 round trips per tile, which also drained the weight DMA each time
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 xo = xw[(nt * 4 + g) * 64];
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[nt][4 * g + e] = xo[e];
                }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bias = ldc4(c_b2, nt, g, half);
                    const f32x4 gate = ldc4(mb + 5 * D, nt, g, half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[nt][4 * g + e] = res_gate(x[nt][4 * g + e], gate[e], acc[nt][4 * g + e], bias[e]);
                }
            if (active && (DO_QKV || a.out0 == nullptr || a.keep_x)) {   // final residual stream of this block
#pragma unroll
                for (int G = 0; G < 16; ++G) {
                    f32x4 t;
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = x[G >> 2][4 * (G & 3) + e];
                    xw[G * 64] = t;
                }
            }
        }
    } else {
        ROWS_SYNC();  // chunk 0 landed
    }

    // ---- fused final layer of the LAST block (transformer.py:182-191): affine LayerNorm (eps 1e-5),
    // Linear 128 -> 4, unpatchify out[s][(2ww+pw)*30 + 2hh+ph] = y[ph*2+pw]; the lane pair of a token holds its
    // 128 features, so everything is lane-local up to one cross-half add per output
    if constexpr (DO_MLP && !DO_QKV) {
        if (a.out0 != nullptr) {
            float mean, rstd;
            ln_mean_rstd32(x, 1e-5f, mean, rstd);
            // four dot products over the token's 128 features, in the shared reduction tree: partial fe over the lane's
            // elements e = 0, 2 (feature group P_half), fo over e = 1, 3 (P_{half+2}), each in ascending feature order
            float fe[4] = {0.f, 0.f, 0.f, 0.f}, fo[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = 32 * nt + 8 * g + 4 * half;
                    const f32x4 gam = *reinterpret_cast<const f32x4*>(a.f_lnw + col);
                    const f32x4 bet = *reinterpret_cast<const f32x4*>(a.f_lnb + col);
                    f32x4 y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = ln_affine(x[nt][4 * g + e], mean, rstd, gam[e], bet[e]);
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(a.f_ow + p * D + col);
                        fe[p] = __builtin_fmaf(y[0], w[0], fe[p]);
                        fo[p] = __builtin_fmaf(y[1], w[1], fo[p]);
                        fe[p] = __builtin_fmaf(y[2], w[2], fe[p]);
                        fo[p] = __builtin_fmaf(y[3], w[3], fo[p]);
                    }
                }
            float fl[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) fl[p] = fe[p] + fo[p];
            const float f0 = fl[0] + xhalf(fl[0]), f1 = fl[1] + xhalf(fl[1]);
            const float f2 = fl[2] + xhalf(fl[2]), f3 = fl[3] + xhalf(fl[3]);
            if (active) {   // lane half 0 writes patch outputs p = 0,1; half 1 writes p = 2,3
                const int n = (tile - seq * (NTOK / 32)) * 32 + (lane & 31);
                const int hh = n >> 5, ww = n & 31;
                float* dst = (seq < a.split) ? a.out0 + (size_t)seq * LAT : a.out1 + (size_t)(seq - a.split) * LAT;
                // p = 2 half + q2 -> element (2 ww + q2, 2 hh + half); selects, not an indexed array (that went to scratch)
                const float ob0 = half ? a.f_ob[2] : a.f_ob[0], ob1 = half ? a.f_ob[3] : a.f_ob[1];
                dst[(2 * ww + 0) * LATW + 2 * hh + half] = (half ? f2 : f0) + ob0;
                dst[(2 * ww + 1) * LATW + 2 * hh + half] = (half ? f3 : f1) + ob1;
            }
        }
    }
    if constexpr (DO_QKV) {
        f32x16 xm[4];
        ln_modulate(x, xm, cm, cm + D, half, 1e-6f);
        const int tile_in_seq = tile - seq * (NTOK / 32);
#pragma unroll 1
        for (int t = 0; t < 12; ++t) {  // output tile t = which*4 + head
            prefetch(ci);
            const f32x4* wb = wring + (ci % ROWS_SLOTS) * ROWS_CHUNK_F4 + lane;
            const int which = t >> 2, head = t & 3;
            float* base = which == 0 ? a.q : (which == 1 ? a.k : a.v);
            f32x4* dst = reinterpret_cast<f32x4*>(base) +
                         (((size_t)seq * NH + head) * (NTOK / 32) + tile_in_seq) * 4 * 64 + lane;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (which < 2) {
                // q / k tile, transposed product: lane = token, registers = features d
                f32x4 bias[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    bias[g] = *reinterpret_cast<const f32x4*>(c_bq + 32 * t + 8 * g + 4 * half);
                ktile_mfma<false>(wb, [&](int G, int e) { return xm[G >> 2][4 * (G & 3) + e]; }, acc);
                if (active) {
                    {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            f32x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = acc[4 * g + e] + bias[g][e];
                            dst[g * 64] = o;
                        }
                    }
                }
            } else {
                // v tile with the MFMA operands swapped: lane = feature d, registers = tokens
                // {8g+4half+e}: exactly the V^T fragment the attention kernel consumes
                const float bias = c_bq[32 * t + (lane & 31)];
                ktile_mfma<true>(wb, [&](int G, int e) { return xm[G >> 2][4 * (G & 3) + e]; }, acc);
                if (active) {
                    {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            f32x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = acc[4 * g + e] + bias;
                            dst[g * 64] = o;
                        }
                    }
                }
            }
            // Counted wait + raw barrier: the next chunk's DMA pieces must have landed; the younger chunks and the
            // q/k/v stores of the last ROWS_DIST tiles (4 each) stay in flight across the barrier.
            // (__syncthreads() would drain them: vmcnt(0).)  Tail waves store nothing.
            chunk_done(ci, active ? 4 : 0);
            ++ci;
        }
    }
}

template <bool DO_MLP, bool DO_QKV>
inline int launch_dit_rows(const RowArgs& a, hipStream_t st) {
    if (a.M <= 0 || a.M % 32 != 0) {
        set_error("dit_rows: M=%d must be a positive multiple of 32", a.M);
        return T2S_E_INVALID;
    }
    const int tiles = a.M / 32;
    static const int extra_lds = getenv("T2S_ROWS_EXTRA_LDS") ? atoi(getenv("T2S_ROWS_EXTRA_LDS")) : 0;  // diagnostic: force fewer workgroups per CU
    if (extra_lds > 0) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dit_rows_kernel<DO_MLP, DO_QKV>), hipFuncAttributeMaxDynamicSharedMemorySize, ROWS_LDS_BYTES + extra_lds);
    dit_rows_kernel<DO_MLP, DO_QKV><<<(tiles + ROWS_NW - 1) / ROWS_NW, 64 * ROWS_NW, ROWS_LDS_BYTES + extra_lds, st>>>(a);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

// raise the dynamic-LDS cap of the three instantiations (t2s_dit_create; never under stream capture)
inline int dit_rows_init() {
    T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dit_rows_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ROWS_LDS_BYTES));
    T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dit_rows_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ROWS_LDS_BYTES));
    T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dit_rows_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, ROWS_LDS_BYTES));
    return T2S_OK;
}

}  // namespace t2s
