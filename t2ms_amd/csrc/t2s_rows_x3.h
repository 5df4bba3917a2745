// The row-local chain of a DiT block (t2s_rows.h) in "bf16x3" arithmetic (t2s_x3.h): the same
// register-resident chain, chunk order and LDS-DMA weight ring, with every fp32 product evaluated
// as six bf16 MFMAs (fp32-accurate) -- 48 v_mfma_f32_32x32x16_bf16 per chunk instead of 64
// v_mfma_f32_32x32x2_f32 at half the cycles each.  T2S_MATH_BF16X3: what infer.py / Sampler select by default since round 5
// (the bench headline stays on the f32 kernels).
//
// Operand flow.  The 32x32 fp32 accumulator layout (lane = token, registers = features) is still the
// next product's B operand: registers 8s..8s+7 of feature tile nt are k-step 2nt+s, split into three
// bf16 planes in registers (split3_acc).  Where the register budget allows the planes stay resident
// for the whole GEMM (attention output for proj, LayerNorm outputs for fc1 and qkv: 96 registers; the
// fp32 copy is dead by then, and the residual stream is parked in HBM during the MLP loop).
//
// Weights: three bf16 planes in chunk order, 24 KiB per chunk = 24 LDS-DMA pieces of 1 KiB:
//   K = 128 chunks (proj / fc1 / qkv tile of 32 outputs): piece (ks, p), ks = k-step 0..7, p = plane,
//     lane (n & 31, h) holds W[n][32 (ks>>1) + 16 (ks&1) + 8 (j>>2) + 4 h + (j&3)], j = 0..7;
//   fc2 chunk c (the 32 hidden units of fc1 chunk c):  piece (nt, s, p), lane holds
//     W2[32 nt + (n & 31)][32 c + 16 s + 8 (j>>2) + 4 h + (j&3)].
#pragma once
#include "t2s_rows.h"
#include "t2s_x3.h"

namespace t2s {

// Raise the wave's issue priority for its VALU-heavy sections (LayerNorm, GELU, operand splits): the two
// waves of a SIMD are arbitrated by priority, then age, and a partner in an MFMA section needs the issue
// port for only 8 of every 32 cycles (measured: 375 -> 364 us average per launch).
#ifndef T2S_X3_NO_PRIO
#define X3_PRIO(p) __builtin_amdgcn_s_setprio(p);
#else
#define X3_PRIO(p)
#endif

// -DT2S_X3_STAMP (tools/x3_stamp.sh; diagnosis only): every wave attributes the s_memtime cycles between consecutive stamps to a
// category -- 0 prologue (first loads / operand split until chunk 0 has landed), 1 MFMA groups, 2 VALU sections (LayerNorm,
// GELU, splits, gate / residual), 3 `s_waitcnt vmcnt(0)` in front of a chunk barrier, 4 the barrier itself, 5 issuing global
// loads / stores and LDS-DMA, 6 epilogue -- and the first workgroups write their sums to RowArgsX3::stamp.
#ifdef T2S_X3_STAMP
__device__ __forceinline__ unsigned long long x3_clk() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define X3_STAMP_DECL unsigned long long st_last = x3_clk(), st_acc[7] = {0, 0, 0, 0, 0, 0, 0}; const unsigned long long st_t0 = st_last;
#define X3_STAMP(k) { const unsigned long long n_ = x3_clk(); st_acc[k] += n_ - st_last; st_last = n_; }
#define X3_SYNC() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); X3_STAMP(3) __builtin_amdgcn_s_barrier(); X3_STAMP(4) }
#define X3_SYNC_BUT16() { asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory"); X3_STAMP(3) __builtin_amdgcn_s_barrier(); X3_STAMP(4) }
#else
#define X3_STAMP_DECL
#define X3_STAMP(k)
// (explicit vmcnt(0): every weight DMA of this kernel is issued by untracked inline asm, the fence inside wg_sync() only covers
// what hipcc tracks -- as ROWS_SYNC in t2s_rows.h)
#define X3_SYNC() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); wg_sync(); }
// the same, but the 16 YOUNGEST vector-memory operations of the wave may still be in flight (the tile's residual stream, issued
// last on purpose: it is first needed after the proj chunks).  Not wg_sync(): its release fence waits for vmcnt(0).
#define X3_SYNC_BUT16() { asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
#endif
constexpr int X3_CHUNK_UNITS = 24 * 64;                    // 16-byte units per chunk (24 KiB)
#ifdef T2S_X3_LONE   // diagnosis (tools/x3_variant.sh x3_lone -DT2S_X3_LONE): pad the allocation so ONE workgroup fits a CU = one wave per SIMD
constexpr int ROWS_X3_LDS_BYTES = 100 * 1024;
#else
constexpr int ROWS_X3_LDS_BYTES = 2 * X3_CHUNK_UNITS * 16 + (ROWS_CB_FLOATS + 4 * ROWS_CM_FLOATS) * 4;
#endif

struct RowArgsX3 {
    float* x;          // (M,128) residual stream, fragment-major, in place
    const float* x_in; // where the stream is READ at kernel entry: x itself, or (block 0 of a CFG pass) the patchified
    int in_seqs;       //   tokens of the in_seqs distinct sequences, sequence s reading slot s % in_seqs
    // <qkv only> kernel of block 0 with p_lat != NULL: the tokens are GENERATED in the prologue from the latent (patchify,
    // exactly as in t2s_rows.h: same helpers, same bits as patchify_kernel) and written to x_in for block 0's <proj + MLP> kernel
    const float* p_lat;
    int p_B;
    const float *p_cw, *p_cb, *p_pw, *p_pb, *p_pos;
    // last block only (DO_MLP && !DO_QKV): fused final layer when out0 != NULL; sequences [0,split) -> out0, rest -> out1;
    // keep_x = 0 skips the store of the final residual stream (only the t2s_dit_read_stream tap reads it)
    const float *f_lnw, *f_lnb, *f_ow, *f_ob;
    float *out0, *out1;
    int split, keep_x;
    const float* ao;   // (M,128) attention output (pre-proj), fragment-major
    const float* mod;  // (S,MODROW), or the sampler's whole-run table when mod_step != NULL (see RowArgs, t2s_rows.h)
    const int* mod_step;
    int mod_rows, mod_uncond, mod_row0;
    int M;
    int blk;
    int qkv_blk;
    const bf16x8 *Wp, *W1, *W2c, *Wq;   // split weights in chunk order (see above)
    const float *bp, *b1, *b2, *bq;
    float* q;          // q fragment-major fp32 (the attention scales and splits it once per head)
    __bf16 *k3, *v3;   // k, V^T split planes (t2s_x3.h)
#ifdef T2S_X3_STAMP
    unsigned long long* stamp;   // [workgroup < 256][wave][8]: 7 category sums + total
#endif
};

// fp32 packed weights (packed_index order, or the fc2 chunk order of pack_weight_kernel mode 1) ->
// split planes in the chunk order above.  One thread per (chunk, piece-without-plane, lane).
static __global__ void pack_rows_x3_kernel(const float* __restrict__ P, bf16x8* __restrict__ dst, int N, int K, int fc2) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_chunks = fc2 ? K / 32 : N / 32;
    if (idx >= n_chunks * 8 * 64) return;
    const int lane = idx & 63, piece = (idx >> 6) & 7, chunk = idx >> 9;
    const int i = lane & 31, h = lane >> 5;
    f32x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int n, k;
        if (!fc2) {
            n = chunk * 32 + i;
            k = 32 * (piece >> 1) + 16 * (piece & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
            v[j] = P[packed_index(n, k, K)];
        } else {
            const int nt = piece >> 1, s = piece & 1;
            n = 32 * nt + i;
            k = 32 * chunk + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
            const int G = k >> 3, hh = (k >> 2) & 1, e = k & 3, c = G >> 2, g = G & 3;
            v[j] = P[((((size_t)(c * (N >> 5) + (n >> 5)) * 4 + g) * 64) + (hh * 32 + (n & 31))) * 4 + e];
        }
    }
    const Split3 sp = split3(v);
    bf16x8* d = dst + ((size_t)(chunk * 8 + piece) * 3) * 64 + lane;
    d[0] = sp.h;
    d[64] = sp.m;
    d[128] = sp.l;
}

inline int pack_rows_x3(const float* P, bf16x8* dst, int N, int K, int fc2, hipStream_t st) {
    const int n = (fc2 ? K / 32 : N / 32) * 8 * 64;
    pack_rows_x3_kernel<<<(n + 255) / 256, 256, 0, st>>>(P, dst, N, K, fc2);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

// the three planes of weight piece `pc` of the chunk in ring slot `wb` (wb already offset by lane)
__device__ __forceinline__ Split3 ldw3(const bf16x8* wb, int pc) {
    Split3 w;
    w.h = wb[(pc * 3 + 0) * 64];
    w.m = wb[(pc * 3 + 1) * 64];
    w.l = wb[(pc * 3 + 2) * 64];
    return w;
}

// One chunk's eight k-steps with the NEXT k-step's three weight fragments requested before the current k-step's six MFMAs
// (sched_barrier pins that order; the waitcnt pass then emits counted lgkmcnt waits instead of `3 ds_read, lgkmcnt(0), 6 MFMA`
// per k-step): rows -1.5 %, sampler +1.1 % in a same-box A/B (profiles/r05_x3_pingpong_ab.txt).  `hook(ks)` runs between the
// k-steps (the LDS-DMA piece of the next chunk).
template <bool SWAP, typename BOP, typename HOOK>
__device__ __forceinline__ void ktile_x3(const bf16x8* wb, BOP&& bop, f32x16& acc, HOOK&& hook) {
    Split3 w = ldw3(wb, 0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        Split3 wn = w;
        if (ks + 1 < 8) wn = ldw3(wb, ks + 1);
        __builtin_amdgcn_sched_barrier(0);
        acc = SWAP ? mfma_x3(bop(ks), w, acc) : mfma_x3(w, bop(ks), acc);
        hook(ks);
        __builtin_amdgcn_sched_barrier(0);
        w = wn;
    }
}

template <bool DO_MLP, bool DO_QKV>
__global__ __launch_bounds__(256, 2) T2S_X3_KERNEL void dit_rows_x3_kernel(const RowArgsX3 a) {
    extern __shared__ __attribute__((aligned(16))) bf16x8 wring3[];  // [2][X3_CHUNK_UNITS]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    const int n_tiles = a.M >> 5;
    int tile = blockIdx.x * 4 + wave;             // 32-token tile of this wave
    const bool active = tile < n_tiles;           // tail waves compute on a clamped tile, store nothing
    if (!active) tile = n_tiles - 1;
    const int seq = (tile * 32) / NTOK;           // 480 = 15*32: a tile never straddles sequences
    const float* __restrict__ modrow = a.mod + (size_t)seq * MODROW;
    if (a.mod_step != nullptr)      // (spelled out here: a helper without this kernel's target attribute would not inline)
        modrow = a.mod + ((size_t)(*a.mod_step) * a.mod_rows + (seq < a.mod_uncond ? 0 : 1 + a.mod_row0 + (seq - a.mod_uncond))) * MODROW;

    constexpr int N_CHUNKS = (DO_MLP ? 20 : 0) + (DO_QKV ? 12 : 0);
    auto chunk_src = [&](int ci) T2S_X3_KERNEL -> const bf16x8* {
        if constexpr (DO_MLP) {
            if (ci < 4) return a.Wp + (size_t)ci * X3_CHUNK_UNITS;
            if (ci < 20) {
                const int j = ci - 4;
                return ((j & 1) ? a.W2c : a.W1) + (size_t)(j >> 1) * X3_CHUNK_UNITS;
            }
            ci -= 20;
        }
        return a.Wq + (size_t)ci * X3_CHUNK_UNITS;
    };
    // each wave DMAs pieces {wave, wave+4, ..., wave+20} of the chunk
    auto fill = [&](int ci) T2S_X3_KERNEL {
        const bf16x8* src = chunk_src(ci) + lane;
        bf16x8* dst = wring3 + (ci & 1) * X3_CHUNK_UNITS;
#pragma unroll
        for (int p = 0; p < 6; ++p)      // untracked (glds16_asm), like the pieces below: with a TRACKED LDS-DMA pending hipcc turns
            glds16_asm(reinterpret_cast<const f32x4*>(src + (wave + 4 * p) * 64),      // every wait of the prologue into vmcnt(0) --
                       reinterpret_cast<f32x4*>(dst + (wave + 4 * p) * 64));           // one full round trip per constant load
    };

    X3_STAMP_DECL
    // The next chunk's six LDS-DMA pieces are issued BETWEEN the k-steps of the current chunk's MFMA group, one per k-step:
    // an LDS-DMA instruction costs ~100 issue cycles in an MFMA gap against ~150 in front of the group, where nothing covers
    // the wave's vector-memory issue (-DT2S_X3_STAMP put 14 % of a wave's cycles into the six-instruction burst; spread out
    // they cost 10 %: rows -1.0 %, sampler +0.7 % in a same-box A/B, profiles/r05_x3_dma_mix_ab.txt).  Untracked inline asm
    // (glds16_asm) so hipcc keeps its counted lgkmcnt waits for the fragment reads: every chunk therefore ends with an
    // explicit vmcnt wait in front of its barrier.
    auto fill_piece = [&](int ci, int p) T2S_X3_KERNEL {
        const bf16x8* src = chunk_src(ci) + lane + (wave + 4 * p) * 64;
        bf16x8* dst = wring3 + (ci & 1) * X3_CHUNK_UNITS + (wave + 4 * p) * 64;
        glds16_asm(reinterpret_cast<const f32x4*>(src), reinterpret_cast<f32x4*>(dst));
    };
#define X3_FILL_MIX(ci, step) if ((step) >= 1 && (step) <= 6) fill_piece(ci, (step) - 1);
#define X3_DMA_LANDED() asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fill(0);
    // The attention output of the tile is requested FIRST, in front of the constants below (whose loads the compiler waits for
    // one by one before their ds_writes): its round trip then runs under theirs instead of after them.
    f32x4 araw[DO_MLP ? 16 : 1];
    if constexpr (DO_MLP) {
        const f32x4* ar = reinterpret_cast<const f32x4*>(a.ao) + (size_t)tile * 16 * 64 + lane;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
#ifdef T2S_X3_NOPROLOGUE   // TIMING ONLY (results invalid): the bound of a perfect prefetch of the tile's inputs
            araw[g] = f32x4{0.01f * lane, 0.3f, 0.1f * g, -0.7f};
            (void)ar;
#else
            araw[g] = ar[g * 64];
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- per-feature constants in LDS (visible after the first barrier), as in t2s_rows.h ----
    float* cb = reinterpret_cast<float*>(wring3 + 2 * X3_CHUNK_UNITS);
    float* cm = cb + ROWS_CB_FLOATS + wave * ROWS_CM_FLOATS;
    {   // every load first, then every ds_write: written as load -> store pairs each pair cost its own L2 round trip
        const int t0 = threadIdx.x, t1 = t0 + 256;
        float c0 = 0.f, c1 = 0.f, q0 = 0.f, q1 = 0.f;
        f32x4 m0 = {}, m1 = {}, m2 = {}, mq = {};
        if constexpr (DO_MLP) {
            c0 = t0 < 128 ? a.bp[t0] : a.b1[t0 - 128];                  // i = t0 < 256
            c1 = t1 < 384 ? a.b1[t1 - 128] : a.b2[t1 - 384];            // i = t1 in [256, 512)
            const float* src = modrow + a.blk * MODW;
            m0 = *reinterpret_cast<const f32x4*>(src + lane * 4);
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
            *reinterpret_cast<f32x4*>(cm + lane * 4) = m0;
            *reinterpret_cast<f32x4*>(cm + (64 + lane) * 4) = m1;
            *reinterpret_cast<f32x4*>(cm + (128 + lane) * 4) = m2;
        }
        if constexpr (DO_QKV) {
            cb[512 + t0] = q0;
            if (t1 < 384) cb[512 + t1] = q1;
            *reinterpret_cast<f32x4*>(cm + 768 + lane * 4) = mq;
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
            // patchify in the prologue (as t2s_rows.h): patch_emb weight in the (unused) proj / MLP bias slots of LDS, bias in the
            // wave's (unused) MLP adaLN slots; visible after this barrier
            for (int i = threadIdx.x; i < 512; i += 256) cb[i] = a.p_pw[i];
            *reinterpret_cast<f32x4*>(cm + 256 + (lane & 31) * 4) = *reinterpret_cast<const f32x4*>(a.p_pb + (lane & 31) * 4);
            wg_sync();
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
    auto load_x = [&]() T2S_X3_KERNEL {
        const f32x4* xr = reinterpret_cast<const f32x4*>(a.x_in) + (size_t)tile_src * 16 * 64 + lane;
#pragma unroll
        for (int G = 0; G < 16; ++G) {
#ifdef T2S_X3_NOPROLOGUE   // TIMING ONLY (results invalid): the bound of a perfect prefetch of the tile's inputs
            const f32x4 t = {0.25f * lane, 0.5f, -0.125f * G, 1.0f};
            (void)xr;
#else
            const f32x4 t = xr[G * 64];
#endif
#pragma unroll
            for (int e = 0; e < 4; ++e) x[G >> 2][4 * (G & 3) + e] = t[e];
        }
    };
    if (!DO_MLP && !generated) load_x();
    int ci = 0;

    if constexpr (DO_MLP) {
        const float* mb = cm;   // [shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp] of a.blk
        // ---------------- x += gate_msa * (proj(ao) + b): ao planes resident ----------------
        {
            Split3 aop[8];      // k-step ks = features 16 ks .. 16 ks + 15 in the permuted order
            // Order of the prologue's vector-memory operations: chunk 0's DMA, the attention output (top of the kernel), the
            // constants, and LAST the 16 loads of the residual stream -- it is first needed after the four proj chunks, so the
            // first barrier does not wait for it (X3_SYNC_BUT16) and the split of ao runs while it is in flight.  A timing-only
            // build without any prologue load bounded this at +5.3 % of the sampler (profiles/EXPERIMENTS.md 0.11).
            __builtin_amdgcn_sched_barrier(0);
            load_x();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const f32x4 lo = araw[2 * ks], hi = araw[2 * ks + 1];
                const f32x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                aop[ks] = split3(v);
            }
            X3_STAMP(0)
#ifdef T2S_X3_NOPROLOGUE
            X3_SYNC()
#else
            X3_SYNC_BUT16()  // chunk 0 landed, ao consumed; the residual stream may still be on its way
#endif
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                X3_STAMP(5)
                const bf16x8* wb = wring3 + (ci & 1) * X3_CHUNK_UNITS + lane;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                ktile_x3<false>(wb, [&](int ks) T2S_X3_KERNEL -> const Split3& { return aop[ks]; }, acc,
                                [&](int ks) T2S_X3_KERNEL { X3_FILL_MIX(ci + 1, ks) });
                X3_STAMP(1)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bias = ldc4(c_bp, nt, g, half);
                    const f32x4 gate = ldc4(mb + 2 * D, nt, g, half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[nt][4 * g + e] += gate[e] * (acc[4 * g + e] + bias[e]);
                }
                X3_STAMP(2)
                X3_DMA_LANDED()
                X3_SYNC()
                ++ci;
            }
        }
        // ---------------- x += gate_mlp * (fc2(gelu(fc1(mod(LN(x))))) + b2) ----------------
        f32x4* xw = reinterpret_cast<f32x4*>(a.x) + (size_t)tile * 16 * 64 + lane;
        {
            Split3 xmp[8];      // LayerNorm + modulate output as resident planes (the fp32 copy dies here)
            X3_PRIO(2)
            {
                f32x16 xm[4];
                ln_modulate(x, xm, mb + 3 * D, mb + 4 * D, half, 1e-6f);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) xmp[ks] = split3_acc(xm[ks >> 1], ks & 1);
            }
            X3_STAMP(2)
            // park the post-attention residual in HBM for the MLP loop (t2s_rows.h)
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
            X3_PRIO(0)
            X3_STAMP(5)
#pragma unroll 1
            for (int c = 0; c < 8; ++c) {  // 32 hidden units per chunk; ci = 4 + 2c (even) here
                X3_STAMP(5)
                f32x16 hT;
                {
                    const bf16x8* wb = wring3 + lane;  // ci even -> ring slot 0
#pragma unroll
                    for (int r = 0; r < 16; ++r) hT[r] = 0.f;
                    ktile_x3<false>(wb, [&](int ks) T2S_X3_KERNEL -> const Split3& { return xmp[ks]; }, hT,
                                    [&](int ks) T2S_X3_KERNEL { X3_FILL_MIX(ci + 1, ks) });
                    X3_STAMP(1)
                    X3_PRIO(2)   // GELU + split: let this wave's VALU win the issue arbitration over the partner's MFMA stream
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 bias = *reinterpret_cast<const f32x4*>(c_b1 + 32 * c + 8 * g + 4 * half);
#pragma unroll
                        for (int e = 0; e < 4; ++e) hT[4 * g + e] = gelu_tanh_f(hT[4 * g + e] + bias[e]);
                    }
                }
                X3_STAMP(2)
                X3_DMA_LANDED()
                X3_SYNC()
                ++ci;
                X3_STAMP(5)
                {   // fc2 partial over the 32 hidden units of this chunk: pieces (nt, s)
                    const bf16x8* wb = wring3 + X3_CHUNK_UNITS + lane;  // ci odd -> ring slot 1
                    const Split3 h0 = split3_acc(hT, 0), h1 = split3_acc(hT, 1);
                    X3_PRIO(0)
                    X3_STAMP(2)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        acc[nt] = mfma_x3(ldw3(wb, nt * 2 + 0), h0, acc[nt]);
                        if (ci + 1 < N_CHUNKS) { X3_FILL_MIX(ci + 1, 2 * nt) }
                        acc[nt] = mfma_x3(ldw3(wb, nt * 2 + 1), h1, acc[nt]);
                        if (ci + 1 < N_CHUNKS) { X3_FILL_MIX(ci + 1, 2 * nt + 1) }
                    }
                    X3_STAMP(1)
                }
                X3_DMA_LANDED()
                X3_SYNC()
                ++ci;
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bias = ldc4(c_b2, nt, g, half);
                    const f32x4 gate = ldc4(mb + 5 * D, nt, g, half);
                    const f32x4 xo = xw[(nt * 4 + g) * 64];   // the parked residual
                    f32x4 t;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        t[e] = xo[e] + gate[e] * (acc[nt][4 * g + e] + bias[e]);
                        x[nt][4 * g + e] = t[e];
                    }
                    if (active && (DO_QKV || a.out0 == nullptr || a.keep_x)) xw[(nt * 4 + g) * 64] = t;   // final residual stream of this block
                }
            X3_STAMP(2)
        }
    } else {
        X3_STAMP(0)
        X3_SYNC()  // chunk 0 landed
    }

    // ---- fused final layer of the LAST block (transformer.py:182-191): affine LayerNorm (eps 1e-5),
    // Linear 128 -> 4, unpatchify out[s][(2ww+pw)*30 + 2hh+ph] = y[ph*2+pw]; the lane pair of a token holds its
    // 128 features, so everything is lane-local up to one cross-half add per output
    if constexpr (DO_MLP && !DO_QKV) {
        if (a.out0 != nullptr) {
            float s1 = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    s1 += (x[nt][4 * g] + x[nt][4 * g + 1]) + (x[nt][4 * g + 2] + x[nt][4 * g + 3]);
            s1 += xhalf(s1);
            const float mean = s1 * (1.0f / 128.0f);
            float s2 = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[nt][4 * g + e] -= mean;
                    s2 += (x[nt][4 * g] * x[nt][4 * g] + x[nt][4 * g + 1] * x[nt][4 * g + 1]) +
                          (x[nt][4 * g + 2] * x[nt][4 * g + 2] + x[nt][4 * g + 3] * x[nt][4 * g + 3]);
                }
            s2 += xhalf(s2);
            const float rstd = rsqrtf(s2 * (1.0f / 128.0f) + 1e-5f);
            float fa[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = 32 * nt + 8 * g + 4 * half;
                    const f32x4 gam = *reinterpret_cast<const f32x4*>(a.f_lnw + col);
                    const f32x4 bet = *reinterpret_cast<const f32x4*>(a.f_lnb + col);
                    f32x4 y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = (x[nt][4 * g + e] * rstd) * gam[e] + bet[e];
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(a.f_ow + p * D + col);
                        fa[p] += (y.x * w.x + y.y * w.y) + (y.z * w.z + y.w * w.w);
                    }
                }
#pragma unroll
            for (int p = 0; p < 4; ++p) fa[p] += xhalf(fa[p]);
            if (active) {   // lane half 0 writes patch outputs p = 0,1; half 1 writes p = 2,3
                const int n = (tile - seq * (NTOK / 32)) * 32 + (lane & 31);
                const int hh = n >> 5, ww = n & 31;
                float* dst = (seq < a.split) ? a.out0 + (size_t)seq * LAT : a.out1 + (size_t)(seq - a.split) * LAT;
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const int p = 2 * half + q2;
                    dst[(2 * ww + (p & 1)) * LATW + 2 * hh + (p >> 1)] = (half ? fa[2 + q2] : fa[q2]) + a.f_ob[p];
                }
            }
        }
    }
    if constexpr (DO_QKV) {
        Split3 xmp[8];      // LayerNorm + modulate output as resident planes
        X3_PRIO(2)
        {
            f32x16 xm[4];
            ln_modulate(x, xm, cm + 768, cm + 768 + D, half, 1e-6f);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) xmp[ks] = split3_acc(xm[ks >> 1], ks & 1);
        }
        const int tile_in_seq = tile - seq * (NTOK / 32);
        X3_PRIO(0)
        X3_STAMP(2)
#pragma unroll 1
        for (int t = 0; t < 12; ++t) {  // output tile t = which*4 + head
            X3_STAMP(5)
            const bf16x8* wb = wring3 + (ci & 1) * X3_CHUNK_UNITS + lane;
            const int which = t >> 2, head = t & 3;
            const size_t head_tile = ((size_t)seq * NH + head) * (NTOK / 32) + tile_in_seq;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (which < 2) {
                // q / k tile, transposed product: lane = token, registers = features d
                ktile_x3<false>(wb, [&](int ks) T2S_X3_KERNEL -> const Split3& { return xmp[ks]; }, acc,
                                [&](int ks) T2S_X3_KERNEL { if (ci + 1 < N_CHUNKS) { X3_FILL_MIX(ci + 1, ks) } });
                X3_STAMP(1)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bias = *reinterpret_cast<const f32x4*>(c_bq + 32 * t + 8 * g + 4 * half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[4 * g + e] += bias[e];
                }
                if (active) {
                    if (which == 0) {
                        f32x4* dst = reinterpret_cast<f32x4*>(a.q) + head_tile * 4 * 64 + lane;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 o = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                            dst[g * 64] = o;
                        }
                    } else {
                        bf16x8* d3 = reinterpret_cast<bf16x8*>(a.k3) + head_tile * X3_TILE_UNITS + lane;
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) {
                            const Split3 sp = split3_acc(acc, s2);
                            d3[(0 + s2) * 64] = sp.h;
                            d3[(2 + s2) * 64] = sp.m;
                            d3[(4 + s2) * 64] = sp.l;
                        }
                    }
                }
            } else {
                // v tile with the MFMA operands swapped: lane = feature d, registers = keys
                const float bias = c_bq[32 * t + (lane & 31)];
                ktile_x3<true>(wb, [&](int ks) T2S_X3_KERNEL -> const Split3& { return xmp[ks]; }, acc,
                               [&](int ks) T2S_X3_KERNEL { if (ci + 1 < N_CHUNKS) { X3_FILL_MIX(ci + 1, ks) } });
                X3_STAMP(1)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += bias;
                if (active) {
                    bf16x8* d3 = reinterpret_cast<bf16x8*>(a.v3) + head_tile * X3_TILE_UNITS + lane;
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const Split3 sp = split3_acc(acc, s2);
                        d3[(0 + s2) * 64] = sp.h;
                        d3[(2 + s2) * 64] = sp.m;
                        d3[(4 + s2) * 64] = sp.l;
                    }
                }
            }
            // counted wait + raw barrier: the next chunk's 6 DMA pieces must have landed; the q (4) or
            // k / v plane (6) stores issued after them stay in flight.  Tail waves store nothing.
            X3_STAMP(2)
            if (!active)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (which >= 1)
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            X3_STAMP(3)
            __builtin_amdgcn_s_barrier();
            X3_STAMP(4)
            ++ci;
        }
    }
#ifdef T2S_X3_STAMP
    X3_STAMP(6)
    if (a.stamp != nullptr && blockIdx.x < 256 && lane == 0) {
        unsigned long long* d = a.stamp + ((size_t)blockIdx.x * 4 + wave) * 8;
        for (int k = 0; k < 7; ++k) d[k] = st_acc[k];
        d[7] = st_last - st_t0;
    }
#endif
}

template <bool DO_MLP, bool DO_QKV>
inline int dit_rows_x3_init() {
    T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dit_rows_x3_kernel<DO_MLP, DO_QKV>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, ROWS_X3_LDS_BYTES));
    return T2S_OK;
}

template <bool DO_MLP, bool DO_QKV>
inline int launch_dit_rows_x3(const RowArgsX3& a, hipStream_t st) {
    if (a.M <= 0 || a.M % 32 != 0) {
        set_error("dit_rows_x3: M=%d must be a positive multiple of 32", a.M);
        return T2S_E_INVALID;
    }
    const int tiles = a.M / 32;
#ifdef T2S_X3_STAMP
    static unsigned long long* buf = nullptr;
    static int calls = 0;
    if (!buf) {
        T2S_HIP_CHECK(hipMalloc((void**)&buf, 256 * 4 * 8 * sizeof(unsigned long long)));
        T2S_HIP_CHECK(hipMemset(buf, 0, 256 * 4 * 8 * sizeof(unsigned long long)));
    }
    RowArgsX3 a2 = a;
    a2.stamp = buf;
    dit_rows_x3_kernel<DO_MLP, DO_QKV><<<(tiles + 3) / 4, 256, ROWS_X3_LDS_BYTES, st>>>(a2);
    T2S_LAUNCH_CHECK();
    if (++calls == 40 && tiles >= 4096) {      // one dump per instance, well after warm-up, at a chip-filling launch
        static unsigned long long host[256 * 4 * 8];
        T2S_HIP_CHECK(hipStreamSynchronize(st));
        T2S_HIP_CHECK(hipMemcpy(host, buf, sizeof(host), hipMemcpyDeviceToHost));
        double sum[8] = {};
        const int n = (tiles + 3) / 4 < 256 ? (tiles + 3) / 4 : 256;
        for (int i = 0; i < n * 4; ++i)
            for (int k = 0; k < 8; ++k) sum[k] += (double)host[i * 8 + k];
        fprintf(stderr, "x3_stamp <%d,%d> tiles %d: cycles per wave (s_memtime, 100 MHz ticks x? see tools/x3_stamp.sh) total %.0f | prologue %.0f mfma %.0f valu %.0f "
                        "vmcnt %.0f barrier %.0f issue %.0f epilogue %.0f\n", (int)DO_MLP, (int)DO_QKV, tiles, sum[7] / (n * 4), sum[0] / (n * 4),
                sum[1] / (n * 4), sum[2] / (n * 4), sum[3] / (n * 4), sum[4] / (n * 4), sum[5] / (n * 4), sum[6] / (n * 4));
    }
    return T2S_OK;
#else
    dit_rows_x3_kernel<DO_MLP, DO_QKV><<<(tiles + 3) / 4, 256, ROWS_X3_LDS_BYTES, st>>>(a);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
#endif
}

}  // namespace t2s
