// EXPERIMENT (round 1, NOT part of the build): persistent one-wave-per-SIMD row chain.
// Result on MI355X: 814 / 571 / 242 us for the three variants against 604 / 390 / 225 us for the
// 2-workgroups-per-CU kernel in t2ms_amd/csrc/t2s_rows.h.  Ablating the DMA waits and the chunk
// barriers recovered only 6 %; with a single wave per SIMD the LDS-DMA issue cost (~150 cycles x 4
// per chunk), the VGPR<->AGPR shuffling of a 512-register allocation and the LayerNorm / GELU VALU
// are all exposed, which the second wave of the shipped kernel hides.  Kept as a record.
// Persistent, one-wave-per-SIMD version of the DiT row chain (see t2s_rows.h for the math and
// the register-resident transposed formulation).
//
// Why: in-kernel stamps (tools/probe_rows.py) showed that a wave of the 2-workgroups-per-CU
// kernel spends 38 % of its life outside MFMA/VALU work even when it has the SIMD to itself:
// a 16 k-cycle prologue per 32-token tile (weight DMA + activation loads + barrier), a spill-
// ridden LayerNorm phase, and ~1.5 k cycles per weight chunk of exposed LDS latency + barrier.
// On gfx950 f32 MFMA and VALU share the lanes, so a second wave per SIMD cannot hide compute
// behind compute anyway -- it only hid those stalls, at the price of halving the register file.
//
// Here ONE 4-wave workgroup per CU stays resident (one wave per SIMD, all 512 VGPRs):
//   * it walks its tiles in a loop; the next tile's x / attention rows (128 VGPRs) and adaLN
//     vectors are prefetched during the current tile's qkv stage;
//   * the weight ring (3 x 16 KiB, LDS-DMA two chunks ahead) runs continuously across tiles;
//   * each chunk's 16 fragments are read from LDS into registers ONE CHUNK AHEAD (64 VGPRs,
//     two alternating sets), right after the chunk barrier, so both the LDS latency and the
//     barrier hide behind the current chunk's 64 MFMAs;
//   * nothing spills, the post-attention residual is never parked in HBM;
//   * biases enter through the MFMA C operand (accumulator initialised from LDS) -- no VALU.
#pragma once
#include "t2s_rows.h"

namespace t2s {

constexpr int ROWSP_SLOTS = 3;
constexpr int ROWSP_LDS_BYTES = ROWSP_SLOTS * ROWS_CHUNK_F4 * 16 + (ROWS_CB_FLOATS + 4 * ROWS_CM_FLOATS) * 4;

// counted wait: all but the VM youngest vector-memory operations of this wave are complete
template <int VM>
__device__ __forceinline__ void wait_vm() {
    static_assert(VM >= 0 && VM < 64, "vmcnt is a 6-bit field");
#if defined(T2S_EXP) && (T2S_EXP & 128)
    return;   // diagnostic: never wait for the weight DMA (results are wrong)
#endif
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");
}

template <bool DO_MLP, bool DO_QKV>
__global__ __launch_bounds__(256, 1) void dit_rows_p_kernel(const RowArgs a) {
    extern __shared__ __attribute__((aligned(16))) f32x4 wring[];  // [3][1024] | cb | cm x 4
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    float* cb = reinterpret_cast<float*>(wring + ROWSP_SLOTS * ROWS_CHUNK_F4);
    float* cm = cb + ROWS_CB_FLOATS + wave * ROWS_CM_FLOATS;
    const float* c_bp = cb;
    const float* c_b1 = cb + 128;
    const float* c_b2 = cb + 384;
    const float* c_bq = cb + 512;

    constexpr int NCH = (DO_MLP ? 20 : 0) + (DO_QKV ? 12 : 0);   // chunks per tile (even)
    const int n_tiles = a.M >> 5;
    const int stride = gridDim.x * 4;
    const int n_iter = (n_tiles + stride - 1) / stride;
    const int total_chunks = n_iter * NCH;

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
    // DMA bookkeeping (wave-uniform scalars): next chunk to fetch, its index inside the tile, its slot
    int f_gc = 0, f_ci = 0, f_slot = 0;
    auto fill_next = [&]() {
        const int ci = f_gc < total_chunks ? f_ci : 0;       // past the end: harmless re-fetch of chunk 0
        const f32x4* src = chunk_src(ci) + lane;
        f32x4* dst = wring + f_slot * ROWS_CHUNK_F4;
#pragma unroll
        for (int p = 0; p < 4; ++p) glds16(src + (wave + 4 * p) * 64, dst + (wave + 4 * p) * 64);
        ++f_gc;
        f_ci = (f_ci + 1 == NCH) ? 0 : f_ci + 1;
        f_slot = (f_slot + 1 == ROWSP_SLOTS) ? 0 : f_slot + 1;
    };
    int r_slot = 0;   // slot of the chunk whose fragments are loaded NEXT
    // chunk boundary: make the next chunk visible, keep the DMA two ahead, fetch its fragments
    auto advance = [&](f32x4 (&wn)[16], auto vm_tag) {
        wait_vm<decltype(vm_tag)::value>();
#if !defined(T2S_EXP) || !(T2S_EXP & 256)
        __builtin_amdgcn_s_barrier();
#endif
        fill_next();
        const f32x4* wb = wring + r_slot * ROWS_CHUNK_F4 + lane;
#pragma unroll
        for (int p = 0; p < 16; ++p) wn[p] = wb[p * 64];
        r_slot = (r_slot + 1 == ROWSP_SLOTS) ? 0 : r_slot + 1;
    };
    using VM0 = std::integral_constant<int, 0>;
    using VM4 = std::integral_constant<int, 4>;
    using VM16 = std::integral_constant<int, 16>;
    constexpr int N_PREFETCH = 16 + (DO_MLP ? 16 + 3 : 0) + (DO_QKV ? 1 : 0);   // loads of load_tile()
    using VMPF = std::integral_constant<int, 4 + N_PREFETCH>;

    // ---- one-time: biases -> LDS, first two weight chunks in flight
    if constexpr (DO_MLP) {
        for (int i = threadIdx.x; i < 512; i += 256)
            cb[i] = i < 128 ? a.bp[i] : (i < 384 ? a.b1[i - 128] : a.b2[i - 384]);
    }
    if constexpr (DO_QKV) {
        for (int i = threadIdx.x; i < 384; i += 256) cb[512 + i] = a.bq[i];
    }
    fill_next();
    fill_next();

    // ---- first tile: activations and adaLN vectors straight from memory
    int tile = blockIdx.x * 4 + wave;
    bool active = tile < n_tiles;
    if (!active) tile = n_tiles - 1;
    f32x4 xf[16], af[16], mf[4];   // x rows, attention rows (fragments), adaLN vectors of the tile's sequence
    auto load_tile = [&](int t, f32x4 (&xo)[16], f32x4 (&ao)[16], f32x4 (&mo)[4]) {
        const f32x4* xr = reinterpret_cast<const f32x4*>(a.x) + (size_t)t * 16 * 64 + lane;
#pragma unroll
        for (int G = 0; G < 16; ++G) xo[G] = xr[G * 64];
        if constexpr (DO_MLP) {
            const f32x4* ar = reinterpret_cast<const f32x4*>(a.ao) + (size_t)t * 16 * 64 + lane;
#pragma unroll
            for (int G = 0; G < 16; ++G) ao[G] = ar[G * 64];
        }
        const float* modrow = a.mod + (size_t)((t * 32) / NTOK) * MODROW;
        if constexpr (DO_MLP) {
            const float* src = modrow + a.blk * MODW;
#pragma unroll
            for (int i = 0; i < 3; ++i) mo[i] = *reinterpret_cast<const f32x4*>(src + (i * 64 + lane) * 4);
        }
        if constexpr (DO_QKV) mo[3] = *reinterpret_cast<const f32x4*>(modrow + a.qkv_blk * MODW + lane * 4);
    };
    load_tile(tile, xf, af, mf);

    f32x4 wa[16], wb_[16];
    // chunk 0 landed -> visible; fragments of chunk 0 into wa
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (also completes the tile loads above)
    __builtin_amdgcn_s_barrier();
    {
        const f32x4* wb = wring + lane;
#pragma unroll
        for (int p = 0; p < 16; ++p) wa[p] = wb[p * 64];
        r_slot = 1;
    }

#pragma unroll 1
    for (int it = 0; it < n_iter; ++it) {
        // per-wave adaLN vectors of this tile's sequence -> LDS (read back by this wave only)
        if constexpr (DO_MLP) {
#pragma unroll
            for (int i = 0; i < 3; ++i) *reinterpret_cast<f32x4*>(cm + (i * 64 + lane) * 4) = mf[i];
        }
        if constexpr (DO_QKV) *reinterpret_cast<f32x4*>(cm + 768 + lane * 4) = mf[3];
        const int seq = (tile * 32) / NTOK;
        const int next_tile_raw = tile + stride;
        const bool have_next = it + 1 < n_iter;
        const int next_tile = next_tile_raw < n_tiles ? next_tile_raw : n_tiles - 1;
        f32x4 xn[16], an[16], mn[4];

        f32x16 x[4];
#pragma unroll
        for (int G = 0; G < 16; ++G)
#pragma unroll
            for (int e = 0; e < 4; ++e) x[G >> 2][4 * (G & 3) + e] = xf[G][e];

        if constexpr (DO_MLP) {
            const float* mb = cm;
            // ---------------- x += gate_msa * (proj(ao) + b): 4 chunks ----------------
            auto proj_chunk = [&](int nt, const f32x4 (&w)[16]) {
                f32x16 acc;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b = ldc4(c_bp, nt, g, half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[4 * g + e] = b[e];
                }
#pragma unroll
                for (int G = 0; G < 16; ++G)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = mfma32(w[G][e], af[G][e], acc);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 gate = ldc4(mb + 2 * D, nt, g, half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[nt][4 * g + e] += gate[e] * acc[4 * g + e];
                }
            };
            advance(wb_, VM0{}); proj_chunk(0, wa);
            advance(wa, VM0{});  proj_chunk(1, wb_);
            advance(wb_, VM0{}); proj_chunk(2, wa);
            advance(wa, VM0{});  proj_chunk(3, wb_);

            // ---------------- x += gate_mlp * (fc2(gelu(fc1(mod(LN(x))))) + b2): 16 chunks ----------------
            f32x16 xm[4];
            ln_modulate(x, xm, mb + 3 * D, mb + 4 * D, half, 1e-6f);
            f32x16 acc[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b = ldc4(c_b2, nt, g, half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[nt][4 * g + e] = b[e];
                }
#pragma unroll 1
            for (int c = 0; c < 8; ++c) {  // 32 hidden units per chunk pair
                advance(wb_, VM0{});       // fc1 chunk c is in wa
                f32x16 hT;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(c_b1 + 32 * c + 8 * g + 4 * half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) hT[4 * g + e] = b[e];
                }
#pragma unroll
                for (int G = 0; G < 16; ++G)
#pragma unroll
                    for (int e = 0; e < 4; ++e) hT = mfma32(wa[G][e], xm[G >> 2][4 * (G & 3) + e], hT);
#pragma unroll
                for (int r = 0; r < 16; ++r) hT[r] = gelu_tanh_f(hT[r]);
                advance(wa, VM0{});        // fc2 fragments [nt][g] of chunk c are in wb_
                if constexpr (!DO_QKV) {
                    if (c == 5 && have_next) load_tile(next_tile, xn, an, mn);   // a chunk old at the next wait
                }
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[nt] = mfma32(wb_[nt * 4 + g][e], hT[4 * g + e], acc[nt]);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 gate = ldc4(mb + 5 * D, nt, g, half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[nt][4 * g + e] += gate[e] * acc[nt][4 * g + e];
                }
            // ---------------- residual stream of this block -> HBM (16 stores) ----------------
            if (active) {
                f32x4* xw = reinterpret_cast<f32x4*>(a.x) + (size_t)tile * 16 * 64 + lane;
#pragma unroll
                for (int G = 0; G < 16; ++G) {
                    f32x4 t;
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = x[G >> 2][4 * (G & 3) + e];
                    xw[G * 64] = t;
                }
            }
        }

        if constexpr (DO_QKV) {
            f32x16 xm[4];
            ln_modulate(x, xm, cm + 768, cm + 768 + D, half, 1e-6f);
            const int tile_in_seq = tile - seq * (NTOK / 32);
            auto qkv_chunk = [&](int t, const f32x4 (&w)[16]) {
                const int which = t >> 2, head = t & 3;
                float* base = which == 0 ? a.q : (which == 1 ? a.k : a.v);
                f32x4* dst = reinterpret_cast<f32x4*>(base) +
                             (((size_t)seq * NH + head) * (NTOK / 32) + tile_in_seq) * 4 * 64 + lane;
                f32x16 acc;
                if (which < 2) {
                    // q / k tile, transposed product: lane = token, registers = features d; bias via C
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 b = *reinterpret_cast<const f32x4*>(c_bq + 32 * t + 8 * g + 4 * half);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[4 * g + e] = b[e];
                    }
#pragma unroll
                    for (int G = 0; G < 16; ++G)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc = mfma32(w[G][e], xm[G >> 2][4 * (G & 3) + e], acc);
                } else {
                    // v tile with swapped operands: lane = feature d, registers = tokens -> V^T fragments
                    const float b = c_bq[32 * t + (lane & 31)];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = b;
#pragma unroll
                    for (int G = 0; G < 16; ++G)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc = mfma32(xm[G >> 2][4 * (G & 3) + e], w[G][e], acc);
                }
                if (active) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 o = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                        dst[g * 64] = o;
                    }
                }
            };
            // the first boundary follows the 16 residual-stream stores (if any), later ones 4 q/k/v stores
            if (DO_MLP && active) advance(wb_, VM16{}); else advance(wb_, VM0{});
            qkv_chunk(0, wa);
            // next tile's activations: issued behind this chunk's DMA, a whole chunk old at the next wait
            if (have_next) load_tile(next_tile, xn, an, mn);
            // chunk 2's DMA (issued above, before the 4 stores and the prefetch) must have landed; the
            // stores and the prefetch loads stay in flight
            if (active && have_next) advance(wa, VMPF{}); else if (active) advance(wa, VM4{}); else advance(wa, VM0{});
            qkv_chunk(1, wb_);
#pragma unroll 1
            for (int t = 2; t < 12; t += 2) {
                if (active) advance(wb_, VM4{}); else advance(wb_, VM0{});
                qkv_chunk(t, wa);
                if (active) advance(wa, VM4{}); else advance(wa, VM0{});
                qkv_chunk(t + 1, wb_);
            }
        }
        // ---- switch to the prefetched tile
        if (have_next) {
#pragma unroll
            for (int G = 0; G < 16; ++G) { xf[G] = xn[G]; if (DO_MLP) af[G] = an[G]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) mf[i] = mn[i];
        }
        tile = next_tile;
        active = next_tile_raw < n_tiles;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the trailing (unused) DMAs
}

template <bool DO_MLP, bool DO_QKV>
inline int launch_dit_rows_p(const RowArgs& a, int n_cu, hipStream_t st) {
    dit_rows_p_kernel<DO_MLP, DO_QKV><<<n_cu, 256, ROWSP_LDS_BYTES, st>>>(a);
    T2S_LAUNCH_CHECK();
    return T2S_OK;
}

template <bool DO_MLP, bool DO_QKV>
inline int dit_rows_p_init() {
    T2S_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dit_rows_p_kernel<DO_MLP, DO_QKV>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, ROWSP_LDS_BYTES));
    return T2S_OK;
}

}  // namespace t2s
