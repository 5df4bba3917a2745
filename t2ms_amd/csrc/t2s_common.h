// Internal helpers shared by the HIP translation units of libt2s_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/t2s.h"

namespace t2s {

void set_error(const char* fmt, ...);

#define T2S_HIP_CHECK(expr)                                                                    \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            ::t2s::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                             __LINE__);                                                        \
            return T2S_E_HIP;                                                                  \
        }                                                                                      \
    } while (0)

#define T2S_REQUIRE(cond, ...)                                                                 \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            ::t2s::set_error(__VA_ARGS__);                                                     \
            return T2S_E_INVALID;                                                              \
        }                                                                                      \
    } while (0)

#define T2S_LAUNCH_CHECK()                                                                     \
    do {                                                                                       \
        hipError_t _e = hipGetLastError();                                                     \
        if (_e != hipSuccess) {                                                                \
            ::t2s::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),        \
                             __FILE__, __LINE__);                                              \
            return T2S_E_HIP;                                                                  \
        }                                                                                      \
    } while (0)

// The C ABI carries pointers without sizes.  Where the library reads a KNOWN number of bytes behind a caller's pointer
// (weights at create / update), it first asks HIP for the extent of the allocation the pointer lies in: a buffer that ends
// before those bytes do is T2S_E_INVALID with `what` in the message, never an out-of-bounds read (round 4's fault).  A pointer
// HIP cannot place (not from hipMalloc) is passed through: nothing known, nothing refused.
inline int check_device_extent(const void* p, size_t bytes, const char* what) {
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) != hipSuccess) {
        (void)hipGetLastError();
        return T2S_OK;
    }
    const size_t left = (size_t)((const char*)base + size - (const char*)p);
    if (left < bytes) {
        set_error("%s needs %zu bytes but its device allocation ends after %zu", what, bytes, left);
        return T2S_E_INVALID;
    }
    return T2S_OK;
}

constexpr int D = T2S_D_MODEL;      // 128
constexpr int NTOK = T2S_N_TOK;     // 480
constexpr int NH = T2S_N_HEADS;     // 4
constexpr int DH = T2S_HEAD_DIM;    // 32
constexpr int NBLK = T2S_N_BLOCKS;  // 4
constexpr int LATC = T2S_LAT_C;     // 64
constexpr int LATW = T2S_LAT_W;     // 30
constexpr int LAT = T2S_LAT_ELEMS;  // 1920
constexpr int MODW = 6 * D;         // 768 adaLN outputs per block
constexpr int MODROW = NBLK * MODW; // 3072 per sequence: [blk][shift1,scale1,gate1,shift2,scale2,gate2][128]

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// v_mfma_f32_32x32x2_f32: D(32x32) += A(32x2) * B(2x32); lane l supplies A[l&31][l>>5] and
// B[l>>5][l&31]; result register r of lane l is D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// First MFMA of an accumulation chain whose C operand must SURVIVE (the sticky softmax reference, 16 registers per
// query tile, reused by every key block): d = a.b + c with d in registers of its own.  hipcc always selects the tied
// form (vdst == src2) for the builtin and copies c first -- 16 v_mov_b64 per key block and tile pair in the attention
// loops; the instruction itself takes any src2.  The builtin MFMAs that follow accumulate in place on d
// (vdst == src2 exactly overlapped: back to back, no software wait states).
__device__ __forceinline__ f32x16 mfma32_from(float a, float b, const f32x16& c) {
    f32x16 d;
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ float xhalf(float v) {  // value held by lane ^ 32
    return __shfl_xor(v, 32, 64);
}

// Weight packing for the B operand (W is a torch Linear weight (N,K) row-major, the GEMM is
// x @ W^T).  For n-tile nt (32 outputs) and k-group g (8 inputs) lane l owns the float4
//   W[32*nt + (l&31)][8*g + 4*(l>>5) + 0..3]
// stored at packed[((nt * (K/8)) + g) * 64 + l] so that one wave-level load is a contiguous
// 1 KiB.  MFMA e (0..3) of the group then contracts k-pair {8g+e, 8g+4+e}; the A operand
// uses the same assignment (float4 at A[row][8g + 4*(l>>5)]).
__host__ __device__ inline size_t packed_index(int n, int k, int K) {
    int nt = n >> 5, j = n & 31, g = k >> 3, h = (k >> 2) & 1, e = k & 3;
    return ((((size_t)nt * (K >> 3) + g) * 64) + (h * 32 + j)) * 4 + e;
}

// "Fragment-major" activation layout used BETWEEN the library's own kernels (never at the C ABI):
// a (rows, C) fp32 matrix with rows % 32 == 0 and C % 8 == 0 is stored as
//   [tile = row/32][G = col/8][lane = 32*((col/4)&1) + row%32][e = col%4]
// i.e. exactly one MFMA operand/accumulator fragment (float4 per lane) per 1 KiB, so every
// wave-level load/store of a token tile is one contiguous 1 KiB transaction.
__host__ __device__ inline size_t frag_index(int row, int col, int C) {
    const int tile = row >> 5, i = row & 31, G = col >> 3, h = (col >> 2) & 1, e = col & 3;
    return ((((size_t)tile * (C >> 3) + G) * 64) + (h * 32 + i)) * 4 + e;
}

// LDS-DMA: 64 lanes x 16 B from per-lane global addresses to a wave-uniform LDS base (+16*lane)
__device__ __forceinline__ void glds16(const f32x4* gsrc_lane, f32x4* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)gsrc_lane,
        (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// The same LDS-DMA hidden from hipcc's s_waitcnt bookkeeping (inline asm; M0 = wave-uniform LDS byte address, saved and
// restored inside the statement: it is compiler-reserved).  Why: while a TRACKED LDS-DMA is pending hipcc treats lgkmcnt
// as out of order and turns every counted wait of the ds_read -> MFMA pipelines into `s_waitcnt lgkmcnt(0)`, i.e. it
// also waits for the fragment it has just requested (seen in the ISA of t2s_rows.h: lgkmcnt(1) in the one k-loop that
// runs with no DMA in flight, lgkmcnt(0) everywhere else).  The caller owns the synchronisation: `s_waitcnt vmcnt(N)`
// of its own, then a barrier, before any ds_read of the destination (__syncthreads() alone does NOT wait for it).
__device__ __forceinline__ void glds16_asm(const f32x4* gsrc_lane, f32x4* lds_wave_base) {
    unsigned keep;
    const unsigned dst = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds_wave_base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc_lane), "s"(dst)
                 : "memory");
}

}  // namespace t2s
