// "bf16x3" arithmetic: fp32-accurate products on the bf16 matrix cores (opt-in, T2S_MATH_BF16X3).
//
// An fp32 value p is split into three bf16 terms  p = h + m + l  (h = rn_bf16(p), m = rn_bf16(p - h),
// l = rn_bf16(p - h - m); the two subtractions are exact in fp32, the last rounding loses < 2^-25 |p|).
// A product a.b is evaluated as the six bf16 x bf16 products whose weight is >= 2^-16:
//     ah.bh + ah.bm + am.bh + ah.bl + al.bh + am.bm
// (each exact in the fp32 accumulator: 8 x 8 significant bits), dropping am.bl, al.bm, al.bl
// <= 3 * 2^-24 |a.b|, i.e. the rounding level of an fp32 multiply.  Six v_mfma_f32_32x32x16_bf16
// (32 cycles, 16-deep) replace eight v_mfma_f32_32x32x2_f32 (64 cycles, 2-deep): 2.67x fewer matrix
// cycles at the same accuracy, and bf16 MFMAs co-issue with VALU work, which the f32 form does not.
#pragma once
#include "t2s_bf16.h"

// Packed fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) do NOT overlap with bf16 MFMAs on gfx950:
// a wave issuing them while its SIMD partner streams MFMAs takes the SUM of both times, every other VALU
// instruction hides behind the matrix pipe (tools/ubench_valu.hip: 8 MFMA | 48 x v_pk_add_f32: 214 ns together vs
// 112 / 108 alone; v_fma_f32: 135 vs 114 / 105).  The bf16x3 kernels live on that overlap, so they are compiled
// without packed fp32 (T2S_X3_KERNEL on the __global__ function; the inlined helpers follow the kernel).
#if defined(__HIP_DEVICE_COMPILE__)
#define T2S_X3_KERNEL __attribute__((target("no-packed-fp32-ops")))
#else
#define T2S_X3_KERNEL
#endif

namespace t2s {

// __syncthreads() spelled out: the HIP header's inline function is not inlined into a kernel whose target features
// differ from the default (it became a call), the builtins are
__device__ __forceinline__ void wg_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

struct Split3 {   // the three bf16 planes of one 8-element MFMA operand fragment
    bf16x8 h, m, l;
};

// the fp32 values of a packed bf16x8, by bit operations on the packed words (hipcc turns
// convertvector(bf16 -> f32) after convertvector(f32 -> bf16) into one v_cvt PER ELEMENT plus a
// shift, doubling the conversions; low half: << 16, high half: & 0xffff0000)
__device__ __forceinline__ f32x8 widen8(bf16x8 b) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 w = __builtin_bit_cast(u32x4, b);
    f32x8 f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(w[i] << 16);
        f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
    return f;
}

__device__ __forceinline__ Split3 split3(f32x8 v) {
    Split3 s;
    s.h = __builtin_convertvector(v, bf16x8);
    const f32x8 r = v - widen8(s.h);
    s.m = __builtin_convertvector(r, bf16x8);
    const f32x8 r2 = r - widen8(s.m);
    s.l = __builtin_convertvector(r2, bf16x8);
    return s;
}

// registers 8s..8s+7 of a 32x32 accumulator as a split operand fragment (see acc_frag: element j of
// lane half h is accumulator row 16s + 8(j>>2) + 4h + (j&3))
__device__ __forceinline__ Split3 split3_acc(const f32x16& c, int s) {
    const f32x8 v = {c[8 * s + 0], c[8 * s + 1], c[8 * s + 2], c[8 * s + 3],
                     c[8 * s + 4], c[8 * s + 5], c[8 * s + 6], c[8 * s + 7]};
    return split3(v);
}

// acc += a . b to fp32 accuracy (smallest terms first)
__device__ __forceinline__ f32x16 mfma_x3(const Split3& a, const Split3& b, f32x16 acc) {
    acc = mfma16(a.m, b.m, acc);
    acc = mfma16(a.l, b.h, acc);
    acc = mfma16(a.h, b.l, acc);
    acc = mfma16(a.m, b.h, acc);
    acc = mfma16(a.h, b.m, acc);
    acc = mfma16(a.h, b.h, acc);
    return acc;
}

// Split K / V^T operand planes of one (sequence, head), written by the row-chain kernel's qkv
// epilogue and consumed by attn_fwd_x3_kernel, in units of 16 bytes (one lane's 8 bf16):
//     [(bh * 15 + tile) * 6 + plane * 2 + s][lane]         plane 0/1/2 = h/m/l, s = k-step
// i.e. 6 KiB per 32-token tile, each 1 KiB piece one LDS-DMA instruction.
constexpr int X3_TILE_UNITS = 6 * 64;

}  // namespace t2s
