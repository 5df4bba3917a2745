// Microbenchmark: issue cost of the VALU instructions the bf16x3 kernels are made of (gfx950), alone and in the
// gaps of bf16 MFMAs.  Each wave runs `iters` iterations of NM x [one v_mfma_f32_32x32x16_bf16 (if MF), NV
// independent instances of the instruction]; 1, 2 or 4 waves per SIMD.  Reports cycles per iteration per wave.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/bin/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { OP_FMA, OP_SUB, OP_EXP, OP_CVTPK, OP_PKADD, OP_PKMUL, OP_AND, OP_LSHL, OP_MAX3, OP_MOV, OP_RCP, OP_DOT2C, N_OPS };
static const char* op_name[N_OPS] = {"v_fma_f32", "v_sub_f32", "v_exp_f32", "v_cvt_pk_bf16_f32", "v_pk_add_f32", "v_pk_mul_f32",
                                     "v_and_b32", "v_lshlrev_b32", "v_max3_f32", "v_mov_b32", "v_rcp_f32", "v_dot2c_f32_bf16"};

template <int OP>
__device__ __forceinline__ void one(float& a, float& b, f32x2& p, float c) {
    if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(c));
    if (OP == OP_SUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(c));
    if (OP == OP_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
    if (OP == OP_CVTPK) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == OP_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(p));
    if (OP == OP_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(p));
    if (OP == OP_AND) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(a));
    if (OP == OP_LSHL) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(a));
    if (OP == OP_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (OP == OP_MOV) asm volatile("v_mov_b32 %0, %0" : "+v"(a));
    if (OP == OP_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a));
    if (OP == OP_DOT2C) asm volatile("v_dot2c_f32_bf16 %0, -1.0, %1" : "+v"(a) : "v"(b));   // a += (-1, 0) . (b.lo, b.hi): the x3 split's residual
}

template <int OP, int MF, int NM, int NV>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float a0) {
    extern __shared__ float pad[];
    f32x16 acc[2];
    for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float v[8], w[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { v[i] = a0 + i + threadIdx.x * 1e-6f; w[i] = 0.25f * i; p[i] = f32x2{v[i], w[i]}; }
    bf16x8 ab;
    for (int i = 0; i < 8; ++i) ab[i] = (__bf16)(a0 + i);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (MF) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc[m & 1], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) one<OP>(v[j & 7], w[j & 7], p[j & 7], a0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    for (int i = 0; i < 8; ++i) s += v[i] + w[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s + pad[0];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

static double g_wall_ns = 0;
template <int OP, int MF, int NM, int NV>
double run(int waves_per_simd, float* d, unsigned long long* c) {
    const int lds = 160 * 1024 / waves_per_simd - 2048;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<OP, MF, NM, NV>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int iters = 1000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP, MF, NM, NV><<<256 * waves_per_simd, 256, lds>>>(d, c, iters, 1.0f);
    hipEventRecord(e0);
    k<OP, MF, NM, NV><<<256 * waves_per_simd, 256, lds>>>(d, c, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    unsigned long long h = 0;
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    g_wall_ns = (double)ms * 1e6 / iters;      // wall time of one iteration (whole kernel / iters)
    return (double)h / iters;
}

template <int OP>
void op_rows(float* d, unsigned long long* c) {
    // alone: 64 instances per iteration -> s_memtime cycles (100 MHz ticks on gfx950? reported raw) per instruction
    double a[3], m[3], an[3], mn[3];
    int ws[3] = {1, 2, 4};
    for (int i = 0; i < 3; ++i) {
        a[i] = run<OP, 0, 8, 8>(ws[i], d, c) / 64.0; an[i] = g_wall_ns / 64.0;
        m[i] = run<OP, 1, 8, 6>(ws[i], d, c) / 8.0; mn[i] = g_wall_ns / 8.0;
    }
    printf("%-18s alone ticks/instr/wave @1,2,4 w/SIMD: %6.2f %6.2f %6.2f (wall ns %5.2f %5.2f %5.2f) | MFMA+6: ticks/group %6.2f %6.2f %6.2f (wall ns %5.1f %5.1f %5.1f)\n",
           op_name[OP], a[0], a[1], a[2], an[0], an[1], an[2], m[0], m[1], m[2], mn[0], mn[1], mn[2]);
}

// Heterogeneous pair: waves 0-3 of a 512-thread workgroup (one per SIMD) run MFMAs only, waves 4-7 (their SIMD
// partners) run the VALU instruction only.  MODE 1: MFMA waves alone, 2: VALU waves alone, 3: both.
template <int OP, int MODE>
__global__ __launch_bounds__(512) void het(float* out, int iters, float a0) {
    extern __shared__ float pad[];
    const int wave = threadIdx.x >> 6;
    f32x16 acc[2];
    for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float v[8], w[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { v[i] = a0 + i + threadIdx.x * 1e-6f; w[i] = 0.25f * i; p[i] = f32x2{v[i], w[i]}; }
    bf16x8 ab;
    for (int i = 0; i < 8; ++i) ab[i] = (__bf16)(a0 + i);
    if (wave < 4) {
        if (MODE & 1)
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc[m & 1], 0, 0, 0);
    } else {
        if (MODE & 2)
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int j = 0; j < 48; ++j) one<OP>(v[j & 7], w[j & 7], p[j & 7], a0);
    }
    float s = 0.f;
    for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    for (int i = 0; i < 8; ++i) s += v[i] + w[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s + pad[0];
}

template <int OP, int MODE>
double run_het(float* d) {
    const int lds = 150 * 1024;   // one workgroup per CU
    hipFuncSetAttribute(reinterpret_cast<const void*>(het<OP, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int iters = 1000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    het<OP, MODE><<<256, 512, lds>>>(d, iters, 1.0f);
    hipEventRecord(e0);
    het<OP, MODE><<<256, 512, lds>>>(d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return (double)ms * 1e6 / iters;   // ns per iteration (8 MFMAs | 48 VALU instructions)
}

template <int OP>
void het_row(float* d) {
    const double m = run_het<OP, 1>(d), v = run_het<OP, 2>(d), b = run_het<OP, 3>(d);
    printf("pair: 8 MFMA wave | 48 x %-18s wave: MFMA alone %6.1f ns, VALU alone %6.1f ns, together %6.1f ns (max %6.1f, sum %6.1f)\n",
           op_name[OP], m, v, b, m > v ? m : v, m + v);
}

// Two MIXED waves per SIMD, each alternating a matrix phase (MG MFMAs, two accumulator chains) and a VALU phase (VG
// instructions, 3/4 v_fma + 1/4 v_exp like the bf16x3 attention loop): does the pair run in max(...) or in the sum?
// OFFSET 1: waves 4-7 start with their VALU phase (anti-phase start); GRAIN: the phases are cut into GRAIN slices that
// alternate (GRAIN = 1: one matrix block then one VALU block; 12: [4 MFMA, 20 VALU] x 12).
template <int MG, int VG, int OFFSET, int GRAIN, int DEP = 0>
__global__ __launch_bounds__(1024) void phases(float* out, int iters, float a0) {
    extern __shared__ float pad[];
    const int wave = threadIdx.x >> 6;
    f32x16 acc[2];
    for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float v[8], w[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { v[i] = a0 + i + threadIdx.x * 1e-6f; w[i] = 0.25f * i; p[i] = f32x2{v[i], w[i]}; }
    bf16x8 ab;
    for (int i = 0; i < 8; ++i) ab[i] = (__bf16)(a0 + i);
    auto mphase = [&]() {
#pragma unroll
        for (int m = 0; m < MG / GRAIN; ++m) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %1, %0" : "+v"(acc[m & 1]) : "v"(ab));
    };
    auto vphase = [&]() {
        if (DEP) {   // the VALU phase consumes the matrix results (after the wait states hipcc would insert) ...
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(acc[j & 1][j]));
        }
#pragma unroll
        for (int j = 0; j < VG / GRAIN; ++j) {
            if ((j & 3) == 3) one<OP_EXP>(v[j & 7], w[j & 7], p[j & 7], a0);
            else one<OP_FMA>(v[j & 7], w[j & 7], p[j & 7], a0);
        }
        if (DEP) {   // ... and produces the next matrix phase's operand
            typedef float f32x2v __attribute__((ext_vector_type(2)));
            typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16x2v c = __builtin_convertvector(f32x2v{v[2 * j], v[2 * j + 1]}, bf16x2v);
                ab[2 * j] = c[0]; ab[2 * j + 1] = c[1];
            }
            asm volatile("" : "+v"(ab));
        }
    };
    if (OFFSET && wave >= 4) {
#pragma unroll
        for (int g = 0; g < GRAIN; ++g) vphase();
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < GRAIN; ++g) {
            mphase();
            vphase();
        }
    }
    float s = 0.f;
    for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    for (int i = 0; i < 8; ++i) s += v[i] + w[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s + pad[0];
}

template <int MG, int VG, int OFFSET, int GRAIN, int DEP = 0>
double run_phases(float* d, int waves) {
    const int lds = 150 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(phases<MG, VG, OFFSET, GRAIN, DEP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int iters = 400;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    phases<MG, VG, OFFSET, GRAIN, DEP><<<256, 64 * waves, lds>>>(d, iters, 1.0f);
    hipEventRecord(e0);
    phases<MG, VG, OFFSET, GRAIN, DEP><<<256, 64 * waves, lds>>>(d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return (double)ms * 1e6 / iters;
}

template <int MG, int VG>
void phase_rows(float* d) {
    const double m1 = run_phases<MG, 0, 0, 1>(d, 4), v1 = run_phases<0, VG, 0, 1>(d, 4), b1 = run_phases<MG, VG, 0, 1>(d, 4);
    printf("mixed waves, %d MFMA + %d VALU per iteration: ONE wave/SIMD: matrix only %.0f ns, VALU only %.0f ns, both %.0f ns\n", MG, VG, m1, v1, b1);
    printf("   TWO waves/SIMD (ideal = 2 x matrix only = %.0f ns; serial = 2 x both = %.0f ns): same phase %.0f | anti-phase start %.0f | "
           "12 slices in phase %.0f | 12 slices anti-phase %.0f\n", 2 * m1, 2 * b1,
           run_phases<MG, VG, 0, 1>(d, 8), run_phases<MG, VG, 1, 1>(d, 8), run_phases<MG, VG, 0, 12>(d, 8), run_phases<MG, VG, 1, 12>(d, 8));
}

int main() {
    float* d; unsigned long long* c;
    hipMalloc(&d, 1024); hipMalloc(&c, 64);
    {   // reference: MFMA alone, ticks per MFMA
        double r[3], rn[3]; int ws[3] = {1, 2, 4};
        for (int i = 0; i < 3; ++i) { r[i] = run<OP_FMA, 1, 8, 0>(ws[i], d, c) / 8.0; rn[i] = g_wall_ns / 8.0; }
        printf("%-18s ticks per MFMA per wave @1,2,4 w/SIMD: %6.2f %6.2f %6.2f (wall ns per MFMA per wave %5.1f %5.1f %5.1f; 32 cycles @2.4 GHz = 13.3 ns)\n",
               "mfma 32x32x16 bf16", r[0], r[1], r[2], rn[0], rn[1], rn[2]);
    }
    op_rows<OP_FMA>(d, c); op_rows<OP_SUB>(d, c); op_rows<OP_EXP>(d, c); op_rows<OP_CVTPK>(d, c); op_rows<OP_PKADD>(d, c);
    op_rows<OP_PKMUL>(d, c); op_rows<OP_AND>(d, c); op_rows<OP_LSHL>(d, c); op_rows<OP_MAX3>(d, c); op_rows<OP_MOV>(d, c);
    op_rows<OP_RCP>(d, c); op_rows<OP_DOT2C>(d, c);
    het_row<OP_SUB>(d); het_row<OP_FMA>(d); het_row<OP_EXP>(d); het_row<OP_CVTPK>(d); het_row<OP_PKADD>(d); het_row<OP_LSHL>(d); het_row<OP_DOT2C>(d);
    phase_rows<48, 240>(d);
    phase_rows<48, 120>(d);
    // true dependencies (VALU phase reads the matrix results, matrix phase reads the VALU results), same work per SIMD
    // spread over 1, 2 and 4 waves
    printf("dependent phases, 96 MFMA + 480 VALU per SIMD and round: 1 wave x (96 + 480): %.0f ns | 2 waves x (48 + 240): %.0f ns | "
           "4 waves x (24 + 120): %.0f ns   (matrix bound %.0f ns)\n",
           run_phases<96, 480, 0, 1, 1>(d, 4), run_phases<48, 240, 0, 1, 1>(d, 8), run_phases<24, 120, 0, 1, 1>(d, 16),
           2 * run_phases<48, 0, 0, 1>(d, 4));
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
