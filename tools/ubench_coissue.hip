// Microbenchmark: do bf16 (and f32) MFMAs overlap with VALU work on one SIMD of gfx950?
// Each wave runs `iters` iterations of [NM MFMAs, each followed by NV independent v_fma_f32]; one
// workgroup of 256 threads per CU slot; 1 or 2 waves per SIMD.  Reports cycles per iteration per wave
// (s_memtime) next to the two standalone costs.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_coissue.hip -o tools/bin/ubench_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NM, int NV, int CH = 2>   // KIND 0: bf16 32x32x16, 1: f32 32x32x2; CH = accumulator chains
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float a0) {
    extern __shared__ float pad[];
    f32x16 acc[2];
    for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a0 + i + threadIdx.x * 1e-6f;
    bf16x8 ab;
    for (int i = 0; i < 8; ++i) ab[i] = (__bf16)(a0 + i);
    const float af = a0, bf = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < (NM > 0 ? NM : 1); ++m) {
            if (NM > 0) {
                if (KIND == 0) acc[m % CH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc[m % CH], 0, 0, 0);
                else acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[m & 1], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j & 7]) : "v"(af));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) out[0] = s + pad[0];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int NM, int NV, int CH = 2>
void run(const char* name, int waves_per_simd, float* d, unsigned long long* c) {
    const int lds = 160 * 1024 / waves_per_simd - 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<KIND, NM, NV, CH>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND, NM, NV, CH><<<256 * waves_per_simd, 256, lds>>>(d, c, iters, 1.0f);
    hipEventRecord(e0);
    k<KIND, NM, NV, CH><<<256 * waves_per_simd, 256, lds>>>(d, c, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h = 0;
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    const double flop = (double)256 * waves_per_simd * 4 * iters * NM * (KIND == 0 ? 32768.0 : 4096.0);
    printf("%-34s waves/SIMD=%d : %7.1f cycles / iteration / wave   kernel %.3f ms = %.0f TFLOP/s (MFMA)  (%s)\n", name,
           waves_per_simd, (double)h / iters, ms, flop / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

int main() {
    float* d; unsigned long long* c;
    hipMalloc(&d, 1024); hipMalloc(&c, 64);
    for (int w : {1, 2}) {
        run<0, 16, 0>("bf16: 16 MFMA", w, d, c);
        run<0, 16, 0, 1>("bf16: 16 MFMA, ONE dependent chain", w, d, c);
        run<0, 0, 96>("96 v_fma (no MFMA)", w, d, c);
        run<0, 16, 3>("bf16: 16 x (MFMA + 3 v_fma)", w, d, c);
        run<0, 16, 6>("bf16: 16 x (MFMA + 6 v_fma)", w, d, c);
        run<0, 16, 12>("bf16: 16 x (MFMA + 12 v_fma)", w, d, c);
        run<1, 16, 0>("f32: 16 MFMA", w, d, c);
        run<1, 16, 6>("f32: 16 x (MFMA + 6 v_fma)", w, d, c);
        run<1, 16, 12>("f32: 16 x (MFMA + 12 v_fma)", w, d, c);
    }
    return 0;
}
