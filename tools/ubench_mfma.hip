// Microbenchmark: v_mfma_f32_32x32x2_f32 issue rate vs. dependent-chain count and waves/SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma.hip -o gpurun_out/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    extern __shared__ float pad[];  // dynamic LDS size limits workgroups per CU
    f32x16 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.678f) out[0] = s + pad[0];
}

template <int CHAINS>
void run(int wgs_per_cu, float* d) {
    const int lds = 160 * 1024 / wgs_per_cu - 1024;  // force exactly wgs_per_cu workgroups per CU
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<CHAINS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int iters = 4096 / CHAINS;
    const int grid = 256 * wgs_per_cu * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<CHAINS><<<grid, 256, lds>>>(d, iters, 1.0f, 0.5f);
    hipEventRecord(e0);
    k<CHAINS><<<grid, 256, lds>>>(d, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 /*waves*/ * iters * 16 * CHAINS * 4096.0;
    printf("chains=%d waves/SIMD=%d : %.1f TFLOP/s (%.3f ms)\n", CHAINS, wgs_per_cu, flop / ms / 1e9, ms);
}

int main() {
    float* d; hipMalloc(&d, 1024);
    for (int w : {1, 2, 4}) { run<1>(w, d); run<2>(w, d); run<4>(w, d); }
    return 0;
}
