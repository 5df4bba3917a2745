// Micro-benchmark of the bf16 streaming GEMM (t2s_bf16.h): times each instantiation at the
// training shape M = 1152*480 and prints GB/s of algorithmic HBM bytes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I t2ms_amd/csrc tools/ubench_bgemm.hip -o /tmp/ubench_bgemm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "t2s_bf16.h"
namespace t2s { void set_error(const char*, ...) {} }
using namespace t2s;

template <int K, int N, int PRO, int EPI>
void run(const char* name, int M, void* A, bf16x8* Wp, float* bias, __bf16* out, float* mod, __bf16* save, __bf16* aux,
         double bytes) {
    BGemmArgs a{};
    a.A = A; a.Wp = Wp; a.bias = bias; a.out = out; a.M = M; a.N = N; a.mod = mod; a.shift_off = 0; a.scale_off = 128;
    a.save_A = save; a.aux = aux; a.q = out; a.k = out + (size_t)M * 128; a.v = out + (size_t)M * 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) launch_bgemm<K, N, PRO, EPI>(a, 0);
    hipEventRecord(e0, 0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) launch_bgemm<K, N, PRO, EPI>(a, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.1f us  %7.0f GB/s  (%s)\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9, hipGetErrorString(hipGetLastError()));
}

__global__ void fill_rand(unsigned* p, size_t n, unsigned seed, unsigned mask, unsigned orv) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (x & mask) | orv;
}

int main(int argc, char** argv) {
    const bool rnd = argc > 1;
    const int S = 1152, M = S * 480;
    void *A; bf16x8* Wp; float *bias, *mod; __bf16 *out, *save, *aux;
    hipMalloc(&A, (size_t)M * 384 * 4);
    hipMalloc(&Wp, 384 * 128 * 2);
    hipMalloc(&bias, 384 * 4);
    hipMalloc(&mod, (size_t)S * MODROW * 4);
    hipMalloc(&out, (size_t)M * 384 * 2);
    hipMalloc(&save, (size_t)M * 256 * 2);
    hipMalloc(&aux, (size_t)M * 256 * 2);
    hipMemset(A, 0, (size_t)M * 384 * 4); hipMemset(Wp, 0, 384 * 128 * 2); hipMemset(bias, 0, 384 * 4);
    hipMemset(mod, 0, (size_t)S * MODROW * 4); hipMemset(aux, 0, (size_t)M * 256 * 2);
    if (rnd) {   // random bf16 pairs / fp32 values in (-2, 2): sign + exponent bits confined to small magnitudes
        auto fill = [&](void* p, size_t bytes, unsigned mask, unsigned orv) {
            fill_rand<<<(unsigned)((bytes / 4 + 255) / 256), 256>>>((unsigned*)p, bytes / 4, 12345u, mask, orv);
        };
        fill(A, (size_t)M * 384 * 4, 0x80ff80ffu | 0x007f007fu, 0x3f003f00u);   // as bf16 pairs: |x| in [0.5, 2)
        fill(Wp, 384 * 128 * 2, 0x80ff80ffu, 0x3c003c00u);
        fill(aux, (size_t)M * 256 * 2, 0x80ff80ffu, 0x3f003f00u);
        hipDeviceSynchronize();
    }
    const double u = (double)M * 128 * 2;   // one bf16 (M,128) tensor
    run<128, 128, BPRO_BF16, BEPI_BF16>("128->128 plain", M, A, Wp, bias, out, mod, nullptr, aux, 2 * u);
    run<128, 384, BPRO_LN, BEPI_QKV>("128->384 LN qkv", M, A, Wp, bias, out, mod, save, aux, 6 * u);
    run<128, 256, BPRO_LN, BEPI_BF16>("128->256 LN", M, A, Wp, bias, out, mod, save, aux, 5 * u);
    run<256, 128, BPRO_GELU, BEPI_BF16>("256->128 gelu", M, A, Wp, bias, out, mod, save, aux, 5 * u);
    run<128, 256, BPRO_BF16, BEPI_GELUBWD>("128->256 gelubwd", M, A, Wp, bias, out, mod, nullptr, aux, 5 * u);
    run<256, 128, BPRO_BF16, BEPI_BF16>("256->128 plain", M, A, Wp, bias, out, mod, nullptr, aux, 3 * u);
    run<384, 128, BPRO_BF16, BEPI_BF16>("384->128 plain", M, A, Wp, bias, out, mod, nullptr, aux, 4 * u);
    // weight-gradient kernel: dW (N,K) += dY^T X
    float* dW; hipMalloc(&dW, 384 * 256 * 4); hipMemset(dW, 0, 384 * 256 * 4);
    const size_t scr = wgrad16_scratch_floats(M, 256); float* scratch; hipMalloc(&scratch, scr * 4);
    const int shapes[4][2] = {{128, 128}, {256, 128}, {384, 128}, {128, 256}};
    for (auto& sh : shapes) {
        const int N = sh[0], K = sh[1];
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w = 0; w < 2; ++w) launch_wgrad16((const __bf16*)out, (const __bf16*)A, dW, bias, M, N, K, scratch, scr, 256, 0);
        hipEventRecord(e0, 0);
        for (int r = 0; r < 5; ++r) launch_wgrad16((const __bf16*)out, (const __bf16*)A, dW, bias, M, N, K, scratch, scr, 256, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)M * (N + K) * 2;
        printf("wgrad16 N=%d K=%d  %8.1f us  %7.0f GB/s (single-read bytes) (%s)\n", N, K, ms / 5 * 1e3,
               bytes / (ms / 5 * 1e-3) / 1e9, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
