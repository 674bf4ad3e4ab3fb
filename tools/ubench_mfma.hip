// Microbenchmark: issue rate of v_mfma_i32_32x32x32_i8 (cycles per instruction per SIMD) on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma.hip -o build/ubench_mfma && build/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int CHAINS>
__global__ void __launch_bounds__(512) k(int iters, int* out, unsigned long long* cyc) {
    v16i acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0;
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    int s = 0;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 0x7fffffff) out[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CHAINS>
void run(int threads) {
    unsigned long long* d; int* o; CK(hipMalloc(&d, 8 * 256)); CK(hipMalloc(&o, 4));
    const int iters = 4096;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<CHAINS>, dim3(256), dim3(threads), 0, 0, iters, o, d);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<CHAINS>, dim3(256), dim3(threads), 0, 0, iters, o, d);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[256]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i]; avg /= 256;
    const int waves_per_simd = threads / 256;
    const double mfma_per_simd = (double)iters * CHAINS * (waves_per_simd < 1 ? 1 : waves_per_simd);
    const double ops = 2.0 * 32 * 32 * 32 * (double)iters * CHAINS * (threads / 64) * 256;
    printf("chains %d waves/WG %d: %.1f counter-cycles per MFMA per SIMD | %.1f us -> %.0f TOP/s chip (%.2f ns per MFMA per SIMD)\n", CHAINS, threads / 64,
           avg / mfma_per_simd, ms * 1e3, ops / (ms * 1e-3) / 1e12, ms * 1e6 / mfma_per_simd);
}

int main() {
    run<1>(256); run<2>(256); run<4>(256); run<4>(512); run<8>(256);
    return 0;
}
