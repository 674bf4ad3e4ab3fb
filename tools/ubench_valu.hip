// Microbenchmark: issue cost of the requantisation VALU instructions (v_bfe_u32, v_add3_u32, v_ashrrev_i32, v_med3_i32, v_perm_b32) per
// wave64 instruction and SIMD on gfx950, alone and beside v_mfma_i32_32x32x32_i8 from the same / the other wave of the SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o build/ubench_valu && build/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ int med3i(int v, int lo, int hi) { int r; asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi)); return r; }

// MODE 0: 16 independent chains of (bfe, add3, ashr, med3) = 64 VALU per iteration; MODE 1: + 4 MFMA per iteration (one per 16 VALU);
// MODE 2: 4 MFMA only
template <int MODE>
__global__ void __launch_bounds__(512) k(int iters, int n, int* out, unsigned long long* cyc) {
    int v[16];
    for (int c = 0; c < 16; ++c) v[c] = threadIdx.x * 77 + c * 1315423911;
    v16i acc[2];
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0; acc[1][r] = 0; }
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (MODE != 0 && (c & 3) == 0) acc[(c >> 2) & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[(c >> 2) & 1], 0, 0, 0);
            if (MODE != 2) {
                const unsigned odd = __builtin_amdgcn_ubfe((unsigned)v[c], (unsigned)n, 1u);
                int r; asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(v[c]), "v"(odd), "v"(i));
                asm volatile("v_ashrrev_i32 %0, %1, %2" : "=v"(r) : "v"(n), "v"(r));
                v[c] = med3i(r, -1000000 + c, 1000000) + v[c];
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    int s = 0;
    for (int c = 0; c < 16; ++c) s += v[c];
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
    if (s == 0x7fffffff) out[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(int threads) {
    unsigned long long* d; int* o; CK(hipMalloc(&d, 8 * 256)); CK(hipMalloc(&o, 4));
    const int iters = 2048;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, iters, 7, o, d);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, iters, 7, o, d);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[256]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i]; avg /= 256;
    const int wps = threads / 256 < 1 ? 1 : threads / 256;
    printf("mode %d (%s), %d wave(s) per SIMD: %.0f counter-cycles per iteration and wave (80 VALU%s) | %.1f us total -> %.1f ns per iteration per SIMD\n", MODE,
           MODE == 0 ? "VALU only" : MODE == 1 ? "VALU + 4 MFMA" : "4 MFMA only", wps, avg / iters, MODE ? " + 4 MFMA" : "", ms * 1e3, ms * 1e6 / iters / wps);
}

int main() {
    run<0>(256); run<0>(512); run<2>(256); run<2>(512); run<1>(256); run<1>(512);
    return 0;
}
