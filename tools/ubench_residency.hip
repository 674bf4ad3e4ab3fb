// Microbenchmark: how many 512-thread workgroups with a large dynamic LDS allocation are resident at once on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_residency.hip -o /tmp/ubench_residency && /tmp/ubench_residency
// Each workgroup spins for a fixed number of cycles; kernel time vs grid size shows the number of rounds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int VG>
__global__ void __launch_bounds__(512) spin(int cycles, int* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    lds[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    int v = lds[(threadIdx.x * 7) & 511];
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)cycles) v = v * 3 + 1;
    if (v == 0x7fffffff) out[0] = v;
}

int main() {
    int* out; CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int kbs[] = {32, 64, 80, 96, 120, 136, 160};
    for (int kb : kbs) {
        CK(hipFuncSetAttribute((const void*)spin<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024));
        printf("LDS %3d KB:", kb);
        for (int grid : {64, 128, 192, 256, 320, 512}) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(spin<0>, dim3(grid), dim3(512), kb * 1024, 0, 100000, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("  grid %3d: %6.1f us", grid, best * 1e3f);
        }
        printf("\n");
    }
    return 0;
}
