// Microbenchmark: what bandwidth do the epilogue access patterns reach on int32 [M][C] rows?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_stream.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// coalesced: lane = 16 B consecutive
template <int MODE>  // 0 rw in place, 1 read only, 2 write only
__global__ void __launch_bounds__(256) k_coal(int* p, size_t n16, int* sink) {
    int acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        v4i v = {1, 2, 3, 4};
        if (MODE != 2) v = ((v4i*)p)[i];
        v.x += 1; v.y ^= 3; v.z += v.x; v.w -= 1;
        if (MODE != 1) ((v4i*)p)[i] = v; else acc += v.x + v.y + v.z + v.w;
    }
    if (MODE == 1 && acc == 0x12345678) *sink = acc;
}

// MFMA 32x32 D-layout: block 256 thr = 4 waves (2x2), tile 128 px x 128 co, wave 64x64,
// lane l: pixel l&31, channels 8g+4(l>>5)..+3  (16 B), g = 0..3; TILES = 2x2 per wave
template <int MODE, int C>
__global__ void __launch_bounds__(256) k_mfma(int* p, int M, int* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wpx = wave >> 1, wco = wave & 1;
    const int tilesN = C / 128;
    const int tile_n = blockIdx.x % tilesN, tile_m = blockIdx.x / tilesN;
    const int l31 = lane & 31, lh = lane >> 5;
    v4i r[2][2][4];
    int acc = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = tile_m * 128 + wpx * 64 + j * 32 + l31;
            const int co = tile_n * 128 + wco * 64 + i * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v4i v = {1, 2, 3, 4};
                if (MODE != 2 && m < M) v = *(v4i*)(p + (size_t)m * C + co + 8 * g + 4 * lh);
                r[i][j][g] = v;
            }
        }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = tile_m * 128 + wpx * 64 + j * 32 + l31;
            const int co = tile_n * 128 + wco * 64 + i * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v4i v = r[i][j][g];
                v.x += 1; v.y ^= 3; v.z += v.x; v.w -= 1;
                if (MODE != 1) { if (m < M) *(v4i*)(p + (size_t)m * C + co + 8 * g + 4 * lh) = v; }
                else acc += v.x + v.y + v.z + v.w;
            }
        }
    if (MODE == 1 && acc == 0x12345678) *sink = acc;
}

template <typename F> float timeit(F f, int reps = 20) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char** argv) {
    const int imgs = argc > 1 ? atoi(argv[1]) : 128;
    const int M = imgs * 56 * 56; constexpr int C = 256;
    const size_t bytes = (size_t)M * C * 4;
    int* p; int* sink; CK(hipMalloc(&p, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(p, 1, bytes));
    const size_t n16 = bytes / 16;
    printf("int32 [%d][%d] = %.1f MB\n", M, C, bytes / 1e6);
    for (int grid : {2048, 8192}) {
        float t;
        t = timeit([&] { k_coal<0><<<grid, 256>>>(p, n16, sink); }); printf("coalesced rw  grid %5d: %7.1f us  %6.0f GB/s (r+w)\n", grid, t * 1e3, 2 * bytes / t / 1e6);
        t = timeit([&] { k_coal<1><<<grid, 256>>>(p, n16, sink); }); printf("coalesced r   grid %5d: %7.1f us  %6.0f GB/s\n", grid, t * 1e3, bytes / t / 1e6);
        t = timeit([&] { k_coal<2><<<grid, 256>>>(p, n16, sink); }); printf("coalesced w   grid %5d: %7.1f us  %6.0f GB/s\n", grid, t * 1e3, bytes / t / 1e6);
    }
    const int grid = ((M + 127) / 128) * (C / 128);
    float t;
    t = timeit([&] { k_mfma<0, C><<<grid, 256>>>(p, M, sink); }); printf("mfma-layout rw (grid %d): %7.1f us  %6.0f GB/s (r+w)\n", grid, t * 1e3, 2 * bytes / t / 1e6);
    t = timeit([&] { k_mfma<1, C><<<grid, 256>>>(p, M, sink); }); printf("mfma-layout r : %7.1f us  %6.0f GB/s\n", t * 1e3, bytes / t / 1e6);
    t = timeit([&] { k_mfma<2, C><<<grid, 256>>>(p, M, sink); }); printf("mfma-layout w : %7.1f us  %6.0f GB/s\n", t * 1e3, bytes / t / 1e6);
    return 0;
}
