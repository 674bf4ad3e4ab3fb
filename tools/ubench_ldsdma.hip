// Microbenchmark: what does a CU's LDS-direct DMA stream (`buffer_load_dwordx4 ... lds`) sustain?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_ldsdma.hip -o /tmp/ubench_ldsdma && /tmp/ubench_ldsdma
// A workgroup of 512 threads streams STEPS stages of S bytes through an LDS ring of depth D (counted vmcnt +
// one barrier per step, exactly the K-loop skeleton of the conv kernels, no MFMA / ds_read), from
//   src A: a small buffer every workgroup re-reads (weights: L2-resident), rows of 64 B at stride `pitch`
//   src B: a private slice per workgroup of a large buffer (activations: HBM / MALL)
// Reports bytes/clk/CU and chip TB/s for grids of 1 and 2 workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// S = stage bytes (multiple of 8192: one 16-B slot per thread per 8 KB), D = ring depth
template <int S, int D, bool PRIVATE>
__global__ void __launch_bounds__(512) k_stream(const char* src, unsigned src_bytes, int steps, int pitch, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int L = S / 8192;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, src_bytes, 0x00020000);
    unsigned base[L];
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int slot = tid + i * 512;
        if (PRIVATE) base[i] = (unsigned)(((size_t)blockIdx.x * (size_t)steps * S + (size_t)slot * 16) % src_bytes);
        else base[i] = (unsigned)((slot >> 2) * pitch + (slot & 3) * 16);          // rows of 64 B at stride pitch
    }
    auto issue = [&](int j, int slot) {
#pragma unroll
        for (int i = 0; i < L; ++i) {
            unsigned off = PRIVATE ? base[i] + (unsigned)(j * S) : base[i] + (unsigned)((j * 64) % pitch);
            if (off >= src_bytes) off -= src_bytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + slot * S + i * 8192 + wave * 1024), 16, off, 0, 0, 0);
        }
    };
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int s = 0; s < D - 1; ++s) issue(s, s);
    for (int j = 0; j < steps; ++j) {
        const int ahead = (steps - 1 - j) < (D - 2) ? (steps - 1 - j) : (D - 2);
        if (ahead <= 0) wait_vmcnt<0>();
        else if (ahead == 1) wait_vmcnt<1 * L>();
        else if (ahead == 2) wait_vmcnt<2 * L>();
        else if (ahead <= 4) wait_vmcnt<(3 * L > 15 ? 15 : 3 * L)>();
        else wait_vmcnt<(6 * L > 15 ? 15 : 6 * L)>();
        __builtin_amdgcn_s_barrier();
        if (j + D - 1 < steps) issue(j + D - 1, (j + D - 1) % D);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    if (steps < 0) cyc[0] = lds[tid];
}

template <int S, int D, bool PRIVATE>
void run(const char* name, const char* src, unsigned src_bytes, int pitch, int wg_per_cu, unsigned long long* d_cyc) {
    const int steps = 256, grid = 256 * wg_per_cu;
    CK(hipFuncSetAttribute((const void*)k_stream<S, D, PRIVATE>, hipFuncAttributeMaxDynamicSharedMemorySize, S * D));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_stream<S, D, PRIVATE>), dim3(grid), dim3(512), S * D, 0, src, src_bytes, steps, pitch, d_cyc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid);
    CK(hipMemcpy(h.data(), d_cyc, grid * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += (double)v; avg /= grid;
    const double bytes = (double)steps * S;
    printf("%-10s S=%5d D=%d wg/cu=%d : %7.1f cyc/step  %5.1f B/clk/WG  %5.1f B/clk/CU | kernel %7.1f us  %6.2f TB/s chip\n", name, S, D, wg_per_cu,
           avg / steps, bytes / avg, bytes / avg * wg_per_cu, ms * 1e3, bytes * grid / (ms * 1e-3) / 1e12);
}

int main() {
    char* small; char* big; unsigned long long* d_cyc;
    const unsigned small_bytes = 256 * 2304, big_bytes = 1u << 30;
    CK(hipMalloc(&small, small_bytes)); CK(hipMalloc(&big, big_bytes)); CK(hipMalloc(&d_cyc, 8 * 4096));
    CK(hipMemset(small, 1, small_bytes)); CK(hipMemset(big, 1, big_bytes));
    printf("== shared L2-resident source (weight rows, 64 B at pitch 2304)\n");
    run<8192, 2, false>("weights", small, small_bytes, 2304, 1, d_cyc);
    run<8192, 3, false>("weights", small, small_bytes, 2304, 1, d_cyc);
    run<8192, 4, false>("weights", small, small_bytes, 2304, 1, d_cyc);
    run<8192, 8, false>("weights", small, small_bytes, 2304, 1, d_cyc);
    run<8192, 4, false>("weights", small, small_bytes, 2304, 2, d_cyc);
    run<8192, 8, false>("weights", small, small_bytes, 2304, 2, d_cyc);
    run<16384, 4, false>("weights", small, small_bytes, 2304, 1, d_cyc);
    run<16384, 4, false>("weights", small, small_bytes, 2304, 2, d_cyc);
    run<24576, 3, false>("weights", small, small_bytes, 2304, 1, d_cyc);
    printf("== shared L2-resident source, power-of-two pitches (1x1 conv weights [Cout][K], K = 1024 / 2048) vs odd pitch\n");
    run<16384, 4, false>("pitch1024", small, small_bytes, 1024, 1, d_cyc);
    run<16384, 4, false>("pitch2048", small, small_bytes, 2048, 1, d_cyc);
    run<16384, 4, false>("pitch1088", small, small_bytes, 1088, 1, d_cyc);
    run<16384, 4, false>("pitch2304", small, small_bytes, 2304, 1, d_cyc);
    run<24576, 3, false>("pitch1024", small, small_bytes, 1024, 1, d_cyc);
    run<24576, 3, false>("pitch1088", small, small_bytes, 1088, 1, d_cyc);
    printf("== shared L2-resident source, contiguous (pitch 64)\n");
    run<8192, 4, false>("contig", small, small_bytes, 64, 1, d_cyc);
    run<16384, 4, false>("contig", small, small_bytes, 64, 2, d_cyc);
    printf("== private slices of a 1 GiB buffer (activations from HBM)\n");
    run<8192, 4, true>("private", big, big_bytes, 0, 1, d_cyc);
    run<8192, 8, true>("private", big, big_bytes, 0, 1, d_cyc);
    run<16384, 4, true>("private", big, big_bytes, 0, 1, d_cyc);
    run<16384, 4, true>("private", big, big_bytes, 0, 2, d_cyc);
    run<24576, 3, true>("private", big, big_bytes, 0, 2, d_cyc);
    return 0;
}
