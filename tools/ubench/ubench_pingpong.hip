// Microbenchmark (gfx950): one workgroup of 8 waves per CU runs a loop of two phases — M: a matrix-pipe-bound K loop (52 x v_mfma_i32_32x32x32_i8 per wave, one
// ds_read_b128 per MFMA: P1 + P2 of chain_kernel<256,64,56,56,...>), V: a vector-bound phase (8 units of P3: 2 MFMAs + join + integer requantisation + store per unit) —
//   lockstep : all 8 waves run M, s_barrier, V, s_barrier (what chain_kernel does: the two waves of a SIMD are always in the SAME phase);
//   pingpong : waves 0-3 (one per SIMD) and waves 4-7 are two GROUPS with barriers of their own (an LDS counter per group), group B half a period behind:
//              on every SIMD one wave is in M while the other is in V.
// Question (round 5): how much of a chain block's time is the lockstep itself?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/ubench_pingpong.hip -o tools/ubench/ubench_pingpong.bin && tools/ubench/ubench_pingpong.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned rq4_int(int a, int b, int c, int d, int n) {
    const unsigned hm1 = (1u << (n - 1)) - 1u;
    const int ta = (int)((unsigned)a + hm1 + __builtin_amdgcn_ubfe((unsigned)a, (unsigned)n, 1u)), tb = (int)((unsigned)b + hm1 + __builtin_amdgcn_ubfe((unsigned)b, (unsigned)n, 1u));
    const int tc = (int)((unsigned)c + hm1 + __builtin_amdgcn_ubfe((unsigned)c, (unsigned)n, 1u)), td = (int)((unsigned)d + hm1 + __builtin_amdgcn_ubfe((unsigned)d, (unsigned)n, 1u));
    unsigned lo, hi;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "=v"(lo) : "v"(ta), "v"(tb), "s"(n));
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "=v"(hi) : "v"(tc), "v"(td), "s"(n));
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

// group barrier: `members` waves; generation g (1, 2, ...).  One lane arrives, every lane of the wave waits on the same LDS word.
__device__ __forceinline__ void group_barrier(unsigned* ctr, unsigned target) {
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < (int)target) __builtin_amdgcn_s_sleep(1);
}

// Round 6 (VERDICT r5 #1a): modes 1 / 2 run a FIXED iteration count per group, so the figure printed for the slower group mixes the time it shares the CU
// with the time it runs alone after the first-dispatched group has finished; "aggregate" cannot be read off it.  Mode 3 measures it: the two groups PULL
// iterations from one LDS counter until 2 x iters group-iterations are done (the work of `iters` lockstep iterations); mode 4 = one group alone on the CU.
template <int MODE>      // 0: lockstep, 1: pingpong, 2: pingpong without the initial offset (both groups start in M), 3: two groups pulling work, 4: one group alone
__global__ void __launch_bounds__(512) pp_kernel(int iters, int n, int sh, int* out, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int grp = wave >> 2;
    unsigned* const ctr = (unsigned*)(lds + 140 * 1024) + grp * 32;
    if (tid < 64) ((unsigned*)(lds + 140 * 1024))[tid] = 0u;
    v16i res[7];
    for (int t = 0; t < 7; ++t) for (int r = 0; r < 16; ++r) res[t][r] = tid * 31 + t * 7 + r;
    for (int i = tid; i < 32768; i += 512) ((int*)lds)[i] = i * 2654435761u;
    __syncthreads();
    v4i w0 = {tid, tid + 1, tid + 2, tid + 3}, w1 = {tid * 3, tid * 5, tid * 7, tid * 9};
    const char* bsrc = lds + 4096 + (lane & 31) * 80 + (lane >> 5) * 16;
    char* xdst = lds + 65536 + (lane & 31) * 272 + (lane >> 5) * 16 + (wave & 7) * 32;
    v16i breg; for (int r = 0; r < 16; ++r) breg[r] = tid + r;
    v16i acc0 = breg, acc1 = breg;
    auto phase_m = [&]() {          // 52 MFMAs on two accumulators, one B fragment each from LDS (a step ahead is left to the compiler)
#pragma unroll
        for (int k = 0; k < 26; ++k) {
            const v4i x0 = *(const v4i*)(bsrc + k * 2560), x1 = *(const v4i*)(bsrc + k * 2560 + 32);
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0, x0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1, x1, acc1, 0, 0, 0);
        }
        // epilogue of P1 / P2: 2 x 16 values requantised and stored twice
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            unsigned d[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) d[g] = rq4_int(acc0[4 * g] + q, acc0[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3], n) ^ 0x80808080u;
            const v4i o = {(int)d[0], (int)d[1], (int)d[2], (int)d[3]};
            *(v4i*)(xdst + q * 32 * 272) = o;
        }
    };
    auto phase_v = [&]() {          // 7 units of P3
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            const v4i x0 = *(const v4i*)(bsrc + t * 2560), x1 = *(const v4i*)(bsrc + t * 2560 + 32);
            v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0, x0, breg, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1, x1, acc, 0, 0, 0);
            v16i& rr = res[t];
#pragma unroll
            for (int r = 0; r < 16; ++r) rr[r] = max((int)(((unsigned)acc[r] << sh) + (unsigned)rr[r]), 0);
            unsigned d[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) d[g] = rq4_int(rr[4 * g], rr[4 * g + 1], rr[4 * g + 2], rr[4 * g + 3], n) ^ 0x80808080u;
            auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
            const v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
            *(v4i*)(xdst + t * 32 * 272) = o;
        }
    };
    unsigned gen = 0;
    auto bar = [&]() {
        if constexpr (MODE == 0) __syncthreads();
        else { gen += 4; group_barrier(ctr, gen); }      // (mode 4: the idle group never arrives at group 0's counter — groups have counters of their own)
    };
    const unsigned long long t0 = __builtin_readcyclecounter();
    int done = 0;
    if constexpr (MODE == 3) {
        unsigned* const taken = (unsigned*)(lds + 140 * 1024) + 96;        // shared ticket counter; slot[grp][parity] = the group's current ticket
        unsigned* const slot = (unsigned*)(lds + 140 * 1024) + 128 + grp * 2;
        unsigned par = 0;
        for (;;) {
            if ((wave & 3) == 0 && lane == 0) slot[par] = __hip_atomic_fetch_add(taken, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            bar();
            const unsigned tk = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(slot + par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            par ^= 1u;
            if (tk >= (unsigned)(2 * iters)) break;
            phase_m(); bar();
            phase_v();
            ++done;
        }
    } else if constexpr (MODE == 4) {
        if (grp == 0) for (int it = 0; it < iters; ++it) { phase_m(); bar(); phase_v(); bar(); ++done; }
    } else {
        if (MODE == 1 && grp == 1) { phase_m(); bar(); }        // group B: half a period behind
        for (int it = 0; it < iters; ++it) {
            phase_m(); bar();
            phase_v(); bar();
            ++done;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    int s = 0; for (int t = 0; t < 7; ++t) for (int r = 0; r < 16; ++r) s += res[t][r];
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 0x7fffffff) out[0] = s;
    if (lane == 0) { cyc[blockIdx.x * 8 + wave] = t1 - t0; cyc[2048 + blockIdx.x * 8 + wave] = (unsigned long long)done; }
}

template <int MODE> static void run(const char* name, int iters) {
    int* out; unsigned long long* cyc;
    CK(hipMalloc((void**)&out, 64)); CK(hipMalloc((void**)&cyc, 2 * 256 * 8 * 8)); CK(hipMemset(cyc, 0, 2 * 256 * 8 * 8));
    auto k = pp_kernel<MODE>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 150 * 1024, 0, 4, 8, 0, out, cyc); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 150 * 1024, 0, iters, 8, 0, out, cyc); CK(hipDeviceSynchronize());
    static unsigned long long h[2 * 256 * 8]; CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
    double mx = 0, a = 0, b = 0, da = 0, db = 0;
    for (int i = 0; i < 256; ++i) { double m = 0; for (int w = 0; w < 8; ++w) m = h[i * 8 + w] > m ? h[i * 8 + w] : m; mx += m; a += h[i * 8]; b += h[i * 8 + 7]; da += h[2048 + i * 8]; db += h[2048 + i * 8 + 7]; }
    // group-iterations per workgroup and their aggregate rate: lockstep = 2 groups x iters
    const double gi = (da + db) / 256;
    printf("%-52s %8.0f cycles per (M + V) iteration of BOTH groups (slowest wave / (group-iterations / 2)); wave 0: %6.0f cycles total / %5.1f iterations, wave 7: %6.0f / %5.1f; aggregate %.4f group-iterations per k-cycle\n",
           name, mx / 256 / (gi / 2), a / 256, da / 256, b / 256, db / 256, 1e3 * gi / (mx / 256));
    CK(hipFree(out)); CK(hipFree(cyc));
}
int main() {
    run<0>("lockstep: 8 waves, s_barrier per phase", 200);
    run<1>("pingpong: two 4-wave groups, half a period apart", 200);
    run<2>("two groups, own barriers, same start", 200);
    run<3>("two groups PULLING iterations from one counter", 200);
    run<4>("one 4-wave group alone on the CU", 200);
    return 0;
}
