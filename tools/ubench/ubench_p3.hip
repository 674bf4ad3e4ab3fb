// Microbenchmark (gfx950): the P3 phase of chain_kernel<256,64,56,56,...> (f8_chain.hip) — per (pixel tile, channel tile) unit: bias + B-fragment LDS reads, two
// dependent v_mfma_i32_32x32x32_i8, the residual join (v_lshl_add_u32, v_max_i32), the integer requantisation (v_bfe_u32, v_add3_u32, v_ashr_pk_u8_i32,
// v_perm_b32), two v_permlane32_swap and one ds_write_b128 — with the tile's 56 units spread over 8, 12 or 16 waves of ONE workgroup per CU.
// Question: what does P3 cost when the same work runs on four waves per SIMD instead of two?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/ubench_p3.hip -o tools/ubench/ubench_p3.bin && tools/ubench/ubench_p3.bin
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
__device__ __forceinline__ unsigned rq4_flt(int a, int b, int c, int d, float sc) {
    unsigned r = __builtin_amdgcn_cvt_pk_u8_f32((float)a * sc, 0u, 0u);
    r = __builtin_amdgcn_cvt_pk_u8_f32((float)b * sc, 1u, r);
    r = __builtin_amdgcn_cvt_pk_u8_f32((float)c * sc, 2u, r);
    return __builtin_amdgcn_cvt_pk_u8_f32((float)d * sc, 3u, r);
}

// NW waves; waves 0..7 own TA units of the stream, waves 8.. own TB.  FLT: float-converter requantisation.  PIPE: the next unit's LDS reads and MFMAs are issued
// in front of the current unit's vector work (software pipeline by one unit).
template <int NW, int TA, int TB, bool FLT, bool PIPE, int ABL = 0>
__global__ void __launch_bounds__(NW * 64) p3_kernel(int iters, int n, int sh, int* out, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    constexpr int TM = TA > TB ? TA : TB;
    v16i res[TM];
    for (int t = 0; t < TM; ++t) for (int r = 0; r < 16; ++r) res[t][r] = tid * 31 + t * 7 + r;
    for (int i = tid; i < 40960; i += NW * 64) ((int*)lds)[i] = i * 2654435761u;
    __syncthreads();
    v4i w0 = {tid, tid + 1, tid + 2, tid + 3}, w1 = {tid * 3, tid * 5, tid * 7, tid * 9};
    const float sc = __builtin_ldexpf(1.0f, -n);
    const char* bias = lds + (wave & 7) * 128 + (lane >> 5) * 16;
    const char* bsrc = lds + 4096 + (lane & 31) * 80 + (lane >> 5) * 16;
    char* xdst = lds + 65536 + (lane & 31) * 272 + (lane >> 5) * 16 + (wave & 7) * 32;
    const int mine = wave < 8 ? TA : TB;
    v16i breg;
    for (int g = 0; g < 4; ++g) { const v4i bv = *(const v4i*)(bias + g * 32); for (int e = 0; e < 4; ++e) breg[4 * g + e] = bv[e]; }
    asm volatile("" : "+v"(breg));
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        auto mm = [&](int t) {
            v16i acc;
            if constexpr (!(ABL & 32)) {
#pragma unroll
            for (int g = 0; g < 4; ++g) { const v4i bv = *(const v4i*)(bias + g * 32); for (int e = 0; e < 4; ++e) acc[4 * g + e] = bv[e]; }
            }
            const v4i x0 = *(const v4i*)(bsrc + t * 2560), x1 = *(const v4i*)(bsrc + t * 2560 + 32);
            if constexpr (ABL & 1) { acc[0] ^= x0[0] ^ x1[1]; }       // ablation 1: no MFMA
            else if constexpr (ABL & 32) {                            // variant 32: the bias stays in 16 registers and is the first MFMA's C operand (no LDS read, no copy)
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0, x0, breg, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1, x1, acc, 0, 0, 0); }
            else {
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0, x0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1, x1, acc, 0, 0, 0); }
            return acc;
        };
        auto fin = [&](int t, const v16i& acc) {
            v16i& rr = res[t];
            if constexpr (ABL & 64) {          // variant 64: the join as two passes of 16 independent instructions (no dependent pair back to back)
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("v_lshl_add_u32 %0, %1, %2, %0" : "+v"(rr[r]) : "v"(acc[r]), "s"(sh));
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("v_max_i32 %0, 0, %0" : "+v"(rr[r]));
            } else
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if constexpr (ABL & 2) rr[r] = acc[r];                                   // ablation 2: no join (the MFMA result is the stream)
                else if constexpr (ABL & 16) rr[r] = max(acc[r], 0);                    // ablation 16: the join folded into the MFMA's C operand: ReLU only
                else rr[r] = max((int)(((unsigned)acc[r] << sh) + (unsigned)rr[r]), 0);
            }
            unsigned d[4];
            if constexpr (ABL & 128) {         // variant 128: the integer requantisation in passes of independent instructions
                const unsigned hm1 = (1u << (n - 1)) - 1u;
                int t[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("v_bfe_u32 %0, %1, %2, 1" : "=v"(t[r]) : "v"(rr[r]), "s"(n));
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(t[r]) : "v"(rr[r]), "s"(hm1));
                unsigned h[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "=v"(h[q]) : "v"(t[2 * q]), "v"(t[2 * q + 1]), "s"(n));
#pragma unroll
                for (int g = 0; g < 4; ++g) d[g] = __builtin_amdgcn_perm(h[2 * g + 1], h[2 * g], 0x05040100u) ^ 0x80808080u;
            } else
#pragma unroll
            for (int g = 0; g < 4; ++g) d[g] = (ABL & 4) ? (unsigned)(rr[4 * g] ^ rr[4 * g + 3]) : (FLT ? rq4_flt(rr[4 * g], rr[4 * g + 1], rr[4 * g + 2], rr[4 * g + 3], sc) : rq4_int(rr[4 * g], rr[4 * g + 1], rr[4 * g + 2], rr[4 * g + 3], n)) ^ 0x80808080u;
            v4i o;
            if constexpr (ABL & 8) o = v4i{(int)d[0], (int)d[1], (int)d[2], (int)d[3]};      // ablation 8: no lane swap
            else {
            auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
            o = v4i{(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]}; }
            *(v4i*)(xdst + t * 32 * 272) = o;
        };
        if constexpr (!PIPE) {
#pragma unroll
            for (int t = 0; t < TM; ++t) if (t < mine) { const v16i acc = mm(t); fin(t, acc); }
        } else {
            v16i a0 = mm(0), a1;
#pragma unroll
            for (int t = 0; t < TM; ++t) if (t < mine) {
                v16i& cur = (t & 1) ? a1 : a0; v16i& nxt = (t & 1) ? a0 : a1;
                if (t + 1 < mine) nxt = mm(t + 1);
                fin(t, cur);
            }
        }
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    int s = 0; for (int t = 0; t < TM; ++t) for (int r = 0; r < 16; ++r) s += res[t][r];
    if (s == 0x7fffffff) out[0] = s;
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int NW, int TA, int TB, bool FLT, bool PIPE, int ABL = 0>
static void run(const char* name, int iters) {
    int* out; unsigned long long* cyc;
    CK(hipMalloc((void**)&out, 64)); CK(hipMalloc((void**)&cyc, 256 * 16 * 8)); CK(hipMemset(cyc, 0, 256 * 16 * 8));
    auto k = p3_kernel<NW, TA, TB, FLT, PIPE, ABL>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(256), dim3(NW * 64), 150 * 1024, 0, 4, 8, 0, out, cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(256), dim3(NW * 64), 150 * 1024, 0, iters, 8, 0, out, cyc);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    static unsigned long long h[256 * 16]; CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
    double mx = 0, a0 = 0, a1 = 0;
    for (int b = 0; b < 256; ++b) { double m = 0; for (int w = 0; w < NW; ++w) m = h[b * 16 + w] > m ? h[b * 16 + w] : m; mx += m; a0 += h[b * 16]; a1 += h[b * 16 + NW - 1]; }
    printf("%-34s %2d waves (%d / %d units)  %8.0f cycles per block-phase (slowest wave; wave 0 %6.0f, last wave %6.0f)   %7.2f us per %d phases\n", name, NW, TA, TB,
           mx / 256 / iters, a0 / 256 / iters, a1 / 256 / iters, ms * 1e3, iters);
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
    const int it = 200;
    run<8, 7, 7, false, false>("8 x 7 int", it);
    run<8, 7, 7, true, false>("8 x 7 float", it);
    run<8, 7, 7, false, true>("8 x 7 int, pipelined", it);
    run<8, 7, 7, true, true>("8 x 7 float, pipelined", it);
    run<16, 4, 3, false, false>("16 x (4, 3) int", it);
    run<16, 4, 3, true, false>("16 x (4, 3) float", it);
    run<16, 4, 3, false, true>("16 x (4, 3) int, pipelined", it);
    run<16, 2, 5, false, false>("16 x (2, 5) int (asymmetric)", it);
    run<16, 2, 5, false, true>("16 x (2, 5) int, pipelined", it);
    run<12, 5, 4, false, false>("12 x (5, 4) int", it);
    run<12, 5, 4, false, true>("12 x (5, 4) int, pipelined", it);
    printf("-- ablations (8 waves x 7 units, integer requantisation unless stated)\n");
    run<8, 7, 7, false, false, 32>("bias in registers (C operand)", it);
    run<8, 7, 7, false, true, 32>("bias in registers, pipelined", it);
    run<8, 7, 7, true, true, 32>("float, bias in regs, pipelined", it);
    run<8, 7, 7, false, false, 64>("join in two passes", it);
    run<8, 7, 7, false, false, 192>("join + requant in passes", it);
    run<8, 7, 7, false, false, 224>("passes + bias in regs", it);
    run<8, 7, 7, false, true, 224>("passes + bias in regs, pipelined", it);
    run<16, 4, 3, false, false, 192>("16 waves, passes", it);
    run<8, 7, 7, false, false, 1>("no MFMA", it);
    run<8, 7, 7, false, false, 2>("no join", it);
    run<8, 7, 7, false, false, 16>("join = ReLU only (folded into C)", it);
    run<8, 7, 7, false, false, 4>("no requantisation", it);
    run<8, 7, 7, false, false, 8>("no lane swap", it);
    run<8, 7, 7, false, false, 6>("no join, no requantisation", it);
    run<8, 7, 7, false, false, 7>("no MFMA, join, requantisation", it);
    run<8, 7, 7, true, false, 16>("float, join = ReLU only", it);
    run<16, 4, 3, false, false, 16>("16 waves, join = ReLU only", it);
    run<16, 4, 3, false, false, 6>("16 waves, no join, no requant", it);
    return 0;
}
