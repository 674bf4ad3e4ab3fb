// Microbenchmark (gfx950): issue rate of the vector instructions the epilogues are made of, per wave64 instruction and SIMD, with 16 independent
// dependency chains per wave and 1 / 2 waves per SIMD.  Shader-clock cycles from s_memtime (__builtin_readcyclecounter), cross-checked by events.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/ubench_ops.hip -o tools/ubench/ubench_ops.bin && tools/ubench/ubench_ops.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define OPK(NAME, ASM)                                                                                                        \
    __global__ void __launch_bounds__(512) NAME(int iters, int sn, int* out, unsigned long long* cyc) {                         \
        int v[16]; int w = threadIdx.x * 3 + 1, z = sn;                                                                        \
        for (int c = 0; c < 16; ++c) v[c] = threadIdx.x * 77 + c * 1315423911;                                                 \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                            \
        for (int i = 0; i < iters; ++i) {                                                                                      \
            _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                                      \
            _Pragma("unroll") for (int c = 0; c < 16; ++c) asm volatile(ASM : "+v"(v[c]) : "v"(w), "v"(z), "s"(sn));          \
        }                                                                                                                      \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                            \
        int s = 0; for (int c = 0; c < 16; ++c) s += v[c];                                                                     \
        if (s == 0x7fffffff) out[0] = s;                                                                                       \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;                                        \
    }
OPK(k_lshl_add, "v_lshl_add_u32 %0, %0, %3, %1")
OPK(k_max_i32, "v_max_i32 %0, %0, %1")
OPK(k_add_u32, "v_add_u32 %0, %0, %1")
OPK(k_xor, "v_xor_b32 %0, %0, %1")
OPK(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
OPK(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
OPK(k_mul_f32, "v_mul_f32 %0, %0, %1")
OPK(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
OPK(k_cvt_pk_u8, "v_cvt_pk_u8_f32 %0, %1, 1, %0")
OPK(k_bfe, "v_bfe_u32 %0, %0, %3, 1")
OPK(k_add3, "v_add3_u32 %0, %0, %1, %2")
OPK(k_ashr, "v_ashrrev_i32 %0, %3, %0")
OPK(k_med3, "v_med3_i32 %0, %0, %1, %2")
OPK(k_perm, "v_perm_b32 %0, %0, %1, %2")
OPK(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
OPK(k_mad_i24, "v_mad_i32_i24 %0, %0, %1, %2")
OPK(k_sat_pk_u8_i16, "v_sat_pk_u8_i16 %0, %0")
OPK(k_pk_add_i16, "v_pk_add_i16 %0, %0, %1")
OPK(k_pk_max_i16, "v_pk_max_i16 %0, %0, %1")
OPK(k_pk_ashr_i16, "v_pk_ashrrev_i16 %0, %1, %0")
OPK(k_alignbit, "v_alignbit_b32 %0, %0, %1, %3")
OPK(k_mov_dpp, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
OPK(k_add_dpp, "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
OPK(k_cvt_pk_i16_i32, "v_cvt_pk_i16_i32 %0, %0, %1")
OPK(k_lshl_or, "v_lshl_or_b32 %0, %0, %3, %1")
OPK(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
OPK(k_permlane_swap, "v_permlane32_swap_b32 %0, %0")
OPK(k_ashr_pk_u8, "v_ashr_pk_u8_i32 %0, %0, %1, %3")

// 64-bit (register pair) packed fp32
__global__ void __launch_bounds__(512) k_pk_mul(int iters, int sn, int* out, unsigned long long* cyc) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f v[8]; v2f w = {1.0001f, 0.9999f};
    for (int c = 0; c < 8; ++c) v[c] = v2f{(float)threadIdx.x + c, (float)c};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[c]) : "v"(w));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int c = 0; c < 8; ++c) s += v[c][0] + v[c][1];
    if (s == 12345.f) out[0] = 1;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
__global__ void __launch_bounds__(512) k_pk_fma(int iters, int sn, int* out, unsigned long long* cyc) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f v[8]; v2f w = {1.0001f, 0.9999f}, z = {-0.5f, -0.5f};
    for (int c = 0; c < 8; ++c) v[c] = v2f{(float)threadIdx.x + c, (float)c};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(w), "v"(z));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int c = 0; c < 8; ++c) s += v[c][0] + v[c][1];
    if (s == 12345.f) out[0] = 1;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*kern_t)(int, int, int*, unsigned long long*);
static void run(const char* name, kern_t k) {
    static unsigned long long* d = nullptr; static int* o = nullptr;
    if (!d) { CK(hipMalloc(&d, 8 * 8 * 256)); CK(hipMalloc(&o, 4)); }
    const int iters = 4096;
    double res[2];
    for (int t = 0; t < 2; ++t) {
        const int threads = t ? 512 : 256;
        hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, 64, 3, o, d);
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, iters, 3, o, d);
        CK(hipDeviceSynchronize());
        unsigned long long h[8 * 256]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
        double avg = 0; const int nw = threads / 64; for (int i = 0; i < 256; ++i) for (int w = 0; w < nw; ++w) avg += (double)h[i * 8 + w]; avg /= 256.0 * nw;
        res[t] = avg / iters / 64.0 * (t ? 0.5 : 1.0);        // cycles per instruction per SIMD (two waves share it)
    }
    printf("%-22s %6.2f cycles / wave64 instruction (1 wave per SIMD)   %6.2f (2 waves per SIMD)\n", name, res[0], res[1]);
}
int main() {
#define R(k) run(#k, k)
    R(k_lshl_add); R(k_max_i32); R(k_add_u32); R(k_xor); R(k_cvt_f32_i32); R(k_cvt_i32_f32); R(k_mul_f32); R(k_fma_f32); R(k_cvt_pk_u8); R(k_bfe); R(k_add3); R(k_ashr); R(k_med3);
    R(k_perm); R(k_mul_lo); R(k_mad_i24); R(k_sat_pk_u8_i16); R(k_pk_add_i16); R(k_pk_max_i16); R(k_pk_ashr_i16); R(k_alignbit); R(k_mov_dpp); R(k_add_dpp); R(k_cvt_pk_i16_i32);
    R(k_lshl_or); R(k_and_or); R(k_permlane_swap); R(k_ashr_pk_u8); R(k_pk_mul); R(k_pk_fma);
    printf("(s_memtime counts at a fixed 100 MHz on some parts: compare rows, and the v_fma_f32 row with the guide's 2 cycles)\n");
    return 0;
}
