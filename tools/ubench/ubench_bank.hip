// Microbenchmark (gfx950): does the register BANK (index mod 4) of a vector instruction's VGPR sources matter?  Same instruction, explicit registers.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/ubench_bank.hip -o tools/ubench/ubench_bank.bin && tools/ubench/ubench_bank.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define REP16(X) X X X X X X X X X X X X X X X X
#define KERNEL(NAME, BODY)                                                                                          \
    __global__ void __launch_bounds__(512) NAME(int iters, int sn, unsigned long long* cyc) {                       \
        asm volatile("v_mov_b32 v100, 1\n v_mov_b32 v101, 2\n v_mov_b32 v102, 3\n v_mov_b32 v103, 4\n v_mov_b32 v104, 5\n v_mov_b32 v105, 6\n v_mov_b32 v106, 7\n v_mov_b32 v107, 8\n" \
                     "v_mov_b32 v108, 1\n v_mov_b32 v109, 2\n v_mov_b32 v110, 3\n v_mov_b32 v111, 4\n v_mov_b32 v112, 5\n v_mov_b32 v113, 6\n v_mov_b32 v114, 7\n v_mov_b32 v115, 8\n" \
                     ::: "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123"); \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                \
        for (int i = 0; i < iters; ++i) {                                                                           \
            asm volatile(REP16(BODY) :: "s"(sn) : "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123"); \
        }                                                                                                           \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;                            \
    }
// 8 independent instructions per BODY (x16 = 128 per iteration); destination v116..v123
// lshl_add: sources in the SAME bank (v100 & v104 ...: both = 0 mod 4) vs DIFFERENT banks (v100 & v105)
KERNEL(k_lshl_same, "v_lshl_add_u32 v116, v100, %0, v104\n v_lshl_add_u32 v117, v101, %0, v105\n v_lshl_add_u32 v118, v102, %0, v106\n v_lshl_add_u32 v119, v103, %0, v107\n v_lshl_add_u32 v120, v108, %0, v112\n v_lshl_add_u32 v121, v109, %0, v113\n v_lshl_add_u32 v122, v110, %0, v114\n v_lshl_add_u32 v123, v111, %0, v115\n")
KERNEL(k_lshl_diff, "v_lshl_add_u32 v116, v100, %0, v105\n v_lshl_add_u32 v117, v101, %0, v106\n v_lshl_add_u32 v118, v102, %0, v107\n v_lshl_add_u32 v119, v103, %0, v104\n v_lshl_add_u32 v120, v108, %0, v113\n v_lshl_add_u32 v121, v109, %0, v114\n v_lshl_add_u32 v122, v110, %0, v115\n v_lshl_add_u32 v123, v111, %0, v112\n")
// in place (destination = a source), as the join does: rr = (acc << s) + rr
KERNEL(k_lshl_inplace_same, "v_lshl_add_u32 v104, v100, %0, v104\n v_lshl_add_u32 v105, v101, %0, v105\n v_lshl_add_u32 v106, v102, %0, v106\n v_lshl_add_u32 v107, v103, %0, v107\n v_lshl_add_u32 v112, v108, %0, v112\n v_lshl_add_u32 v113, v109, %0, v113\n v_lshl_add_u32 v114, v110, %0, v114\n v_lshl_add_u32 v115, v111, %0, v115\n")
KERNEL(k_lshl_inplace_diff, "v_lshl_add_u32 v105, v100, %0, v105\n v_lshl_add_u32 v106, v101, %0, v106\n v_lshl_add_u32 v107, v102, %0, v107\n v_lshl_add_u32 v104, v103, %0, v104\n v_lshl_add_u32 v113, v108, %0, v113\n v_lshl_add_u32 v114, v109, %0, v114\n v_lshl_add_u32 v115, v110, %0, v115\n v_lshl_add_u32 v112, v111, %0, v112\n")
KERNEL(k_max, "v_max_i32 v116, 0, v100\n v_max_i32 v117, 0, v101\n v_max_i32 v118, 0, v102\n v_max_i32 v119, 0, v103\n v_max_i32 v120, 0, v104\n v_max_i32 v121, 0, v105\n v_max_i32 v122, 0, v106\n v_max_i32 v123, 0, v107\n")
KERNEL(k_add3_same, "v_add3_u32 v116, v100, %0, v104\n v_add3_u32 v117, v101, %0, v105\n v_add3_u32 v118, v102, %0, v106\n v_add3_u32 v119, v103, %0, v107\n v_add3_u32 v120, v108, %0, v112\n v_add3_u32 v121, v109, %0, v113\n v_add3_u32 v122, v110, %0, v114\n v_add3_u32 v123, v111, %0, v115\n")
KERNEL(k_add3_diff, "v_add3_u32 v116, v100, %0, v105\n v_add3_u32 v117, v101, %0, v106\n v_add3_u32 v118, v102, %0, v107\n v_add3_u32 v119, v103, %0, v104\n v_add3_u32 v120, v108, %0, v113\n v_add3_u32 v121, v109, %0, v114\n v_add3_u32 v122, v110, %0, v115\n v_add3_u32 v123, v111, %0, v112\n")
KERNEL(k_bfe, "v_bfe_u32 v116, v100, %0, 1\n v_bfe_u32 v117, v101, %0, 1\n v_bfe_u32 v118, v102, %0, 1\n v_bfe_u32 v119, v103, %0, 1\n v_bfe_u32 v120, v104, %0, 1\n v_bfe_u32 v121, v105, %0, 1\n v_bfe_u32 v122, v106, %0, 1\n v_bfe_u32 v123, v107, %0, 1\n")
KERNEL(k_ashrpk_same, "v_ashr_pk_u8_i32 v116, v100, v104, %0\n v_ashr_pk_u8_i32 v117, v101, v105, %0\n v_ashr_pk_u8_i32 v118, v102, v106, %0\n v_ashr_pk_u8_i32 v119, v103, v107, %0\n v_ashr_pk_u8_i32 v120, v108, v112, %0\n v_ashr_pk_u8_i32 v121, v109, v113, %0\n v_ashr_pk_u8_i32 v122, v110, v114, %0\n v_ashr_pk_u8_i32 v123, v111, v115, %0\n")
KERNEL(k_ashrpk_diff, "v_ashr_pk_u8_i32 v116, v100, v105, %0\n v_ashr_pk_u8_i32 v117, v101, v106, %0\n v_ashr_pk_u8_i32 v118, v102, v107, %0\n v_ashr_pk_u8_i32 v119, v103, v104, %0\n v_ashr_pk_u8_i32 v120, v108, v113, %0\n v_ashr_pk_u8_i32 v121, v109, v114, %0\n v_ashr_pk_u8_i32 v122, v110, v115, %0\n v_ashr_pk_u8_i32 v123, v111, v112, %0\n")
KERNEL(k_add_vop2, "v_add_u32 v116, v100, v105\n v_add_u32 v117, v101, v106\n v_add_u32 v118, v102, v107\n v_add_u32 v119, v103, v104\n v_add_u32 v120, v108, v113\n v_add_u32 v121, v109, v114\n v_add_u32 v122, v110, v115\n v_add_u32 v123, v111, v112\n")
KERNEL(k_add_vop2_same, "v_add_u32 v116, v100, v104\n v_add_u32 v117, v101, v105\n v_add_u32 v118, v102, v106\n v_add_u32 v119, v103, v107\n v_add_u32 v120, v108, v112\n v_add_u32 v121, v109, v113\n v_add_u32 v122, v110, v114\n v_add_u32 v123, v111, v115\n")

template <typename K> static void run(const char* name, K k, int threads) {
    unsigned long long* cyc; CK(hipMalloc((void**)&cyc, 256 * 8 * 8)); CK(hipMemset(cyc, 0, 256 * 8 * 8));
    const int iters = 2000;
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, 10, 3, cyc); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, iters, 3, cyc); CK(hipDeviceSynchronize());
    static unsigned long long h[256 * 8]; CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
    double s = 0; int n = 0; const int nw = threads / 64;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) { s += h[b * 8 + w]; ++n; }
    const double per_wave = s / n / (iters * 128.0);                 // cycles between two instructions of ONE wave
    printf("%-28s %d waves/SIMD: %.2f cycles per instruction and wave = %.2f per instruction and SIMD\n", name, nw / 4, per_wave, per_wave / (nw / 4));
    CK(hipFree(cyc));
}
int main() {
    for (int t : {256, 512}) {
        run("v_add_u32 (diff banks)", k_add_vop2, t); run("v_add_u32 (same bank)", k_add_vop2_same, t);
        run("v_max_i32 0, v", k_max, t); run("v_bfe_u32 v, s, 1", k_bfe, t);
        run("v_lshl_add_u32 diff banks", k_lshl_diff, t); run("v_lshl_add_u32 same bank", k_lshl_same, t);
        run("v_lshl_add in place, diff", k_lshl_inplace_diff, t); run("v_lshl_add in place, same", k_lshl_inplace_same, t);
        run("v_add3_u32 v,s,v diff", k_add3_diff, t); run("v_add3_u32 v,s,v same", k_add3_same, t);
        run("v_ashr_pk_u8_i32 diff", k_ashrpk_diff, t); run("v_ashr_pk_u8_i32 same", k_ashrpk_same, t);
    }
    return 0;
}
