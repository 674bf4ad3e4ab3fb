// ubench_mfma_hazard.hip — which adjacency of vector code and v_mfma_i32_32x32x32_i8 does gfx950 NOT interlock?  (round 6: f8_cchain.hip's float-converter instance
// returned a few wrong pixels, different from run to run, when the compiler scheduled its epilogue into the last K step.)
// Everything sits in ONE asm block with fixed registers (the compiler pads nothing inside an asm string), 8 waves per workgroup (2 per SIMD), many iterations:
//   WAR_B: PRE independent MFMAs (queue pressure), the MFMA under test reading B = v[44:47], N wait states, then `v_mov_b32 v44, 0` (a write to its B operand)
//   WAR_A: the same with `v_mov_b32 v40, 0` (its A operand)
//   RAW  : the MFMA under test, PRE independent MFMAs behind it, N wait states, then a vector READ of its first result register
// The reference value comes from the same sequence with 64 wait states in front of the hazard instruction.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/ubench_mfma_hazard.hip -o tools/ubench/ubench_mfma_hazard.bin && tools/ubench/ubench_mfma_hazard.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CLOB "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
             "v40","v41","v42","v43","v44","v45","v46","v47", \
             "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
             "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
             "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95", \
             "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111"
#define LOADAB "v_mov_b32 v40, %[a0]\n v_mov_b32 v41, %[a1]\n v_mov_b32 v42, %[a2]\n v_mov_b32 v43, %[a3]\n" \
               "v_mov_b32 v44, %[b0]\n v_mov_b32 v45, %[b1]\n v_mov_b32 v46, %[b2]\n v_mov_b32 v47, %[b3]\n s_nop 7\n"
#define MF(d) "v_mfma_i32_32x32x32_i8 v[" d "], v[40:43], v[44:47], 0\n"
#define PRE0 ""
#define PRE1 MF("64:79")
#define PRE2 MF("64:79") MF("80:95")
#define PRE3 MF("64:79") MF("80:95") MF("96:111")
#define PRE6 MF("64:79") MF("80:95") MF("96:111") MF("64:79") MF("80:95") MF("96:111")
#define LONG "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
#define OUTS : [o0] "=v"(o0), [o1] "=v"(o1) : [a0] "v"(a.x), [a1] "v"(a.y), [a2] "v"(a.z), [a3] "v"(a.w), [b0] "v"(b.x), [b1] "v"(b.y), [b2] "v"(b.z), [b3] "v"(b.w) : CLOB

// mode 0: WAR on B; 1: WAR on A; 2: RAW
#define KERNEL(name, PRE, NOPS, MODE)                                                                                         \
__global__ void __launch_bounds__(512) name(const int4* ab, int* bad, int iters) {                                            \
    const int4 a = ab[threadIdx.x & 63], b = ab[64 + (threadIdx.x & 63)];                                                     \
    int nb = 0;                                                                                                               \
    for (int it = 0; it < iters; ++it) {                                                                                      \
        int o0, o1, r0, r1;                                                                                                   \
        if (MODE == 0) {                                                                                                      \
            asm volatile(LOADAB PRE MF("48:63") NOPS "v_mov_b32 v44, 0\n" LONG "v_mov_b32 %[o0], v48\n v_mov_b32 %[o1], v63\n" OUTS);      \
            asm volatile(LOADAB PRE MF("48:63") LONG "v_mov_b32 v44, 0\n" LONG "v_mov_b32 %[o0], v48\n v_mov_b32 %[o1], v63\n"             \
                         : [o0] "=v"(r0), [o1] "=v"(r1) : [a0] "v"(a.x), [a1] "v"(a.y), [a2] "v"(a.z), [a3] "v"(a.w), [b0] "v"(b.x), [b1] "v"(b.y), [b2] "v"(b.z), [b3] "v"(b.w) : CLOB); \
        } else if (MODE == 1) {                                                                                               \
            asm volatile(LOADAB PRE MF("48:63") NOPS "v_mov_b32 v40, 0\n" LONG "v_mov_b32 %[o0], v48\n v_mov_b32 %[o1], v63\n" OUTS);      \
            asm volatile(LOADAB PRE MF("48:63") LONG "v_mov_b32 v40, 0\n" LONG "v_mov_b32 %[o0], v48\n v_mov_b32 %[o1], v63\n"             \
                         : [o0] "=v"(r0), [o1] "=v"(r1) : [a0] "v"(a.x), [a1] "v"(a.y), [a2] "v"(a.z), [a3] "v"(a.w), [b0] "v"(b.x), [b1] "v"(b.y), [b2] "v"(b.z), [b3] "v"(b.w) : CLOB); \
        } else if (MODE == 3) {                                                                                               \
            asm volatile(LOADAB "v_mov_b32 v48, -1\n v_mov_b32 v63, -1\n s_nop 7\n" MF("64:79") MF("80:95") MF("96:111") MF("16:31") MF("64:79") MF("80:95") MF("48:63") PRE NOPS "v_mov_b32 %[o0], v48\n v_mov_b32 %[o1], v63\n" LONG LONG LONG OUTS);      \
            asm volatile(LOADAB "v_mov_b32 v48, -1\n v_mov_b32 v63, -1\n s_nop 7\n" MF("64:79") MF("80:95") MF("96:111") MF("16:31") MF("64:79") MF("80:95") MF("48:63") PRE LONG LONG LONG LONG "v_mov_b32 %[o0], v48\n v_mov_b32 %[o1], v63\n" LONG            \
                         : [o0] "=v"(r0), [o1] "=v"(r1) : [a0] "v"(a.x), [a1] "v"(a.y), [a2] "v"(a.z), [a3] "v"(a.w), [b0] "v"(b.x), [b1] "v"(b.y), [b2] "v"(b.z), [b3] "v"(b.w) : CLOB); \
        } else {                                                                                                              \
            asm volatile(LOADAB "v_mov_b32 v48, -1\n v_mov_b32 v63, -1\n s_nop 7\n" MF("48:63") PRE NOPS "v_mov_b32 %[o0], v48\n v_mov_b32 %[o1], v63\n" LONG OUTS);      \
            asm volatile(LOADAB "v_mov_b32 v48, -1\n v_mov_b32 v63, -1\n s_nop 7\n" MF("48:63") PRE LONG "v_mov_b32 %[o0], v48\n v_mov_b32 %[o1], v63\n" LONG            \
                         : [o0] "=v"(r0), [o1] "=v"(r1) : [a0] "v"(a.x), [a1] "v"(a.y), [a2] "v"(a.z), [a3] "v"(a.w), [b0] "v"(b.x), [b1] "v"(b.y), [b2] "v"(b.z), [b3] "v"(b.w) : CLOB); \
        }                                                                                                                     \
        nb += (o0 != r0) || (o1 != r1);                                                                                       \
    }                                                                                                                         \
    if (nb) atomicAdd(bad, nb);                                                                                               \
}

#define N0 ""
#define N1 "s_nop 0\n"
#define N2 "s_nop 1\n"
#define N4 "s_nop 3\n"
#define N8 "s_nop 7\n"
#define N16 "s_nop 15\n"
#define FAMILY(mode, tag)                                                                                                     \
    KERNEL(k_##tag##_p0_n0, PRE0, N0, mode) KERNEL(k_##tag##_p0_n1, PRE0, N1, mode) KERNEL(k_##tag##_p0_n2, PRE0, N2, mode) KERNEL(k_##tag##_p0_n4, PRE0, N4, mode) KERNEL(k_##tag##_p0_n8, PRE0, N8, mode) KERNEL(k_##tag##_p0_n16, PRE0, N16, mode) \
    KERNEL(k_##tag##_p3_n0, PRE3, N0, mode) KERNEL(k_##tag##_p3_n1, PRE3, N1, mode) KERNEL(k_##tag##_p3_n2, PRE3, N2, mode) KERNEL(k_##tag##_p3_n4, PRE3, N4, mode) KERNEL(k_##tag##_p3_n8, PRE3, N8, mode) KERNEL(k_##tag##_p3_n16, PRE3, N16, mode) \
    KERNEL(k_##tag##_p6_n0, PRE6, N0, mode) KERNEL(k_##tag##_p6_n1, PRE6, N1, mode) KERNEL(k_##tag##_p6_n2, PRE6, N2, mode) KERNEL(k_##tag##_p6_n4, PRE6, N4, mode) KERNEL(k_##tag##_p6_n8, PRE6, N8, mode) KERNEL(k_##tag##_p6_n16, PRE6, N16, mode)
FAMILY(0, warb)
FAMILY(1, wara)
FAMILY(2, raw)
KERNEL(k_raw_p1_n0, PRE1, N0, 2) KERNEL(k_raw_p1_n1, PRE1, N1, 2) KERNEL(k_raw_p1_n2, PRE1, N2, 2) KERNEL(k_raw_p1_n4, PRE1, N4, 2) KERNEL(k_raw_p1_n8, PRE1, N8, 2) KERNEL(k_raw_p1_n16, PRE1, N16, 2)
KERNEL(k_bl_p0_n0, PRE0, N0, 3) KERNEL(k_bl_p0_n1, PRE0, N1, 3) KERNEL(k_bl_p0_n2, PRE0, N2, 3) KERNEL(k_bl_p0_n4, PRE0, N4, 3) KERNEL(k_bl_p0_n8, PRE0, N8, 3) KERNEL(k_bl_p0_n16, PRE0, N16, 3)
KERNEL(k_bl_p1_n0, PRE1, N0, 3) KERNEL(k_bl_p1_n1, PRE1, N1, 3) KERNEL(k_bl_p1_n2, PRE1, N2, 3) KERNEL(k_bl_p1_n4, PRE1, N4, 3) KERNEL(k_bl_p1_n8, PRE1, N8, 3) KERNEL(k_bl_p1_n16, PRE1, N16, 3)
KERNEL(k_raw_p2_n0, PRE2, N0, 2) KERNEL(k_raw_p2_n1, PRE2, N1, 2) KERNEL(k_raw_p2_n2, PRE2, N2, 2) KERNEL(k_raw_p2_n4, PRE2, N4, 2) KERNEL(k_raw_p2_n8, PRE2, N8, 2) KERNEL(k_raw_p2_n16, PRE2, N16, 2)

typedef void (*kfn)(const int4*, int*, int);
int main() {
    std::vector<int> h(128 * 4);
    srand(7);
    for (auto& v : h) v = rand() ^ (rand() << 16);
    int4* d; int* bad;
    hipMalloc((void**)&d, h.size() * 4); hipMalloc((void**)&bad, 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int iters = 2000, blocks = 256;
    struct { const char* name; kfn f[18]; } fam[3] = {
        {"WAR on SrcB (vector write to the B operand N wait states behind the MFMA)", {k_warb_p0_n0, k_warb_p0_n1, k_warb_p0_n2, k_warb_p0_n4, k_warb_p0_n8, k_warb_p0_n16, k_warb_p3_n0, k_warb_p3_n1, k_warb_p3_n2, k_warb_p3_n4, k_warb_p3_n8, k_warb_p3_n16, k_warb_p6_n0, k_warb_p6_n1, k_warb_p6_n2, k_warb_p6_n4, k_warb_p6_n8, k_warb_p6_n16}},
        {"WAR on SrcA", {k_wara_p0_n0, k_wara_p0_n1, k_wara_p0_n2, k_wara_p0_n4, k_wara_p0_n8, k_wara_p0_n16, k_wara_p3_n0, k_wara_p3_n1, k_wara_p3_n2, k_wara_p3_n4, k_wara_p3_n8, k_wara_p3_n16, k_wara_p6_n0, k_wara_p6_n1, k_wara_p6_n2, k_wara_p6_n4, k_wara_p6_n8, k_wara_p6_n16}},
        {"RAW (vector read of the result N wait states behind the LAST of PRE independent MFMAs that follow the MFMA under test)", {k_raw_p0_n0, k_raw_p0_n1, k_raw_p0_n2, k_raw_p0_n4, k_raw_p0_n8, k_raw_p0_n16, k_raw_p3_n0, k_raw_p3_n1, k_raw_p3_n2, k_raw_p3_n4, k_raw_p3_n8, k_raw_p3_n16, k_raw_p6_n0, k_raw_p6_n1, k_raw_p6_n2, k_raw_p6_n4, k_raw_p6_n8, k_raw_p6_n16}}};
    const int pre[3] = {0, 3, 6}, nops[6] = {0, 1, 2, 4, 8, 16};
    for (auto& F : fam) {
        printf("%s\n  wrong results in %d waves x %d iterations (8 waves per workgroup):\n", F.name, blocks * 8, iters);
        for (int p = 0; p < 3; ++p) {
            printf("  %d independent MFMAs %s:", pre[p], &F == &fam[2] ? "behind it" : "in front");
            for (int n = 0; n < 6; ++n) {
                hipMemset(bad, 0, 4);
                hipLaunchKernelGGL(F.f[p * 6 + n], dim3(blocks), dim3(512), 0, 0, d, bad, iters);
                int hb = -1; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
                printf("  N=%d: %d", nops[n], hb);
            }
            printf("\n");
        }
    }
    {
        kfn extra[2][6] = {{k_raw_p1_n0, k_raw_p1_n1, k_raw_p1_n2, k_raw_p1_n4, k_raw_p1_n8, k_raw_p1_n16}, {k_raw_p2_n0, k_raw_p2_n1, k_raw_p2_n2, k_raw_p2_n4, k_raw_p2_n8, k_raw_p2_n16}};
        for (int p = 0; p < 2; ++p) {
            printf("  RAW, %d independent MFMA(s) behind it:", p + 1);
            for (int n = 0; n < 6; ++n) {
                hipMemset(bad, 0, 4);
                hipLaunchKernelGGL(extra[p][n], dim3(blocks), dim3(512), 0, 0, d, bad, iters);
                int hb = -1; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
                printf("  N=%d: %d", nops[n], hb);
            }
            printf("\n");
        }
    }
    {
        kfn extra[2][6] = {{k_bl_p0_n0, k_bl_p0_n1, k_bl_p0_n2, k_bl_p0_n4, k_bl_p0_n8, k_bl_p0_n16}, {k_bl_p1_n0, k_bl_p1_n1, k_bl_p1_n2, k_bl_p1_n4, k_bl_p1_n8, k_bl_p1_n16}};
        for (int p = 0; p < 2; ++p) {
            printf("  RAW with SIX independent MFMAs IN FRONT of the MFMA under test, %d behind it:", p);
            for (int n = 0; n < 6; ++n) {
                hipMemset(bad, 0, 4);
                hipLaunchKernelGGL(extra[p][n], dim3(blocks), dim3(512), 0, 0, d, bad, iters);
                int hb = -1; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
                printf("  N=%d: %d", nops[n], hb);
            }
            printf("\n");
        }
    }
    hipError_t e = hipDeviceSynchronize();
    printf("%s\n", e == hipSuccess ? "done" : hipGetErrorString(e));
    return 0;
}
