// Probe: is  v_cvt_pk_u8_f32(v_cvt_f32_i32(v) * 2^-n)  ==  clamp(round_half_even(v / 2^n), 0, 255)  for every int32 v?
// (the ReLU -> unsigned 8-bit requantisation of models/fix_quant_ops.py:99-112 in 3 vector operations, packing included)
// The reference here is the NON-wrapping quotient: the library uses the float form only where `v + 2^(n-1)` cannot wrap (bounded conv
// accumulators, f8_net.cpp conv_acc_bounded).  Built by f8net_amd/csrc/build.sh -> tools/ubench/cvt_u8_probe; run by tests/test_gpu_requant_probe.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline int ref_requant(int v, int n) {
    long long x = v;
    long long half = 1ll << (n - 1), mask = (1ll << n) - 1;
    long long q = (x + half) >> n;
    if (((x & mask) == half)) q &= ~1ll;          // tie -> even
    return (int)(q < 0 ? 0 : q > 255 ? 255 : q);
}
// the reference's int32 arithmetic, wrap included (fix_quant_ops.py:100-104): s = v + 2^(n-1) (wraps), q = s >> n, LSB cleared on a tie
__device__ inline int ref_requant_wrap(int v, int n) {
    const int s = (int)((unsigned)v + (1u << (n - 1)));
    int q = s >> n;
    if ((s & (int)((1u << n) - 1u)) == 0) q &= ~1;
    return q < 0 ? 0 : q > 255 ? 255 : q;
}
// mode 0: three operations against the non-wrapping quotient; mode 1: FOUR operations — v_add_u32 (v + 2^(n-1), wrapping like the
// reference), v_cvt_f32_i32, v_fma_f32 (x 2^-n, - 0.5), v_cvt_pk_u8_f32 — against the WRAPPING reference
__global__ void probe(int mode, int n, long long lo, long long count, unsigned long long* bad, int* first) {
    __builtin_amdgcn_s_setreg(1 | (0 << 6) | (3 << 11), 0);        // what every kernel with a float form does first (f8_device.h: set_fp_round_nearest_even)
    const float scale = __builtin_ldexpf(1.0f, -n);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(lo + i);
        int got, want;
        if (mode == 0) {
            got = (int)__builtin_amdgcn_cvt_pk_u8_f32((float)v * scale, 0u, 0u);
            want = ref_requant(v, n);
        } else {
            const int s = (int)((unsigned)v + (1u << (n - 1)));
            got = (int)__builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf((float)s, scale, -0.5f), 0u, 0u);
            want = ref_requant_wrap(v, n);
        }
        if (got != want) { if (atomicAdd(bad, 1ull) == 0) *first = v; }
    }
}
// `cvt_u8_probe.bin smoke`: shifts 1, 8, 16 only (both modes, every int32 value) — what __graft_entry__.smoke() runs
int main(int argc, char** argv) {
    const bool quick = argc > 1;
    unsigned long long* bad; int* first;
    if (hipMalloc(&bad, 8) != hipSuccess || hipMalloc(&first, 4) != hipSuccess) return 2;
    int rc = 0;
    for (int mode = 0; mode < 2; ++mode)
    for (int n = 1; n <= 20; ++n) {
        if (quick && n != 1 && n != 8 && n != 16) continue;
        unsigned long long total = 0; int f = 0;
        // every int32 value
        if (hipMemset(bad, 0, 8) != hipSuccess || hipMemset(first, 0, 4) != hipSuccess) return 2;
        hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, mode, n, -2147483648ll, 4294967296ll, bad, first);
        if (hipDeviceSynchronize() != hipSuccess) return 2;
        if (hipMemcpy(&total, bad, 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        printf("mode=%d n=%2d mismatches=%llu first=%d\n", mode, n, total, f);
        if (n <= 16 && total != 0) rc = 1;           // kRequantU8MaxShift (f8_device.h): the shifts the library uses this form for
    }
    return rc;
}
