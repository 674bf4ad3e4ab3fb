// Probe (gfx950), every int32 value against the reference's arithmetic (models/fix_quant_ops.py:99-112, ReLU -> unsigned 8-bit: clamp [0, 255]):
//   mode 0  requant_u8x4 (f8_device.h): v_cvt_pk_u8_f32(v_cvt_f32_i32(v) * 2^-n) — three vector operations, packing included — against the NON-wrapping
//           quotient clamp(round_half_even(v / 2^n)): the library uses this form only where `v + 2^(n-1)` cannot wrap (planner-bounded values:
//           f8_net.cpp conv_acc_bounded / tensor_amax) and n <= 16; shifts 1 .. 20 (17 .. 20 must differ: kRequantU8MaxShift);
//   mode 1  requant_u8x4_int (f8_device.h): t = v + (2^(n-1) - 1) + bit n of v (v_bfe_u32, v_add3_u32), two values per v_ashr_pk_u8_i32 (arithmetic
//           shift, saturate to [0, 255], pack) — INTEGER only — against the reference's WRAPPING int32 arithmetic, in both operand positions;
//           shifts 1 .. 30.
// Built by f8net_amd/csrc/build.sh -> tools/ubench/cvt_u8_probe.bin; run by tests/test_gpu_requant_probe.py and (`smoke`) by __graft_entry__.smoke().
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
__device__ inline int round_term(int v, int n) { return (int)((unsigned)v + ((1u << (n - 1)) - 1u) + __builtin_amdgcn_ubfe((unsigned)v, (unsigned)n, 1u)); }
__global__ void probe(int mode, int n, long long lo, long long count, unsigned long long* bad, int* first) {
    __builtin_amdgcn_s_setreg(1 | (0 << 6) | (3 << 11), 0);        // what every kernel with a float form does first (f8_device.h: set_fp_round_nearest_even)
    const float scale = __builtin_ldexpf(1.0f, -n);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(lo + i);
        bool ok;
        if (mode == 0) {
            ok = (int)__builtin_amdgcn_cvt_pk_u8_f32((float)v * scale, 0u, 0u) == ref_requant(v, n);
        } else {
            const int u = (int)((unsigned)v * 2654435761u + 12345u);          // a second, unrelated value in the other operand position
            const unsigned p = (unsigned)(unsigned short)__builtin_amdgcn_ashr_pk_u8_i32(round_term(v, n), round_term(u, n), n);
            const unsigned q = (unsigned)(unsigned short)__builtin_amdgcn_ashr_pk_u8_i32(round_term(u, n), round_term(v, n), n);
            ok = (int)(p & 255u) == ref_requant_wrap(v, n) && (int)(p >> 8) == ref_requant_wrap(u, n) && (int)(q >> 8) == ref_requant_wrap(v, n) && (int)(q & 255u) == ref_requant_wrap(u, n);
        }
        if (!ok) { if (atomicAdd(bad, 1ull) == 0) *first = v; }
        if (v == 123456789) atomicAdd(bad + 1, 1ull);              // proof of life: the sweep got here exactly once
    }
}
// `cvt_u8_probe.bin smoke`: shifts 1, 8, 16 (both modes) and 30 (mode 1) only, every int32 value — what __graft_entry__.smoke() runs
int main(int argc, char** argv) {
    const bool quick = argc > 1;
    unsigned long long* bad; int* first;
    if (hipMalloc(&bad, 16) != hipSuccess || hipMalloc(&first, 4) != hipSuccess) return 2;
    int rc = 0;
    for (int mode = 0; mode < 2; ++mode)
    for (int n = 1; n <= (mode ? 30 : 20); ++n) {
        if (quick && n != 1 && n != 8 && n != 16 && !(mode == 1 && n == 30)) continue;
        unsigned long long total = 0, seen = 0; int f = 0;
        if (hipMemset(bad, 0, 16) != hipSuccess || hipMemset(first, 0, 4) != hipSuccess) return 2;
        hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, mode, n, -2147483648ll, 4294967296ll, bad, first);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { printf("probe launch failed\n"); return 2; }
        if (hipMemcpy(&total, bad, 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&seen, bad + 1, 8, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        if (seen != 1) { printf("mode=%d n=%2d: the sweep did not visit every value once (seen %llu)\n", mode, n, seen); return 2; }
        printf("mode=%d n=%2d mismatches=%llu first=%d\n", mode, n, total, f);
        if ((mode == 1 || n <= 16) && total != 0) rc = 1;          // mode 0 beyond kRequantU8MaxShift (f8_device.h) is EXPECTED to differ: the hosts never select it there
    }
    return rc;
}
