// f8_device.h — device-side helpers shared by the HIP translation units (gfx950 only).
#pragma once
#include "f8_internal.h"
#include <type_traits>
#include <utility>

namespace f8 {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

static constexpr unsigned kOOB = 0x80000000u;   // voffset sentinel: beyond any buffer (< 2 GiB each)

// clamp via v_med3_i32 (lo <= hi)
__device__ __forceinline__ int med3i(int v, int lo, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
}

// int_op_only_fix_quant on one value; n, lo, hi are wave-uniform.
// n > 0 (fix_quant_ops.py:99-104): q = (v + 2^(n-1)) >> n, except on an exact tie (v mod 2^n == 2^(n-1)), where the result
// is ((v + 2^(n-1)) >> (n+1)) << 1, i.e. the even neighbour.  With v = k*2^n + f (k = v >> n arithmetic, 0 <= f < 2^n) both
// cases are   q = (v + (2^(n-1) - 1) + (k & 1)) >> n :
//   the carry into bit n happens iff f + (k&1) >= 2^(n-1) + 1: k even -> f > half (a tie stays at k, even);
//   k odd -> f >= half (a tie goes to k+1, even).  Overflow: v + half and v + half - 1 + (k&1) wrap together (when v + half
//   == 2^31 exactly, f == half and k is odd), and 2^32 is a multiple of 2^n, so the identity also holds mod 2^32.
// Four VALU operations (v_bfe_u32, v_add3_u32, v_ashrrev_i32, v_med3_i32) instead of seven, branch-free; the epilogues of
// the small convolutions are VALU-bound (f8_stem.hip: ~2300 of ~6000 cycles per tile were requant arithmetic).
__device__ __forceinline__ int requant_shr(int v, int n, unsigned half, unsigned /*mask*/, int lo, int hi) {
    const unsigned odd = __builtin_amdgcn_ubfe((unsigned)v, (unsigned)n, 1u);
    const int r = (int)((unsigned)v + (half - 1u) + odd);
    return med3i(r >> n, lo, hi);
}
__device__ __forceinline__ int requant_shl(int v, int n, int lo, int hi) {   // n <= 0
    return med3i((int)((unsigned)v << (-n)), lo, hi);
}
// Either direction, BRANCH-FREE (n is wave-uniform, but an `if (n > 0)` per value survives unrolling as one scalar branch
// per value with register shuffles around it: the stem / fused / patch epilogues spent more time there than in the
// arithmetic).  n > 0: the sum above; n <= 0: v << -n (fix_quant_ops.py:105-106); the selects below are scalar and hoisted.
__device__ __forceinline__ int requant1(int v, int n, int lo, int hi) {
    const int shr = n > 0 ? n : 0, shl = n > 0 ? 0 : -n;
    const unsigned hm1 = n > 0 ? (1u << (shr - (n > 0 ? 1 : 0))) - 1u : 0u;
    const unsigned width = n > 0 ? 1u : 0u;                                  // v_bfe_u32 with width 0 yields 0
    const unsigned odd = __builtin_amdgcn_ubfe((unsigned)v, (unsigned)shr, width);
    const int r = (int)(((unsigned)v << shl) + hm1 + odd);
    return med3i(r >> shr, lo, hi);
}

// ReLU -> UNSIGNED 8-bit by a right shift of 1 <= n <= 16 (fix_quant_ops.py:99-112 with the clamp [0, 255]; the ReLU is the clamp's lower
// bound) in THREE vector operations per value, packing included: v_cvt_f32_i32 (exact below 2^24; above it the quotient saturates whatever
// that rounding did, because n <= 16), v_mul_f32 by 2^-n (exact), v_cvt_pk_u8_f32 (round to nearest EVEN, saturate to [0, 255], insert into
// byte k of the dword).  Compared on the device with the integer form for EVERY int32 value and every n in 1 .. 16: identical
// (tools/ubench/cvt_u8_probe.hip, tests/test_gpu_requant_probe.py; n >= 17 differs from 2^24 on, as predicted) — the hosts select the
// instances that use this only when every shift involved is in 1 .. 16 (kRequantU8MaxShift).
constexpr int kRequantU8MaxShift = 16;
// The float forms rely on ROUND-TO-NEAREST-EVEN in v_cvt_pk_u8_f32 (and v_cvt_f32_i32 above 2^24): that is the power-on state of the MODE register
// and what HIP launches kernels with, but nothing else in the program states it — every kernel that instantiates a float form states it itself,
// first thing (s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 4) = 0: FP_ROUND single / double = nearest even).
__device__ __forceinline__ void set_fp_round_nearest_even() { __builtin_amdgcn_s_setreg(1 | (0 << 6) | (3 << 11), 0); }
__device__ __forceinline__ float requant_u8_scale(int n) { return __builtin_ldexpf(1.0f, -n); }   // wave-uniform, hoisted
__device__ __forceinline__ unsigned requant_u8x4(int a, int b, int c, int d, float scale) {
    unsigned r = __builtin_amdgcn_cvt_pk_u8_f32((float)a * scale, 0u, 0u);
    r = __builtin_amdgcn_cvt_pk_u8_f32((float)b * scale, 1u, r);
    r = __builtin_amdgcn_cvt_pk_u8_f32((float)c * scale, 2u, r);
    return __builtin_amdgcn_cvt_pk_u8_f32((float)d * scale, 3u, r);
}
// The same function for ANY int32 value and any shift 1 .. 30, the reference's wrap included, in INTEGER operations only (round 4, gfx950):
//   t = v + (2^(n-1) - 1) + bit n of v     v_bfe_u32, v_add3_u32 per value (requant_shr's rounding term: wraps exactly where the reference's
//                                          `input + 2^(n-1)` does, a tie goes to the even neighbour)
//   {byte 0, byte 1} = sat_u8(t_a >> n), sat_u8(t_b >> n)     ONE v_ashr_pk_u8_i32 per PAIR (arithmetic shift, saturate to [0, 255], pack)
//   the two halves -> one dword             one v_perm_b32 per four values
// = 2 3/4 operations per value instead of requant_shr + pack4's 4 3/4.  Compared on the device with the reference's wrapping int32 arithmetic for
// EVERY int32 value in both operand positions and every shift 1 .. 30 (tools/ubench/cvt_u8_probe.hip mode 1, tests/test_gpu_requant_probe.py).
// What every FAST / FQ == 2 instance uses: option requant_float = 0, a shift beyond 16, or values the planner cannot bound (conv accumulators:
// conv_acc_bounded; the int32 stream of a chain launch: tensor_amax, f8_net.cpp) — the float form above is never used on an unbounded value.
__device__ __forceinline__ unsigned requant_u8x4_int(int a, int b, int c, int d, int n) {
    const unsigned hm1 = (1u << (n - 1)) - 1u;
    const int ta = (int)((unsigned)a + hm1 + __builtin_amdgcn_ubfe((unsigned)a, (unsigned)n, 1u)), tb = (int)((unsigned)b + hm1 + __builtin_amdgcn_ubfe((unsigned)b, (unsigned)n, 1u));
    const int tc = (int)((unsigned)c + hm1 + __builtin_amdgcn_ubfe((unsigned)c, (unsigned)n, 1u)), td = (int)((unsigned)d + hm1 + __builtin_amdgcn_ubfe((unsigned)d, (unsigned)n, 1u));
    // (as asm: the builtin returns 16 bits and the compiler zero-extends them with a v_and_b32 each before the v_perm_b32 that only takes bytes 0 and 1)
    unsigned lo, hi;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "=v"(lo) : "v"(ta), "v"(tb), "s"(n));
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "=v"(hi) : "v"(tc), "v"(td), "s"(n));
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);                                    // {lo.b0, lo.b1, hi.b0, hi.b1}
}

// q = n / d for a divisor known on the host: q = (t + ((n - t) >> sh1)) >> sh2, t = mulhi(n, magic)
// (round-up method, exact for every 32-bit n; host: f8_net.cpp make_magic)
__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned magic, int sh1, int sh2) {
    const unsigned t = __umulhi(n, magic);
    return (t + ((n - t) >> sh1)) >> sh2;
}

// low bytes of four values -> one dword (a = byte 0): three v_perm_b32 instead of and / and / and / shift-or x3
__device__ __forceinline__ unsigned pack4(int a, int b, int c, int d) {
    const unsigned lo = __builtin_amdgcn_perm((unsigned)b, (unsigned)a, 0x0c0c0400u);   // {a.b0, b.b0, 0, 0}
    const unsigned hi = __builtin_amdgcn_perm((unsigned)d, (unsigned)c, 0x0c0c0400u);   // {c.b0, d.b0, 0, 0}
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);                                    // {lo.b0, lo.b1, hi.b0, hi.b1}
}

// The same function — ReLU -> unsigned 8-bit, right shift n >= 1 — by compile-time choice of arithmetic.  RQ == 1: the float-converter form
// above (n <= 16, bounded accumulators); RQ == 2: INTEGER ONLY (requant_u8x4_int: 2 3/4 operations per value, exact for every
// int32 and shift; what the handle's option `requant_float = 0` plans everywhere).
template <int RQ>
__device__ __forceinline__ unsigned requant_u8x4_sel(int a, int b, int c, int d, int n, float scale) {
    if constexpr (RQ == 1) return requant_u8x4(a, b, c, d, scale);
    else return requant_u8x4_int(a, b, c, d, n);
}

// int32 tensors live in an MFMA-fragment-tiled layout ("I32T"), not NHWC: blocks of 32 pixels x 32
// channels (4 KB), inside a block the order is [g = (c%32)/8][lane = ((c/4)&1)*32 + m%32][c%4] — exactly
// the accumulator layout of v_mfma_i32_32x32x32_i8 — so that a wave's residual read / int32 write of
// one accumulator group is ONE contiguous 1 KB transaction (64 lanes x 16 B) instead of 32 scattered
// 32-byte pieces.  m = linear pixel index n*P*Q + p*Q + q; rows are padded to a multiple of 32.
// Returns the int index of channel c (c % 4 == 0 for vector access) of pixel m.
__device__ __forceinline__ size_t i32t_index(int m, int c, int Cs) {
    return ((size_t)(m >> 5) * (size_t)(Cs >> 5) + (size_t)(c >> 5)) * 1024u +
           (size_t)(((((c & 31) >> 3) * 64 + ((c >> 2) & 1) * 32 + (m & 31)) << 2) + (c & 3));
}

__device__ __forceinline__ int clamp_sym31(int v) {   // clamp_(max=2^31-1, min=-(2^31-1))
    return max(v, -2147483647);
}

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM convolution on v_mfma_i32_32x32x32_i8.
//
// Block: 256 threads = 4 waves arranged WPX x WCO; tile BM pixels x BN couts, K step BK bytes.
// MFMA roles: A = weights (rows = cout), B = activations (cols = pixels), so that a lane's 4
// consecutive accumulator registers are 4 consecutive output channels of ONE pixel:
//   D reg r of lane l: cout = (r&3) + 8*(r>>2) + 4*(l>>5), pixel = l&31.
// Both operands are read from LDS as 16-byte K-contiguous chunks (lane l: row l&31, chunk
// 2*kk + (l>>5)); whatever the hardware's internal k order is, it is the same for A and B, and the
// integer sum over k is order-independent.
//
// LDS image: rows of BK bytes; 16-byte chunk c of row r is stored at chunk c ^ f(r),
// f(r) = (r / (256/BK)) % (BK/16): the 16 lanes of a ds_read_b128 service group (distinct rows
// mod 16, same logical chunk) then cover all 64 banks exactly once.
//
// Operand staging is LDS-direct (`buffer_load_dwordx4 ... lds`): each wave instruction deposits
// 64 x 16 B = 1 KB into consecutive LDS slots, the swizzle is applied on the SOURCE side (lane ->
// (row, chunk ^ f(row))), out-of-image taps and tile tails are fetched through the buffer range
// check (the DMA writes zeros).  STAGES tiles are in flight in an LDS ring; the K loop has ONE
// barrier per step and counted `s_waitcnt vmcnt(N)` so that later stages stay in flight across it.
// No data is transformed in flight, which is why unsigned activations are STORED biased (x ^ 0x80,
// i.e. x - 128 as int8) and the zero padding (biased 0 == real 128) is repaired by a per-border-class
// bias: bias[class][cout] = b + 128 * sum over the class's in-image taps of w  (host: pack_conv_weights).
// ---------------------------------------------------------------------------------------------
// XOR swizzle of an LDS image made of ROWB-byte rows (ROWB a power of two >= 64... or any multiple of 256):
// 16-byte chunk c of row r is stored at chunk c ^ f(r), so that the 16 lanes of a ds_read_b128 service group
// (consecutive rows, same logical chunk) cover all 64 banks.
template <int ROWB> struct Swz {                       // LDS bank rows are 256 bytes
    static constexpr int CPR = ROWB / 16;
    static __device__ __forceinline__ int f(int row) {
        if constexpr (ROWB >= 256) return row % 16;     // a row spans whole bank rows: spread rows over the 16 slots
        else return (row / (256 / ROWB)) % CPR;
    }
    static __device__ __forceinline__ unsigned off(int row, int chunk) { return (unsigned)(row * ROWB + ((chunk ^ f(row)) << 4)); }
};

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// f(integral_constant<int, 0>{}) ... f(integral_constant<int, N-1>{}): a loop whose index is a constant expression in the body
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// wave-uniform run-time count (the immediate has 6 bits on gfx9): a scalar jump table; counts it does not list wait for all
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
    switch (n) {
        case 1: wait_vmcnt<1>(); break; case 2: wait_vmcnt<2>(); break; case 3: wait_vmcnt<3>(); break;
         case 4: wait_vmcnt<4>(); break; case 5: wait_vmcnt<5>(); break; case 6: wait_vmcnt<6>(); break; case 7: wait_vmcnt<7>(); break;
         case 8: wait_vmcnt<8>(); break; case 9: wait_vmcnt<9>(); break; case 10: wait_vmcnt<10>(); break; case 11: wait_vmcnt<11>(); break;
         case 12: wait_vmcnt<12>(); break; case 13: wait_vmcnt<13>(); break; case 14: wait_vmcnt<14>(); break; case 15: wait_vmcnt<15>(); break;
         case 16: wait_vmcnt<16>(); break; case 17: wait_vmcnt<17>(); break; case 18: wait_vmcnt<18>(); break; case 19: wait_vmcnt<19>(); break;
         case 20: wait_vmcnt<20>(); break; case 21: wait_vmcnt<21>(); break; case 22: wait_vmcnt<22>(); break; case 23: wait_vmcnt<23>(); break;
         case 24: wait_vmcnt<24>(); break; case 25: wait_vmcnt<25>(); break; case 26: wait_vmcnt<26>(); break; case 27: wait_vmcnt<27>(); break;
         case 28: wait_vmcnt<28>(); break; case 29: wait_vmcnt<29>(); break; case 30: wait_vmcnt<30>(); break; case 31: wait_vmcnt<31>(); break;
         case 32: wait_vmcnt<32>(); break; case 33: wait_vmcnt<33>(); break; case 34: wait_vmcnt<34>(); break; case 35: wait_vmcnt<35>(); break;
         case 36: wait_vmcnt<36>(); break; case 37: wait_vmcnt<37>(); break; case 38: wait_vmcnt<38>(); break; case 39: wait_vmcnt<39>(); break;
         case 40: wait_vmcnt<40>(); break; case 41: wait_vmcnt<41>(); break; case 42: wait_vmcnt<42>(); break; case 43: wait_vmcnt<43>(); break;
         case 44: wait_vmcnt<44>(); break; case 45: wait_vmcnt<45>(); break; case 46: wait_vmcnt<46>(); break; case 47: wait_vmcnt<47>(); break;
        
        default: wait_vmcnt<0>(); break;
    }
}

}  // namespace f8
