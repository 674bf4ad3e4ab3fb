// f8_wreg.hip — 1x1 convolution for the late, weight-heavy layers: weights stream from L2 straight into registers (gfx950).
//
// The stage-opening 1x1 convs of ResNet-50's stages 2 / 3 (512 -> 256 on 28x28, 1024 -> 512 on 14x14; `layer_(res)` of
// /root/reference/models/fix_resnet.py:34 for body.0 of the first Bottleneck of a stage) ran at 0.6 / 0.9 POP/s on
// conv_igemm_kernel: 4-16 K steps per tile, one barrier and one LDS ring stage per step, most of a workgroup's life in prologue and
// epilogue.  The recipe that made the fused 7x7 launch (f8_p12.hip) run its GEMMs at 2.5-3.3 POP/s, as a stand-alone kernel:
//   * a workgroup owns 64 pixels x ALL output channels; its x8 tile (64 x K bytes, 32-64 KB) goes to LDS once (LDS-direct DMA, one
//     barrier) and is read-only afterwards;
//   * the weights are read in MFMA-fragment order ([cout tile][K32 step][lane][16 B], host: pack_frag_weights): one coalesced 1 KB
//     wave instruction per MFMA operand, straight into registers, NBUF batches rotating (no LDS staging, no barrier in the K loop);
//   * every (workgroup, wave) starts its walk over each group of 16 K steps at a different step (integer sums are exact in any
//     order): otherwise all workgroups request the same 1 KB of the weight stream at the same moment and the L2 channels holding it
//     serialise them (f8_p12.hip measured 11.7 -> 25.8 TB/s chip-wide with the rotation).
// Epilogue: bias, ReLU, requantisation to the consumers' int8 formats (up to two), 16-byte stores.  512 threads = 8 waves; wave w
// owns cout tiles w*CW .. w*CW+CW-1 for both pixel tiles.
#include "f8_device.h"

namespace f8 {

template <int CK, int COUT>
__global__ void __launch_bounds__(512) conv1x1_wreg_kernel(const ConvArgs a) {
    constexpr int NK = CK / 32;                          // K32 steps
    constexpr int CW = COUT / 32 / 8;                    // cout tiles per wave
    constexpr int NB = 8 / CW;                           // K steps per prefetch batch (8 weight registers x 4 per batch)
    constexpr int NBUF = CW == 1 ? 4 : 3;
    constexpr int NBAT = NK / NB;
    constexpr int PRE = NBUF - 1 < NBAT ? NBUF - 1 : NBAT;      // batches requested before the first multiply
    constexpr int X_BYTES = 64 * CK;
    static_assert(CW >= 1 && CW <= 2 && NK % 16 == 0 && CK >= 256, "8 waves x CW tiles; rotation groups of 16 K steps; Swz<CK>");
    constexpr int XL = X_BYTES / 16 / 512;               // DMA instructions per thread
    __shared__ __attribute__((aligned(16))) char xs[X_BYTES];
    using SX = Swz<CK>;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
    const int l31 = lane & 31, lh = lane >> 5;
    int t;
    {   // XCD-aware order (bijective): consecutive pixel tiles on one XCD
        const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
        t = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    }
    const int m0 = t * 64;
    const int rot = (wave * 2 + blockIdx.x * 3) & 15;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        const int idx = tid + i * 512;
        const int row = idx / (CK / 16), chunk = (idx % (CK / 16)) ^ SX::f(row);
        const unsigned off = (m0 + row) < a.M ? (unsigned)((m0 + row) * CK + chunk * 16) : kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(xs + i * 8192 + wave * 1024), 16, off, 0, 0, 0);
    }
    // the counted wait below relies on every weight load being YOUNGER than the DMAs: the scheduler may not move loads above this line
    // (it did: four of the 1024-channel instance's weight loads went ahead of the last DMA, and vmcnt(16) returned with it in flight)
    __builtin_amdgcn_sched_barrier(0);

    v16i acc[2][CW];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < CW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0;
    const v4i* const wp = (const v4i*)a.w + (size_t)(wave * CW) * NK * 64 + lane;       // fragment order: [tile][K32 step][lane][16 B]
    v4i wbuf[NBUF][NB][CW];
    auto load_batch = [&](v4i (&dst)[NB][CW], int s0) {          // a batch stays inside a rotation group of 16 steps
        const int gb = s0 & ~15;
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            const int st = gb + ((s0 + s + rot) & 15);
#pragma unroll
            for (int i = 0; i < CW; ++i) dst[s][i] = wp[((size_t)i * NK + st) * 64];
        }
    };
    static_for<PRE>([&](auto bc) { constexpr int B = decltype(bc)::value; load_batch(wbuf[B], B * NB); });
    wait_vmcnt<PRE * NB * CW>();                         // the x8 tile landed (the weight batches in flight are newer)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    static_for<NBAT>([&](auto bc) {
        constexpr int B = decltype(bc)::value;
        if constexpr (B + NBUF - 1 < NBAT) load_batch(wbuf[(B + NBUF - 1) % NBUF], (B + NBUF - 1) * NB);
        constexpr int GB = (B * NB) & ~15, S0 = (B * NB) & 15;
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            const int st = GB + ((S0 + s + rot) & 15);
            v4i xf[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) xf[j] = *(const v4i*)(xs + SX::off(j * 32 + l31, st * 2 + lh));
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < CW; ++i) acc[j][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wbuf[B % NBUF][s][i], xf[j], acc[j][i], 0, 0, 0);
        }
    });

    const int floor0 = a.relu0 ? 0 : INT32_MIN;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + j * 32 + l31;
        const bool ok = m < a.M;
#pragma unroll
        for (int i = 0; i < CW; ++i) {
            const int cot = (wave * CW + i) * 32;
            int y[4][4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4i bv = *(const v4i*)(a.bias + cot + 8 * g + 4 * lh);
#pragma unroll
                for (int e = 0; e < 4; ++e) y[g][e] = max((int)((unsigned)acc[j][i][4 * g + e] + (unsigned)bv[e]), floor0);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!a.q[k].ptr) continue;
                unsigned d[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    d[g] = pack4(requant1(y[g][0], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(y[g][1], a.q[k].n, a.q[k].lo, a.q[k].hi),
                                 requant1(y[g][2], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(y[g][3], a.q[k].n, a.q[k].lo, a.q[k].hi)) ^ a.q[k].bias_xor;
                auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
                if (ok) {
                    const v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
                    *(v4i*)(a.q[k].ptr + (size_t)m * COUT + cot + 16 * lh) = o;
                }
            }
        }
    }
}

// 1x1 / stride 1 / no padding, K = cin bytes, int8 outputs only: the instances
bool conv1x1_wreg_supported(int ck, int coutP) { return (ck == 512 && coutP == 256) || (ck == 1024 && coutP == 512); }

hipError_t launch_conv1x1_wreg(const ConvArgs& a, hipStream_t s) {
    const int grid = (a.M + 63) / 64;
    if (a.CK == 512 && a.coutP == 256) { hipLaunchKernelGGL((conv1x1_wreg_kernel<512, 256>), dim3(grid), dim3(512), 0, s, a); return hipGetLastError(); }
    if (a.CK == 1024 && a.coutP == 512) {
        // 64 KB of static LDS is the default limit exactly: nothing to raise
        hipLaunchKernelGGL((conv1x1_wreg_kernel<1024, 512>), dim3(grid), dim3(512), 0, s, a);
        return hipGetLastError();
    }
    return hipErrorInvalidValue;
}

}  // namespace f8
