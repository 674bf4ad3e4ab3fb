// f8_p12.hip — the first two convolutions of a 7x7 bottleneck identity block in one launch (gfx950).
//
//   x8 (int8 NHWC, C ch, 7x7) --1x1 C->MID, ReLU--> mid1 --3x3, pad 1, ReLU--> mid2 (int8 NHWC, the 1x1 that follows reads it)
//
// body.0 and body.2 of IntBlock.forward (/root/reference/models/fix_resnet.py:26-39) for the stage-3 identity blocks of
// ResNet-50 (C = 2048, MID = 512), every int_op_only_fix_quant (fix_quant_ops.py:90-114) in place.  Unfused these are two
// launches of 23 + 27 us per 128 images at 0.6 / 1.1 POP/s: 6272 output pixels are 98 tiles of 64 — neither launch fills the chip
// and both pay their fixed costs.  Fusing the WHOLE block (as in stages 0-2) does not pay here: a workgroup would stream 4.4 MB of
// weights per image and there are only 128 images.  This kernel takes the middle road:
//   * one image = a PAIR of workgroups (256 workgroups per 128 images: every CU has one);
//   * P1 (1x1, K = C) is computed by BOTH workgroups of the pair — it is 23 % of the block's MACs and the 3x3 needs all of its
//     output channels, so splitting it would need a cross-workgroup exchange — result requantised into an LDS patch
//     (9 x 9 entries of MID bytes, zero border = biased zero);
//   * P2 (3x3) is SPLIT by output channel: workgroup h computes channels [MID/2 * h, MID/2 * (h + 1)) and streams only that
//     half of W2 (1.18 MB instead of 2.36 MB); result requantised and written straight to HBM;
//   * the residual-carrying 1x1 that follows is HBM-bound and stays the launch it was (3.7 TB/s).
// The K loops have NO barriers: a workgroup of 49 pixels does 4 (P1) / 2 (P2) MFMAs per wave and 32 bytes of K, so a barrier per
// LDS ring stage was the whole step time (first version: 66 us, slower than the two launches).  Instead
//   * the WEIGHTS never touch LDS: they are stored a second time in MFMA-fragment order ([cout tile][K32 step][lane][16 B], host:
//     pack_frag_weights), so one wave instruction fetches the A operand of one MFMA as a contiguous 1 KB straight into registers;
//     each wave prefetches a batch of 8 K steps while it multiplies the previous batch;
//   * the ACTIVATIONS are read-only in LDS: x8 arrives in four K quarters of 32 KB (double-buffered, LDS-direct DMA: 4 barriers for
//     the whole of P1), the patch is complete before P2 starts.
// 512 threads = 8 waves.  P1: wave w owns mid channels 64w .. 64w+63 for both pixel tiles (2 x 2 register block);
// P2: wave w owns output channels 32 * (8h + w) .. +31 for both pixel tiles.
#include "f8_device.h"
#include <cstdio>
#include <cstdlib>

namespace f8 {

template <int C, int MID>
struct P12Cfg {
    static constexpr int W = 7, PX = 49, PW = 9;
    static constexpr int PATCH_BYTES = (PW * PW * MID + 255) / 256 * 256;
    static constexpr int KQ = C / 4;                     // K bytes per x8 quarter
    static constexpr int XQ_BYTES = 64 * KQ;
    static constexpr int BIAS_INTS = 2 * MID;
    static constexpr int LDS_BYTES = PATCH_BYTES + 2 * XQ_BYTES + BIAS_INTS * 4;
};

template <int C, int MID>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
fused_p12_kernel(const FusedArgs a) {
    using Cfg = P12Cfg<C, MID>;
    constexpr int W = Cfg::W, PX = Cfg::PX, PW = Cfg::PW, KQ = Cfg::KQ, XQ_BYTES = Cfg::XQ_BYTES;
    constexpr int CM = MID / 32;                         // mid-channel tiles: P1 computes all of them, P2 half of them per workgroup
    constexpr int NK1 = C / 32, NK2 = 9 * (MID / 32);    // K32 steps
    constexpr int NB = 8;                                // K32 steps per prefetch batch in P2 (one weight tile per step)
    constexpr int NB1 = 4;                               // ... in P1 (two weight tiles per step: same 16 registers per batch)
    constexpr int SQ = KQ / 32;                          // K32 steps per x8 quarter
    static_assert(CM == 16, "8 waves x 2 mid tiles in P1; 8 waves x 1 of the workgroup's 8 output tiles in P2");
    static_assert(SQ % (2 * NB1) == 0 && NK2 % NB == 0 && KQ >= 256, "whole batch pairs per quarter; Swz<KQ> rows span whole bank rows");
    static_assert(Cfg::LDS_BYTES <= 160 * 1024, "LDS");
    constexpr int XL = XQ_BYTES / 16 / 512;              // DMA instructions per thread and quarter

    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const patch = lds;
    char* const xq = lds + Cfg::PATCH_BYTES;             // two quarters of x8: [64 rows][KQ bytes], 16-byte chunks XOR-swizzled by the row
    int* const bias_lds = (int*)(xq + 2 * XQ_BYTES);     // b0 | b2
    using SX = Swz<KQ>;
    using SM = Swz<MID>;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
    const int l31 = lane & 31, lh = lane >> 5;
    // Workgroup -> (image, half of the 3x3's output channels), XCD-aware: consecutive block ids go round the 8 XCDs, and every
    // XCD has its own 4 MB L2.  XCDs 0-3 take h = 0, XCDs 4-7 take h = 1, so one XCD streams W0 (1 MB) and ONE half of W2 (1.18 MB):
    // 2.2 MB, L2-resident.  With h = block id & 1 every XCD cycled through all 3.4 MB plus its images' x8 and missed L2 on every
    // pass (58.9 us per block; the weights then come from the memory-side cache at ~10 TB/s chip-wide).
    const int xcd = blockIdx.x & 7, h = xcd >> 2;
    const int n = (blockIdx.x >> 3) * 4 + (xcd & 3);
    if (n >= a.N) return;                                // the grid is rounded up to whole groups of 8
    const int gp = n * PX;                               // first global pixel of the image
    // K-order rotation.  Integer accumulation is exact in any order, so every (workgroup, wave) walks the 16 K32 steps of an x8
    // quarter / of a 3x3 tap starting at a different one.  Without it all 256 workgroups x 8 waves request the SAME 1 KB pieces of
    // the weight stream at the same moment (and the pieces of different cout tiles are a multiple of 64 KB apart): the L2 channels
    // that hold them serialise the requests while the others idle — P2 ran at 19 B/clk/CU (11.7 TB/s chip-wide) with 24 KB in
    // flight per wave.
    const int rot = (wave * 2 + (blockIdx.x >> 3) * 3 + (blockIdx.x & 7)) & 15;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x8, 0, a.x_bytes, 0x00020000);
#ifdef F8_TRACE
    unsigned long long tt[8]; tt[0] = __builtin_readcyclecounter();
#define F8_PT(i) tt[i] = __builtin_readcyclecounter()
#else
#define F8_PT(i)
#endif

    // ---- x8 quarter q -> LDS slot (q & 1)
    unsigned xb[XL];
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        const int idx = tid + i * 512;
        const int row = idx / (KQ / 16), chunk = (idx % (KQ / 16)) ^ SX::f(row);
        xb[i] = row < PX ? (unsigned)((gp + row) * C + chunk * 16) : kOOB;
    }
    auto issue_x = [&](int q) {
        char* base = xq + (q & 1) * XQ_BYTES;
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const unsigned off = xb[i] + (unsigned)(q * KQ);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(base + i * 8192 + wave * 1024), 16, off, 0, 0, 0);
        }
    };
    // biases: requested first (older than the DMAs), parked in registers, stored to LDS after the first barrier of P1 — a store right
    // here made the compiler wait vmcnt(0) behind the DMAs: both x8 quarters had to land before the first weight load was issued
    static_assert(Cfg::BIAS_INTS == 2 * 512, "two bias words per thread");
    const int bias_pre[2] = {a.b0[tid], a.b2[tid]};
    __builtin_amdgcn_sched_barrier(0);
    issue_x(0);
    issue_x(1);
    __builtin_amdgcn_sched_barrier(0);                   // the counted waits of P1 need the weight loads younger than these DMAs
    {   // patch <- biased zero (the border keeps it; P1 writes the 49 interior entries)
        const v4i zv = {(int)a.xor1, (int)a.xor1, (int)a.xor1, (int)a.xor1};
        for (int o = tid * 16; o < PW * PW * MID; o += 512 * 16) *(v4i*)(patch + o) = zv;
    }

    // ================= P1: mid1 = requant(relu(W0 . x8 + b0)) for all MID channels -> patch
    {
        v16i acc[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0;
        // this wave's two weight streams: co tiles 2w, 2w+1, fragment order: [tile][K32 step][lane][16 B]
        const v4i* const wp0 = (const v4i*)a.w0 + (size_t)(wave * 2) * NK1 * 64 + lane;
        const v4i* const wp1 = wp0 + (size_t)NK1 * 64;
        // NBUF batches of NB1 K steps rotate through registers: while one is multiplied, NBUF - 1 are in flight.  All workgroups walk the
        // same weight stream in lock step, so nobody finds a line that somebody else fetched earlier: every batch sees memory latency
        // (~3-4 k cycles measured with one batch ahead: 141 k cycles per workgroup), not L2-hit latency.
        constexpr int NBUF = 3, NBAT1 = NK1 / NB1, BPQ = SQ / NB1;          // batches in P1, batches per x8 quarter
        v4i wbuf[NBUF][NB1][2];
        static_assert(SQ == 16, "rotation inside a quarter of 16 K32 steps");
        auto load_batch = [&](v4i (&dst)[NB1][2], int s0) {              // s0: first K32 step of the batch (a batch stays inside a quarter)
            const int qb = s0 & ~15;
#pragma unroll
            for (int s = 0; s < NB1; ++s) {
                const int st = qb + ((s0 + s + rot) & 15);
                dst[s][0] = wp0[(size_t)st * 64]; dst[s][1] = wp1[(size_t)st * 64];
            }
        };
        auto mul_batch = [&](const v4i (&wv)[NB1][2], const char* xbase, int sq0) {      // sq0: first K32 step inside the quarter
#pragma unroll
            for (int s = 0; s < NB1; ++s) {
                v4i xf[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) xf[j] = *(const v4i*)(xbase + SX::off(j * 32 + l31, ((sq0 + s + rot) & 15) * 2 + lh));
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wv[s][i], xf[j], acc[j][i], 0, 0, 0);
            }
        };
        static_for<NBUF - 1>([&](auto bc) { constexpr int B = decltype(bc)::value; load_batch(wbuf[B], B * NB1); });
        static_for<NBAT1>([&](auto bc) {
            constexpr int B = decltype(bc)::value;
            if constexpr (B % BPQ == 0) {
                constexpr int Q = B / BPQ;
                // quarter Q landed (this wave's DMA: counted wait — the NBUF - 1 weight batches in flight are all newer than that DMA,
                // VMEM retires in order; everybody's: barrier).  Every wave is past quarter Q - 1: its slot takes quarter Q + 1.
                wait_vmcnt<(NBUF - 1) * 2 * NB1>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if constexpr (Q == 0) { bias_lds[tid] = bias_pre[0]; bias_lds[MID + tid] = bias_pre[1]; }      // read after later barriers
                if constexpr (Q >= 1 && Q + 1 < 4) { issue_x(Q + 1); __builtin_amdgcn_sched_barrier(0); }
            }
            if constexpr (B + NBUF - 1 < NBAT1) load_batch(wbuf[(B + NBUF - 1) % NBUF], (B + NBUF - 1) * NB1);
            mul_batch(wbuf[B % NBUF], xq + ((B / BPQ) & 1) * XQ_BYTES, (B % BPQ) * NB1);
        });


        F8_PT(1);
        const int floor0 = a.relu_a ? 0 : INT32_MIN;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int px = j * 32 + l31;
            const bool ok = px < PX;
            const int pr = px / W, pc = px - pr * W;
            const int ent = (pr + 1) * PW + pc + 1;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ct = wave * 2 + i;
                unsigned d[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const v4i bv = *(const v4i*)(bias_lds + ct * 32 + 8 * g + 4 * lh);
                    int y[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = requant1(max((int)((unsigned)acc[j][i][4 * g + e] + (unsigned)bv[e]), floor0), a.n1, a.lo1, a.hi1);
                    d[g] = pack4(y[0], y[1], y[2], y[3]) ^ a.xor1;
                }
                auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
                if (ok) {
                    const v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
                    *(v4i*)(patch + SM::off(ent, ct * 2 + lh)) = o;
                }
            }
        }
    }

    F8_PT(2);
    // ================= P2: this workgroup's half of mid2 = requant(relu(conv3x3(mid1) + b2)) -> HBM
    {
        const int ct = h * (CM / 2) + wave;              // output channel tile of this wave
        v16i acc[2];
        unsigned tapoff[2][9], tapsw[2][9];              // byte offset / swizzle term of the 9 tap entries of this lane's pixel
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int px = j * 32 + l31, oc = px < PX ? px : PX - 1;
            const int orow = oc / W, ocol = oc - orow * W;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ent = (orow + t / 3) * PW + ocol + t % 3;
                tapoff[j][t] = (unsigned)(ent * MID);
                tapsw[j][t] = (unsigned)SM::f(ent);
            }
        }
        const v4i* const wp = (const v4i*)a.w2 + (size_t)ct * NK2 * 64 + lane;      // fragment order: [tile][K32 step][lane][16 B]
        constexpr int NBUF = 4, NBAT2 = NK2 / NB;
        constexpr int SPT = MID / 32;                    // K32 steps per tap (16): a batch of 8 never straddles a tap
        static_assert(SPT % NB == 0, "a batch stays inside one tap");
        v4i wbuf[NBUF][NB];
        static_assert(SPT == 16, "rotation inside a tap of 16 K32 steps");
        auto load_batch = [&](v4i (&dst)[NB], int s0) {                  // a batch stays inside a tap
            const int tb = s0 & ~15;
#pragma unroll
            for (int s = 0; s < NB; ++s) dst[s] = wp[(size_t)(tb + ((s0 + s + rot) & 15)) * 64];
        };
        static_for<NBUF - 1>([&](auto bc) { constexpr int B = decltype(bc)::value; load_batch(wbuf[B], B * NB); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // the patch is complete
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4i bv = *(const v4i*)(bias_lds + MID + ct * 32 + 8 * g + 4 * lh);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][4 * g + e] = bv[e];
            }
        static_for<NBAT2>([&](auto bc) {                 // static batch index: the tap (and with it tapoff / tapsw) is a constant
            constexpr int B = decltype(bc)::value;
            constexpr int T = (B * NB) / SPT, SB = (B * NB) % SPT;
            if constexpr (B + NBUF - 1 < NBAT2) load_batch(wbuf[(B + NBUF - 1) % NBUF], (B + NBUF - 1) * NB);
#pragma unroll
            for (int s = 0; s < NB; ++s)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const v4i xf = *(const v4i*)(patch + tapoff[j][T] + ((((unsigned)(((SB + s + rot) & 15) * 2 + lh)) ^ tapsw[j][T]) << 4));
                    acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wbuf[B % NBUF][s], xf, acc[j], 0, 0, 0);
                }
        });
        F8_PT(3);
        const int floor0 = a.relu_b ? 0 : INT32_MIN;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int px = j * 32 + l31;
            const bool ok = px < PX;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!a.q[k].ptr) continue;
                unsigned d[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    d[g] = pack4(requant1(max(acc[j][4 * g + 0], floor0), a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(max(acc[j][4 * g + 1], floor0), a.q[k].n, a.q[k].lo, a.q[k].hi),
                                 requant1(max(acc[j][4 * g + 2], floor0), a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(max(acc[j][4 * g + 3], floor0), a.q[k].n, a.q[k].lo, a.q[k].hi)) ^ a.q[k].bias_xor;
                auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
                if (ok) {
                    const v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
                    *(v4i*)(a.q[k].ptr + (size_t)(gp + px) * MID + ct * 32 + 16 * lh) = o;
                }
            }
        }
    }
#ifdef F8_TRACE
    if (a.trace && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tt[4] = __builtin_readcyclecounter();
        unsigned long long* tp = (unsigned long long*)a.trace + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 5; ++i) tp[i] = tt[i];
    }
#endif
}

bool fused_p12_supported(int C, int MID, int H, int W) { return C == 2048 && MID == 512 && H == 7 && W == 7; }

hipError_t launch_fused_p12(const FusedArgs& a, hipStream_t s) {
    if (!fused_p12_supported(a.C, a.MID, a.H, a.W)) return hipErrorInvalidValue;
    using Cfg = P12Cfg<2048, 512>;
    static unsigned long long attr_done = 0; int attr_dev = -1;
    if (!dyn_lds_opted_in(&attr_done, &attr_dev)) {
        hipError_t e = hipFuncSetAttribute((const void*)fused_p12_kernel<2048, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        if (attr_dev >= 0) attr_done |= 1ull << attr_dev;
    }
    const int grid = (a.N + 3) / 4 * 8;
#ifdef F8_TRACE
    static unsigned long long* tbuf = nullptr; static int count = 0;
    static const int want = [] { const char* e = getenv("F8_TRACE_P12"); return e ? atoi(e) : -1; }();
    FusedArgs b = a;
    const bool tracing = (count++ == want);
    if (tracing) { if (!tbuf) (void)hipMalloc((void**)&tbuf, (size_t)1 << 20); (void)hipMemsetAsync(tbuf, 0, (size_t)grid * 64, s); b.trace = tbuf; }
    hipLaunchKernelGGL((fused_p12_kernel<2048, 512>), dim3(grid), dim3(512), Cfg::LDS_BYTES, s, b);
    if (tracing) {
        (void)hipStreamSynchronize(s);
        unsigned long long* hb = new unsigned long long[(size_t)grid * 8];
        (void)hipMemcpy(hb, tbuf, (size_t)grid * 64, hipMemcpyDeviceToHost);
        double ph[4] = {0, 0, 0, 0}; int n = 0;
        for (int i = 0; i < grid; ++i) { unsigned long long* p = hb + (size_t)i * 8; if (!p[4]) continue; ++n; for (int k = 0; k < 4; ++k) ph[k] += (double)(p[k + 1] - p[k]); }
        fprintf(stderr, "[trace p12] grid %d: avg cycles per WG: P1 loop %.0f | P1 epi %.0f | P2 loop %.0f | P2 epi+drain %.0f\n", grid, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n);
        delete[] hb;
    }
    return hipGetLastError();
#else
    hipLaunchKernelGGL((fused_p12_kernel<2048, 512>), dim3(grid), dim3(512), Cfg::LDS_BYTES, s, a);
    return hipGetLastError();
#endif
}

}  // namespace f8
