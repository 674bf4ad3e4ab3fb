// f8_fused.hip — one launch for a whole ResNet bottleneck identity block (gfx950).
//
//   x8 (int8 NHWC, C ch) --1x1 C->MID, ReLU--> mid1 --3x3 MID->MID pad 1, ReLU--> mid2 --1x1 MID->C-->
//   + residual (int32, I32T) -> clamp -> ReLU -> y32 (I32T) and/or requantised int8 copies
//
// i.e. IntBlock.forward of /root/reference/models/fix_resnet.py:26-54,77 for a Bottleneck identity block
// (:233-319), with every int_op_only_fix_quant (fix_quant_ops.py:90-114) in place: after the first
// and second conv the values are requantised to the next layer's 8-bit format and stay in LDS.
//
// Why fuse: unfused, the three convs are three launches whose small ones are latency / issue bound
// while the residual-carrying one is HBM bound; fused, the MFMA work of all three hides under the
// residual stream of the block, the two int8 intermediates never touch HBM and two launches per block
// disappear.  The block stays HBM-bound: its floor is (int8 in + int32 residual in + int32/int8 out)
// / HBM bandwidth.
//
// Work unit: R output rows x full width W of ONE image (so only a vertical halo exists).
//   P1  mid1 for rows p0-1 .. p0+R (halo rows recomputed), GEMM [(R+2)*W px] x [MID] x K=C,
//       operands streamed through a 2-stage LDS ring by LDS-direct DMA; result requantised and written
//       into an LDS "patch" [(R+2)][W+2][MID] whose border (image edge / halo columns) holds the
//       biased zero, so the 3x3 needs no border classes.
//   P2  3x3 from the patch (tap = constant LDS offset), weights streamed; result -> LDS mid2.
//   P3  1x1 MID->C in chunks of 64 output channels: weights streamed, residual chunk prefetched one
//       chunk ahead (the first one before P2's epilogue), fused epilogue (align, add, clamp, ReLU,
//       int32 + int8 stores).
// 512 threads = 8 waves in a 4 (pixel tiles) x 2 (channel tiles) grid; MFMA v_mfma_i32_32x32x32_i8 with
// A = weights, B = activations as in conv_igemm_kernel; all LDS rows are XOR-swizzled per 16-byte
// chunk (see f8_kernels.hip).  LDS is dynamic (up to ~120 KB for MID = 256).
#include "f8_device.h"
#include <type_traits>
#include <cstdlib>
#include <cstdio>

namespace f8 {

template <int C, int MID, int W, int R>
struct FusedCfg {
    static constexpr int PW = W + 2, PR = R + 2;
    static constexpr int PATCH_PX = PR * PW;
    static constexpr int P1_PX = PR * W;
    static constexpr int NP1 = (P1_PX + 31) / 32;
    static constexpr int OUT_PX = R * W;
    static constexpr int NPO = (OUT_PX + 31) / 32;
    static constexpr int CM = MID / 32;
    static constexpr int X1_ROWS = NP1 * 32;
    static constexpr int X1_BYTES = X1_ROWS * 64, W_BYTES = MID * 64;
    static constexpr int RING = X1_BYTES + W_BYTES;
    static constexpr int PATCH_BYTES = (PATCH_PX * MID + 255) / 256 * 256, MID2_BYTES = 4 * 32 * MID;
    static constexpr int LDS_BYTES = PATCH_BYTES + MID2_BYTES + 2 * RING;
};

// DS = the stage's FIRST bottleneck when it keeps the resolution (ResNet-50 stage 0: 64 -> 64 -> 64 -> 256 with a 1x1
// shortcut conv 64 -> 256 instead of an identity): P3 has no residual stream to read; the join's second operand is a
// second product Wsc . x over the same pixels, whose x fragments are still in LDS from P1 (C == 64: one K step).
template <int C, int MID, int W, int R, int COUT = C, bool DS = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(MID <= 128 ? 4 : 2)))
fused_bottleneck_kernel(const FusedArgs a) {   // MID <= 128: <= 128 VGPRs so that two workgroups share a CU (LDS allows it)
    using Cfg = FusedCfg<C, MID, W, R>;
    constexpr int PW = Cfg::PW;
    constexpr int P1_PX = Cfg::P1_PX, NP1 = Cfg::NP1, OUT_PX = Cfg::OUT_PX, NPO = Cfg::NPO, CM = Cfg::CM;
    static_assert(NPO <= 4 && NP1 <= 8, "4 pixel-tile groups");
    static_assert(CM % 2 == 0, "2 channel-tile groups");
    constexpr int NP1W = (NP1 + 3) / 4;                  // px tiles per wave in P1
    constexpr int CMW = CM / 2;                          // co tiles per wave in P1 / P2
    constexpr int NK1 = C / 64;                          // P1 K steps
    constexpr int NK2 = 9 * (MID / 64);                  // P2 K steps (tap-major, 64-byte channel chunks)
    constexpr int NC3 = COUT / 64;                       // P3 chunks of 64 output channels
    static_assert(DS || COUT == C, "identity blocks keep the channel count");
    static_assert(!DS || (C == 64 && MID == 64), "DS instance: one P1 K step, 4 KB weight tiles");
    constexpr int KK3 = MID / 32;
    constexpr int X1_BYTES = Cfg::X1_BYTES, RING = Cfg::RING;
    constexpr int PATCH_BYTES = Cfg::PATCH_BYTES, MID2_BYTES = Cfg::MID2_BYTES;
    constexpr int XS1 = Cfg::X1_ROWS * 4, XL1 = (XS1 + 511) / 512;     // P1 X tile slots
    constexpr int WS = MID * 4, WL = (WS + 511) / 512;                  // W0 / W2 tile slots (rows of 64 B); the W4 tile has the same count
    static_assert(NC3 % 2 == 0, "chunk loop is unrolled by two (residual ping-pong)");

    // ALLW (MID == 64, the 56x56 kernels): operands are small enough to sit in LDS WHOLE — all NK1 stages of P1 at once
    // (laid over the regions that only come alive later) and all of W2 (36 KB) — so P1 and P2 are one wait + one barrier
    // followed by MFMAs, instead of 4 + 9 latency-bound ring steps (per-workgroup traces: 13.0k + 10.0k cycles of a 44k life).
    constexpr bool ALLW = (MID == 64);
    constexpr int W4STAGE = (MID * 64) * (DS ? 2 : 1);                   // one P3 stage: W4 tile (+ shortcut tile)
    constexpr int W2ALL = NK2 * (MID * 64);

    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const patch = lds;
    char* const mid2 = lds + PATCH_BYTES;
    char* const ring = lds + PATCH_BYTES + MID2_BYTES;                   // !ALLW: two (X1 + W) stages; ALLW: W2 whole, then the W4 ring
    char* const w4ring = ALLW ? ring + W2ALL : ring;
    constexpr int W4STRIDE = ALLW ? W4STAGE : RING;
    constexpr int NS1 = (!ALLW && MID == 256) ? 4 : 2;                  // P1 ring depth (see issue_p1)
    constexpr int LDS_END = ALLW ? ((NK1 * RING > PATCH_BYTES + MID2_BYTES + W2ALL + 2 * W4STAGE) ? NK1 * RING : PATCH_BYTES + MID2_BYTES + W2ALL + 2 * W4STAGE)
                                 : (NS1 > 2 ? PATCH_BYTES + NS1 * RING : Cfg::LDS_BYTES);

    using S64 = Swz<64>;
    using SM = Swz<MID>;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
    const int wa = wave >> 1, wb = wave & 1;             // pixel-tile group (0..3), channel-tile group (0..1)
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- tile: XCD-aware order (consecutive tiles = vertically adjacent row groups share halo rows in L2)
    int t;
    {
        const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
        t = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    }
    const int n = t / a.tiles_per_img, ti = t - n * a.tiles_per_img;
    const int p0 = ti * R;
    const int rows_out = (a.H - p0) < R ? (a.H - p0) : R;
    const int gp1 = (n * a.H + p0 - 1) * W;              // global pixel index of P1 pixel 0 (may be < 0)
    const int m_tile = (n * a.H + p0) * W;               // global pixel index of output pixel 0

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x8, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w0, 0, a.w0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2, 0, a.w2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw4 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w4, 0, a.w4_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwsc = __builtin_amdgcn_make_buffer_rsrc((void*)(DS ? a.wsc : a.w4), 0, DS ? a.wsc_bytes : 0u, 0x00020000);

#ifdef F8_TRACE
    unsigned long long tt[8]; tt[0] = __builtin_readcyclecounter();
#define F8_TT(i) tt[i] = __builtin_readcyclecounter()
#else
#define F8_TT(i)
#endif
    // ---- patch <- biased zero everywhere (border and out-of-image rows keep it); ALLW: after P1 (its stages lie over the patch)
    auto zero_patch = [&]() {
        const unsigned z = a.xor1;
        const v4i zv = {(int)z, (int)z, (int)z, (int)z};
        for (int o = tid * 16; o < PATCH_BYTES; o += 512 * 16) *(v4i*)(patch + o) = zv;
    };
    if (!ALLW) zero_patch();

    // ---- gather descriptors
    unsigned xb1[XL1];
#pragma unroll
    for (int i = 0; i < XL1; ++i) {
        const int idx = tid + i * 512;
        const int row = idx >> 2, chunk = (idx & 3) ^ S64::f(row);
        const int pr = row / W;                          // patch row of this P1 pixel
        const int hrow = p0 - 1 + pr;
        const bool ok = idx < XS1 && row < P1_PX && hrow >= 0 && hrow < a.H;
        xb1[i] = ok ? (unsigned)((gp1 + row) * C + chunk * 16) : kOOB;
    }
    unsigned w0b[WL], w2b[WL], w4b[WL];
#pragma unroll
    for (int j = 0; j < WL; ++j) {
        const int idx = tid + j * 512;
        const int row = idx >> 2, chunk = (idx & 3) ^ S64::f(row);             // rows of 64 B (W0 / W2 tiles)
        w0b[j] = (unsigned)(row * C + chunk * 16);
        w2b[j] = (unsigned)(row * (9 * MID) + chunk * 16);
        const int r4 = idx / (MID / 16), c4 = (idx % (MID / 16)) ^ SM::f(r4);  // rows of MID B (W4 tile)
        w4b[j] = (unsigned)(r4 * MID + c4 * 16);
    }

    // fragment ping-pong (MID == 256): MFMAs have no memory semantics, so the scheduler is free to hoist them above the barrier,
    // next to the reads that feed them — which turns the pipeline back into read-then-multiply.  Passing the fragments through
    // an empty asm AFTER the next stage's reads are issued pins the multiplies behind those reads.
    auto pin = [](auto& wf, auto& xf) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            asm volatile("" : "+v"(xf[kk]) :: "memory");
#pragma unroll
            for (int i = 0; i < (int)(sizeof(wf[0]) / sizeof(wf[0][0])); ++i) asm volatile("" : "+v"(wf[kk][i]) :: "memory");
        }
    };
    // !ALLW, MID == 256 (one workgroup per CU, LDS to spare): the P1 ring is four stages deep and starts at `mid2`, which is
    // not live before P2's epilogue.  All workgroups stream their x8 tiles at the same time, so the stream runs at the chip's
    // HBM rate (~10 B/clk/CU) with a loaded latency of several thousand cycles: the ring has to cover that (ablation: P1
    // 26.5k cycles, 15.0k with the DMA removed, HBM floor 13k; a 2-stage ring took 30.6k)
    static_assert(ALLW || NS1 == 2 || (PATCH_BYTES + NS1 * RING + 4 * COUT <= 160 * 1024 && (XS1 % 512) == 0 && (WS % 512) == 0), "deep P1 ring: fits, uniform DMA count per thread");
    char* const p1ring = NS1 == 2 ? ring : mid2;
    auto issue_p1 = [&](int ks, int slot) {
#ifdef F8_ABL_NODMA
        if (MID == 256) return;
#endif
        char* base = ALLW ? lds + slot * RING : p1ring + slot * RING;
#pragma unroll
        for (int i = 0; i < XL1; ++i) {
            const unsigned off = xb1[i] + (unsigned)(ks * 64);   // kOOB + ks*64 stays beyond any buffer (< 2 GiB): no select
            if ((i * 512 + wave * 64) < XS1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(base + i * 8192 + wave * 1024), 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < WL; ++j) {
            const unsigned woff = w0b[j] + (unsigned)(ks * 64);
            if ((j * 512 + wave * 64) < WS)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw0, (__attribute__((address_space(3))) void*)(base + X1_BYTES + j * 8192 + wave * 1024), 16, woff, 0, 0, 0);
        }
    };
    // !ALLW: W2 streams through NS2 stages of MID x 64 bytes laid over the two P1 slots (P1's last stage sits in slot 1, so only
    // the stages inside slot 0 may be issued before P2's first barrier)
    constexpr int W2B = MID * 64;
    constexpr int RS2 = 6;                               // deep variant (MID == 256): W2 ring stages, from `mid2` (dead until P2's epilogue)
    static_assert(NS1 == 2 || RS2 * W2B <= NS1 * RING, "deep W2 ring lies inside the dead P1 ring");
    constexpr int NS2 = (2 * RING / W2B) < 5 ? (2 * RING / W2B) : 5;
    constexpr int PRE2 = (RING / W2B) < (NS2 - 1) ? (RING / W2B) : (NS2 - 1);
    static_assert(ALLW || ((NK1 & 1) == 0 && PRE2 >= 1 && (NK2 - 1) % NS2 != 0), "W2 ring placement");
    auto issue_w2 = [&](int j2, int slot) {              // step j2: bytes [j2*64, j2*64+64) of every W2 row (tap-major K)
#ifdef F8_ABL_NODMA
        if (MID == 256) return;
#endif
        char* base = (NS1 > 2 ? mid2 : ring) + slot * W2B;   // deep variant: RS2 stages over mid2 + the old ring (see P2)
#pragma unroll
        for (int j = 0; j < WL; ++j) {
            const unsigned woff = w2b[j] + (unsigned)(j2 * 64);
            if ((j * 512 + wave * 64) < WS)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw2, (__attribute__((address_space(3))) void*)(base + j * 8192 + wave * 1024), 16, woff, 0, 0, 0);
        }
    };
    auto issue_w4 = [&](int c, int slot) {               // 64 output channels x MID bytes
        char* base = w4ring + slot * W4STRIDE;
        if (DS && wave >= 4) {                           // DS: waves 4..7 fetch the shortcut tile (64 couts x C bytes) behind the W4 tile
            const int sl = tid - 256;
            const int row = sl >> 2, chunk = (sl & 3) ^ S64::f(row);
            const unsigned woff = (unsigned)((c * 64 + row) * C + chunk * 16);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rwsc, (__attribute__((address_space(3))) void*)(base + MID * 64 + (wave - 4) * 1024), 16, woff, 0, 0, 0);
            return;
        }
#pragma unroll
        for (int j = 0; j < WL; ++j) {
            const unsigned woff = w4b[j] + (unsigned)(c * 64 * MID);
            if ((j * 512 + wave * 64) < WS)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw4, (__attribute__((address_space(3))) void*)(base + j * 8192 + wave * 1024), 16, woff, 0, 0, 0);
        }
    };

    // MID == 256 (one workgroup per CU): P3 runs D3 = 4 chunks deep — W4 tiles and residual chunks are requested three chunks
    // ahead, so a chunk waits for the stores issued FOUR chunks ago instead of two (VMEM retires in order and loads share the
    // counter with stores: with the 2-deep pipeline every chunk lasted one store round trip, 5.7k cycles in the trace).
    constexpr int D3 = (!ALLW && MID == 256 && !DS) ? 4 : 2;
    v4i xs[2] = {};                                      // DS: x fragments of the shortcut product (read at the end of P1)
    int* const bias_lds = (int*)(lds + LDS_END);  // DS: b4[COUT] then bsc[COUT] (a global bias load per chunk would expose its latency:
    if (D3 == 4) {                                       // deep P3: b4[COUT], so that no bias load sits in the VMEM queue
        for (int i = tid; i < COUT; i += 512) bias_lds[i] = a.b4[i];
    }
    if (DS) {                                            //     there is no residual stream whose prefetch could hide it)
        static_assert(!DS || COUT * 2 <= 512, "one bias word per thread");
        if (tid < COUT) bias_lds[tid] = a.b4[tid];
        else if (tid < 2 * COUT) bias_lds[tid] = a.bsc[tid - COUT];
    }

    // =========================================================================================
    // P1: mid1 = requant(relu(W0 . x8 + b0)) on (R+2) x W pixels  ->  patch
    //     wave (wa, wb): px tiles {wa, wa+4}, co tiles {wb*CMW .. wb*CMW+CMW-1}
    // =========================================================================================
    {
        v16i acc[NP1W][CMW];                             // start at the bias (below): no add per value in the epilogue
        unsigned cof[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) cof[kk] = (unsigned)(((kk * 2 + lh) ^ S64::f(l31)) << 4);
        // biases are fetched BEFORE any DMA / residual load of the phase is issued: VMEM returns in order, so a
        // bias load issued behind them would make the epilogue wait for all of them (measured: 7-9k cycles)
        v4i bq0[CMW][4];
#pragma unroll
        for (int i = 0; i < CMW; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) bq0[i][g] = *(const v4i*)(a.b0 + (wb * CMW + i) * 32 + 8 * g + 4 * lh);
#pragma unroll
        for (int j = 0; j < NP1W; ++j)
#pragma unroll
            for (int i = 0; i < CMW; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = bq0[i][r >> 2][r & 3];

        auto p1_mma = [&](const char* base) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                v4i wf[CMW], xf[NP1W];
#pragma unroll
                for (int i = 0; i < CMW; ++i) wf[i] = *(const v4i*)(base + X1_BYTES + ((wb * CMW + i) * 32 + l31) * 64 + cof[kk]);
#pragma unroll
                for (int j = 0; j < NP1W; ++j) {
                    const int pt = wa + 4 * j;
                    if (pt < NP1) xf[j] = *(const v4i*)(base + (pt * 32 + l31) * 64 + cof[kk]);
                }
#pragma unroll
                for (int j = 0; j < NP1W; ++j) {
                    const int pt = wa + 4 * j;
                    if (pt < NP1)
#pragma unroll
                        for (int i = 0; i < CMW; ++i) acc[j][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[i], xf[j], acc[j][i], 0, 0, 0);
                }
            }
        };
        if constexpr (ALLW) {
            // every P1 stage at once (over the patch / mid2 / W2 regions, which are not live yet): one wait, one barrier
#pragma unroll
            for (int ks = 0; ks < NK1; ++ks) issue_p1(ks, ks);
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int ks = 0; ks < NK1; ++ks) p1_mma(lds + ks * RING);
        } else if constexpr (NS1 == 2) {
            issue_p1(0, 0);
            static_for<NK1>([&](auto kc) {               // unrolled: slots and K offsets are immediates
                constexpr int KS = decltype(kc)::value;
                wait_vmcnt<0>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if constexpr (KS + 1 < NK1) issue_p1(KS + 1, (KS + 1) & 1);
                p1_mma(ring + (KS & 1) * RING);
            });
        } else {
            constexpr int L1 = XL1 + WL;                 // DMA instructions per thread and stage (uniform, see the assert)
            static_assert(NK1 >= NS1, "ring no deeper than the loop");
#pragma unroll
            for (int k = 0; k < NS1 - 1; ++k) issue_p1(k, k);
            // software pipeline over the barrier: the fragments of stage ks are read while the MFMAs of stage ks-1 run (with
            // read-then-multiply inside one step all eight waves hit LDS together, then the matrix cores together:
            // 640 + 512 cycles per step in the trace instead of max(640, 512))
            static_assert(NP1W == 1 && (NK1 % 2) == 0, "fragment ping-pong: one px tile per wave, even step count");
            v4i wfa[2][CMW], xfa[2], wfb[2][CMW], xfb[2];
            auto p1_read = [&](const char* base, v4i (&wf)[2][CMW], v4i (&xf)[2]) {
#ifdef F8_ABL_NOREAD
                return;
#endif
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                    for (int i = 0; i < CMW; ++i) wf[kk][i] = *(const v4i*)(base + X1_BYTES + ((wb * CMW + i) * 32 + l31) * 64 + cof[kk]);
                    xf[kk] = *(const v4i*)(base + (wa * 32 + l31) * 64 + cof[kk]);
                }
            };
            auto p1_mul = [&](const v4i (&wf)[2][CMW], const v4i (&xf)[2]) {
#ifdef F8_ABL_NOMUL
                return;
#endif
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < CMW; ++i) acc[0][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[kk][i], xf[kk], acc[0][i], 0, 0, 0);
            };
            // (two stages per barrier as in P2 was tried here and lost 2 us per launch: the x8 stream is HBM-latency bound and a
            // 4-stage ring that validates two stages at a time drains at every barrier)
            // completely unrolled with compile-time step numbers (ring slots, DMA K offsets and wait counts are immediates)
            auto p1_step_c = [&](auto kc) {
                constexpr int KS = decltype(kc)::value;
                if constexpr (KS + NS1 - 2 < NK1) wait_vmcnt<(NS1 - 2) * L1>(); else wait_vmcnt<0>();   // stage KS landed; the next ones may fly
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if constexpr (KS + NS1 - 1 < NK1) issue_p1(KS + NS1 - 1, (KS + NS1 - 1) % NS1);
            };
            p1_step_c(std::integral_constant<int, 0>{});
            p1_read(p1ring, wfa, xfa);
            auto p1_pair = [&](auto uc) {
                constexpr int KS = 2 * decltype(uc)::value;
                p1_step_c(std::integral_constant<int, KS + 1>{});
                p1_read(p1ring + ((KS + 1) % NS1) * RING, wfb, xfb);
                pin(wfa, xfa);
                p1_mul(wfa, xfa);
                if constexpr (KS + 2 < NK1) {
                    p1_step_c(std::integral_constant<int, KS + 2>{});
                    p1_read(p1ring + ((KS + 2) % NS1) * RING, wfa, xfa);
                }
                pin(wfb, xfb);
                p1_mul(wfb, xfb);
            };
            static_for<NK1 / 2>(p1_pair);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                // every wave is done with the P1 ring before W2 stages land in it
        }
        F8_TT(1);
        if (DS) {   // the shortcut's x fragments (this wave's output pixel tile, all of K = C) from the P1 stage, before P2 reuses the slot
            const int op = wa * 32 + l31;
            const int prow = W + (op < OUT_PX ? op : OUT_PX - 1);        // P1 pixel index of the output pixel (skip the halo row)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) xs[kk] = *(const v4i*)((ALLW ? lds + (NK1 - 1) * RING : ring + ((NK1 - 1) & 1) * RING) + prow * 64 + (((kk * 2 + lh) ^ S64::f(prow)) << 4));
        }
        if constexpr (ALLW) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                // every wave is done reading the P1 stages
            // all of W2 (NK2 stages of MID x 64 bytes, stage-major = the order P2 consumes them) travels during the epilogue
#pragma unroll
            for (int i = 0; i < (NK2 * WS + 511) / 512; ++i) {
                const int gs = tid + i * 512;
                const int j2 = gs / WS, wi = gs - j2 * WS;
                const int row = wi >> 2, chunk = (wi & 3) ^ S64::f(row);
                const unsigned woff = (unsigned)(row * (9 * MID) + chunk * 16 + j2 * 64);
                if ((i * 512 + wave * 64) < NK2 * WS)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw2, (__attribute__((address_space(3))) void*)(ring + i * 8192 + wave * 1024), 16, woff, 0, 0, 0);
            }
            zero_patch();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                // the zero fill is complete before anyone writes real pixels
        } else {
            // the first W2 stages can already travel: slot 0 of the P1 ring was last read two steps ago (deep variant: the
            // whole P1 ring is dead since the barrier above)
#pragma unroll
            for (int k = 0; k < (NS1 > 2 ? RS2 - 1 : PRE2); ++k) issue_w2(k, k);
        }

        // epilogue: bias, ReLU, requant to body.2's input format, into the patch
        const int floor0 = a.relu_a ? 0 : INT32_MIN;
#pragma unroll
        for (int j = 0; j < NP1W; ++j) {
            const int pt = wa + 4 * j;
            if (pt >= NP1) continue;                     // wave-uniform
            const int pix = pt * 32 + l31;
            const int pr = pix / W, pc = pix - pr * W;
            const int hrow = p0 - 1 + pr;
            const bool ok = pix < P1_PX && hrow >= 0 && hrow < a.H;
            const int ppx = pr * PW + pc + 1;
#pragma unroll
            for (int i = 0; i < CMW; ++i) {
                const int cot = (wb * CMW + i) * 32;
                unsigned d[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int y[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = requant1(max(acc[j][i][4 * g + e], floor0), a.n1, a.lo1, a.hi1);
                    d[g] = pack4(y[0], y[1], y[2], y[3]) ^ a.xor1;
                }
                auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
                if (ok) {
                    v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
                    *(v4i*)(patch + SM::off(ppx, cot / 16 + lh)) = o;
                }
            }
        }
    }

    F8_TT(2);
    // ---- output pixel of this lane (px tile wa) and the residual stream of its (px tile, co tile wb)
    const int opix = wa * 32 + l31;
    const bool opix_ok = opix < rows_out * W;
    const int m = m_tile + opix;                         // global output pixel
    const int mc = opix_ok ? m : m_tile;                 // padding lanes: any valid pixel (loads only)
    v4i rv[4] = {}, rn[4] = {};                          // residual of the current / next 64-channel chunk (this wave: 32 ch)
    auto load_res = [&](v4i (&dst)[4], int c) {
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[g] = *(const v4i*)(a.xr + i32t_index(mc, c * 64 + wb * 32 + 8 * g + 4 * lh, C));
    };

    v4i rq[D3 == 4 ? 4 : 1][4] = {};                     // deep P3: residual chunks c .. c+3
    auto w4slot = [&](int slot) -> char* { return slot < 3 ? ring + slot * (MID * 64) : patch; };
    auto issue_w4d = [&](int c, int slot) {              // deep P3: 64 output channels x MID bytes into slot (c % 4)
        char* base = w4slot(slot);
#pragma unroll
        for (int j = 0; j < WL; ++j) {
            const unsigned woff = w4b[j] + (unsigned)(c * 64 * MID);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw4, (__attribute__((address_space(3))) void*)(base + j * 8192 + wave * 1024), 16, woff, 0, 0, 0);
        }
    };

    // =========================================================================================
    // P2: mid2 = requant(relu(conv3x3(mid1) + b2)) on R x W pixels  ->  mid2
    //     wave (wa, wb): px tile wa, co tiles {wb*CMW ..}
    // =========================================================================================
    {
        v16i acc[CMW];
        const int oc = opix < OUT_PX ? opix : OUT_PX - 1;   // padding lanes read a valid pixel, result unused
        const int orow = oc / W, ocol = oc - orow * W;
        const int bpx = orow * PW + ocol;                // patch pixel of tap (0,0)
        unsigned cof[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) cof[kk] = (unsigned)(((kk * 2 + lh) ^ S64::f(l31)) << 4);

        v4i bq2[CMW][4];
#pragma unroll
        for (int i = 0; i < CMW; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) bq2[i][g] = *(const v4i*)(a.b2 + (wb * CMW + i) * 32 + 8 * g + 4 * lh);
#pragma unroll
        for (int i = 0; i < CMW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = bq2[i][r >> 2][r & 3];
        constexpr int S0 = NK1 & 1;                      // ring slot of W2 step 0
        constexpr int CH = MID / 64;                     // 64-byte channel chunks per tap
        int tr = 0, ts = 0, tc = 0;
        auto p2_mma = [&](const char* base) {
            const int ppx = bpx + tr * PW + ts;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const v4i xf = *(const v4i*)(patch + SM::off(ppx, tc * 4 + kk * 2 + lh));
#pragma unroll
                for (int i = 0; i < CMW; ++i) {
                    const v4i wf = *(const v4i*)(base + ((wb * CMW + i) * 32 + l31) * 64 + cof[kk]);
                    acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, xf, acc[i], 0, 0, 0);
                }
            }
            if (++tc == CH) { tc = 0; if (++ts == 3) { ts = 0; ++tr; } }
        };
        if constexpr (ALLW) {
            wait_vmcnt<0>();                             // all of W2 landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                // ... everywhere, and the patch is complete
#pragma unroll
            for (int j = 0; j < NK2; ++j) p2_mma(ring + j * (MID * 64));
        } else {
            if constexpr (NS1 > 2) {
                // as in P1: fragments of step j are read while the MFMAs of step j-1 run
                static_assert((NK2 % 2) == 0, "fragment ping-pong");
                v4i wfa[2][CMW], xfa[2], wfb[2][CMW], xfb[2];
                auto p2_mul = [&](const v4i (&wf)[2][CMW], const v4i (&xf)[2]) {
#ifdef F8_ABL_NOMUL
                    return;
#endif
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int i = 0; i < CMW; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[kk][i], xf[kk], acc[i], 0, 0, 0);
                };
                // SEVERAL K stages per barrier: the fixed cost of a step (counted wait, barrier skew, DMA issue, address
                // arithmetic; ablation: ~1000 of 1360 cycles with neither reads nor MFMAs) is paid 18 times instead of 36
                // (49 k -> 39 k cycles).  A barrier validates the stages up to `upto` (stage 0: prologue);
                // stages <= `freed` are read by then, so stage s may be issued once s <= freed + RS2.
                // The loop is unrolled completely with compile-time stage numbers: ring slots, tap offsets, DMA offsets and wait
                // counts become immediates (the rolled loop spent ~70 VALU + ~50 SALU per 8 MFMAs on that arithmetic — with two
                // waves per SIMD that, not the matrix pipe, set the step time).
                const char* const wrow = mid2 + ((wb * CMW) * 32 + l31) * 64;   // this lane's row of co tile wb*CMW in slot 0
                const int pp[3] = {bpx, bpx + 1, bpx + 2};                        // patch pixel of tap column ts (row 0)
                auto p2_read_c = [&](auto jc, v4i (&wf)[2][CMW], v4i (&xf)[2]) {
#ifdef F8_ABL_NOREAD
                    return;
#endif
                    constexpr int J = decltype(jc)::value;
                    constexpr int TAP = J / CH, TC = J % CH, TR = TAP / 3, TS = TAP % 3, SLOT = J % RS2;
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        // SM::off(pp + TR*PW, c): PW == 16 keeps the swizzle term (row & 15) independent of TR
                        xf[kk] = *(const v4i*)(patch + TR * PW * MID + SM::off(pp[TS], TC * 4 + kk * 2 + lh));
#pragma unroll
                        for (int i = 0; i < CMW; ++i) wf[kk][i] = *(const v4i*)(wrow + SLOT * W2B + i * 32 * 64 + cof[kk]);
                    }
                };
                static_assert(PW == 16 && MID == 256, "patch rows of 16 pixels x 256 bytes: TR*PW pixels = TR*PW*MID bytes, swizzle unchanged");
                // (the DMA builtin stays in the function-scope lambda `issue_w2`: defined in a lambda inside this `if constexpr`
                // block, clang's host pass silently drops the kernel's host stub; stage and slot are constants after inlining)
                auto issue_w2_c = [&](auto jc) { constexpr int J2 = decltype(jc)::value; issue_w2(J2, J2 % RS2); };
                static_assert(NK2 * 64 < 4096, "W2 K offset fits the DMA's 12-bit immediate");
                // barrier u = 0..NK2/2-1 validates stages <= min(2u+2, NK2-1) and frees stages <= 2u; before it, stages
                // < B(u) = min(NK2, 2u + RS2 - 1) are issued (u = 0: RS2, the prologue barrier added one); after it, < B(u+1)
                auto p2_super_c = [&](auto uc) {
                    constexpr int U = decltype(uc)::value;
                    constexpr int UPTO = (2 * U + 2 < NK2) ? 2 * U + 2 : NK2 - 1;
                    constexpr int BEFORE = U == 0 ? RS2 : ((2 * U + RS2 - 1 < NK2) ? 2 * U + RS2 - 1 : NK2);
                    constexpr int AFTER = (2 * U + RS2 + 1 < NK2) ? 2 * U + RS2 + 1 : NK2;
                    constexpr int FLY = (BEFORE - 1 - UPTO) > 0 ? (BEFORE - 1 - UPTO) : 0;
                    wait_vmcnt<FLY * WL>();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if constexpr (BEFORE < AFTER) issue_w2_c(std::integral_constant<int, BEFORE>{});
                    if constexpr (BEFORE + 1 < AFTER) issue_w2_c(std::integral_constant<int, BEFORE + 1>{});
                    static_assert(AFTER - BEFORE <= 2, "at most two stages issued per barrier");
                };
                // prologue barrier: stage 0 landed (4 prologue stages may fly; the bias loads above are newer, which only makes
                // this wait conservative), the patch is complete; one more stage issued
                wait_vmcnt<(RS2 - 2) * WL>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                issue_w2_c(std::integral_constant<int, RS2 - 1>{});
                p2_read_c(std::integral_constant<int, 0>{}, wfa, xfa);
                auto p2_pair = [&](auto uc) {
                    constexpr int U = decltype(uc)::value, J = 2 * U;
                    p2_super_c(uc);
                    p2_read_c(std::integral_constant<int, J + 1>{}, wfb, xfb);
                    pin(wfa, xfa);
                    p2_mul(wfa, xfa);
                    if constexpr (J + 2 < NK2) p2_read_c(std::integral_constant<int, J + 2>{}, wfa, xfa);
                    pin(wfb, xfb);
                    p2_mul(wfb, xfb);
                };
                static_for<NK2 / 2>(p2_pair);
            } else {
                // unrolled with compile-time step numbers: ring slot, tap position, K offset and wait count are constants
                // (the rolled loop's switch / modulo / tap bookkeeping competed with the MFMAs for issue slots)
                static_assert((WS % 512) == 0, "every wave issues every W2 DMA instruction: wait counts are compile-time");
                static_for<NK2>([&](auto jc) {
                    constexpr int J = decltype(jc)::value;
                    constexpr int BEFORE = J == 0 ? PRE2 : ((J - 1 + NS2 < NK2) ? J - 1 + NS2 : NK2);
                    constexpr int AFTER = (J + NS2 < NK2) ? J + NS2 : NK2;
                    wait_vmcnt<(BEFORE - 1 - J) * WL>();     // stage J landed; the stages issued after it may stay in flight
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();            // step 0: also "patch complete" and "P1 ring free"
                    static_for<AFTER - BEFORE>([&](auto ic) { constexpr int S = BEFORE + decltype(ic)::value; issue_w2(S, S % NS2); });
                    constexpr int TAP = J / CH;
                    tr = TAP / 3; ts = TAP % 3; tc = J % CH;  // constants: p2_mma's address arithmetic folds
                    p2_mma(ring + (J % NS2) * W2B);
                });
            }
        }
        F8_TT(3);
        asm volatile("" ::: "memory");
        if constexpr (D3 == 4) {
            // the W4 ring takes over the W2 ring (3 slots) and the head of the patch: both must be done with everywhere
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#ifdef F8_ABL_P3_NORES
            issue_w4d(0, 0); issue_w4d(1, 1); issue_w4d(2, 2); asm volatile("" ::: "memory");
#else
            issue_w4d(0, 0); asm volatile("" ::: "memory"); load_res(rq[0], 0); asm volatile("" ::: "memory");
            issue_w4d(1, 1); asm volatile("" ::: "memory"); load_res(rq[1], 1); asm volatile("" ::: "memory");
            issue_w4d(2, 2); asm volatile("" ::: "memory"); load_res(rq[2], 2); asm volatile("" ::: "memory");
#endif
        } else {
            issue_w4(0, ALLW ? 0 : ((S0 + NK2) & 1));    // ALLW: the W4 ring is its own region (under the last P1 stage: free since the post-P1 barrier)
            asm volatile("" ::: "memory");   // the loads below must stay behind this DMA (counted wait in P3)
            if (!DS) load_res(rv, 0);
        }

        const int floor0 = a.relu_b ? 0 : INT32_MIN;
#pragma unroll
        for (int i = 0; i < CMW; ++i) {
            const int cot = (wb * CMW + i) * 32;
            unsigned d[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = requant1(max(acc[i][4 * g + e], floor0), a.n2, a.lo2, a.hi2);
                d[g] = pack4(y[0], y[1], y[2], y[3]) ^ a.xor2;
            }
            auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
            v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
            *(v4i*)(mid2 + SM::off(opix, cot / 16 + lh)) = o;
        }
    }

    F8_TT(4);
    // =========================================================================================
    // P3: y = clamp((W4 . mid2 + b4) << sa + (x << sr)) [ReLU]  ->  y32 (I32T) / int8 copies
    //     wave (wa, wb): px tile wa, co tile wb of each 64-channel chunk
    // =========================================================================================
    {
        constexpr int S0 = (NK1 + NK2) & 1;
        const int floor1 = a.relu1 ? 0 : -2147483647 /* the join's clamp_(min=-(2^31-1)) and the ReLU floor are one max */;
#ifdef F8_ABL_NOSTORE8_ALL
        const int n_store = (a.out32 ? 4 : 0);           // ablation: int8 stores removed everywhere
#else
        const int n_store = (a.out32 ? 4 : 0) + (a.q[0].ptr ? 1 : 0) + (a.q[1].ptr ? 1 : 0);   // store instructions per wave per chunk
#endif
        static_assert((NPO - 1) * 32 < OUT_PX, "every pixel tile has live lanes (store count in the counted wait)");
        v4i xf[KK3];                                     // this wave's mid2 fragments are chunk-invariant: read once
        // ---- int8 output through LDS (STG): a wave's int8 result of one chunk is 32 bytes per pixel, and 32-byte pieces at a
        // COUT-byte pitch cost as many write requests as the four contiguous 1 KB int32 stores together (ablation: removing
        // the int8 stores — a ninth of the bytes — took 19 % off the 56x56 kernel).  Two chunks (128 channels) are collected
        // in one of two LDS buffers [128 px][128 B] (16-byte columns XOR-swizzled by the row) and written out as full
        // 128-byte lines, 8 pixels per wave instruction, while the next pair fills the other buffer.
        // Measured (128-image launches): identity 56x56 block 208.5 -> 200.2 us; the DS instance (no residual stream, write-
        // bound) 114.6 -> 120.6 us, so it keeps its direct stores.  (Most of the ablation's 19 % turned out to be the bytes
        // themselves: the int8 copy is a tenth of a block's traffic, all of it writes.)
        // Buffers (16 KB each, regions that are dead in P3): MID = 64: the W2 region; MID = 128: the patch and — from the second pair on,
        // when every wave has read its chunk-invariant mid2 fragments (chunk 0) and passed chunk 1's barrier — mid2; MID = 256: the
        // patch behind the W4 slot in its head, and mid2 likewise.  Measured per 128 images (same box): 28x28 blocks 335 -> 324 us; the
        // 14x14 blocks 374 -> 385 us (their P3 is four chunks deep: the extra barrier-coupled LDS round trip costs more than the
        // 32-byte pieces), so MID = 256 keeps its direct stores, like the write-bound DS instance.
        constexpr bool STG = !DS && MID != 256;
        static_assert(!STG || (OUT_PX <= 128 && OUT_PX >= 64 && (NC3 % 2) == 0), "staging buffers / pairs of chunks");
        static_assert(!STG || !ALLW || W2ALL >= 2 * 16384, "MID = 64: both buffers in the W2 region");
        static_assert(!STG || ALLW || MID != 128 || (PATCH_BYTES >= 16384 && MID2_BYTES >= 16384), "MID = 128: patch / mid2");
        static_assert(!STG || ALLW || MID != 256 || (PATCH_BYTES >= 2 * 16384 && MID2_BYTES >= 16384), "MID = 256: patch tail / mid2");
        auto stgbuf = [&](int b) -> char* {
            if constexpr (ALLW) return ring + b * 16384;
            else if constexpr (MID == 256) return b == 0 ? patch + 16384 : mid2;
            else return b == 0 ? patch : mid2;
        };
        const bool stage0 = STG && a.q[0].ptr != nullptr;
        const int n_direct = (a.out32 ? 4 : 0) + ((!STG && a.q[0].ptr) ? 1 : 0) + (a.q[1].ptr ? 1 : 0);   // direct stores per wave per chunk
        auto stores_of = [&](int x) { return n_direct + ((stage0 && x >= 2 && !(x & 1)) ? 2 : 0); };       // VMEM stores issued during chunk x
        auto flush_pair = [&](int c0) {                  // chunks c0, c0+1 -> 128-byte lines; exactly two store instructions per wave
            const char* buf = stgbuf((c0 >> 1) & 1);
            const int npx = rows_out * W;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int t = tid + rr * 512;
                const int row = t >> 3, c16 = t & 7;
                const bool live = row < npx;
                const int rw = live ? row : npx - 1;     // dead rows: one lane re-writes the last row (keeps the instruction count uniform)
                if (live || lane == 0) {
                    const v4i v = *(const v4i*)(buf + rw * 128 + ((c16 ^ (rw & 7)) << 4));
                    *(v4i*)(a.q[0].ptr + (size_t)(m_tile + rw) * COUT + c0 * 64 + c16 * 16) = v;
                }
            }
        };
        // one chunk of 64 output channels; `cur` holds this chunk's residual, `nxt` receives the next one's
        auto chunk = [&](int c, v4i (&cur)[4], v4i (&nxt)[4]) {
            // W4 chunk c landed?  VMEM retires in order, so exactly the operations issued AFTER that DMA may stay in
            // flight: the 4 residual-prefetch loads (fenced behind it) and, from chunk 1 on, the previous chunk's
            // stores (4 for the int32 form + 1 per int8 form; every wave has live lanes, so all of them issue).
            // A smaller count would be safe but would drain the residual prefetch on every chunk (measured: P3
            // 30k -> cycles per tile); a larger one would race.
            // (DS: no residual prefetch, only the previous chunk's stores are newer than the DMA)
            wait_vmcnt_dyn((DS ? 0 : 4) + (c == 0 ? 0 : stores_of(c - 1)));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                // chunk 0: also "mid2 complete"
            const int cot = c * 64 + wb * 32;
            v4i bq4[4], bqs[DS ? 4 : 1];                 // this chunk's biases: requested before the DMA / prefetch below
            if (DS) {                                    // ... or read from LDS (visible since the P1 / P2 barriers)
#pragma unroll
                for (int g = 0; g < 4; ++g) { bq4[g] = *(const v4i*)(bias_lds + cot + 8 * g + 4 * lh); bqs[g] = *(const v4i*)(bias_lds + COUT + cot + 8 * g + 4 * lh); }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) bq4[g] = *(const v4i*)(a.b4 + cot + 8 * g + 4 * lh);
            }
            asm volatile("" ::: "memory");
            if (c + 1 < NC3) issue_w4(c + 1, ALLW ? ((c + 1) & 1) : ((S0 + c + 1) & 1));
            asm volatile("" ::: "memory");
            if (!DS) load_res(nxt, c + 1 < NC3 ? c + 1 : c);   // always 4 loads per wave: the counted wait relies on it
            if (stage0 && c >= 2 && !(c & 1)) flush_pair(c - 2);   // the pair before this one is complete since this chunk's barrier
            const char* base = w4ring + (ALLW ? (c & 1) : ((S0 + c) & 1)) * W4STRIDE;
            if (c == 0) {
#pragma unroll
                for (int kk = 0; kk < KK3; ++kk) xf[kk] = *(const v4i*)(mid2 + SM::off(opix, kk * 2 + lh));
            }
            v16i acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bq4[r >> 2][r & 3];
#pragma unroll
            for (int kk = 0; kk < KK3; ++kk) {
                const v4i wf = *(const v4i*)(base + SM::off(wb * 32 + l31, kk * 2 + lh));
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, xf[kk], acc, 0, 0, 0);
            }
            v16i acs;                                    // DS: shortcut product Wsc . x of this chunk
            if (DS) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acs[r] = bqs[r >> 2][r & 3];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const v4i wf = *(const v4i*)(base + MID * 64 + (wb * 32 + l31) * 64 + (((kk * 2 + lh) ^ S64::f(l31)) << 4));
                    acs = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, xs[kk], acs, 0, 0, 0);
                }
            }
            int y[4][4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned v = (unsigned)acc[4 * g + e];                            // body.4 (the accumulator started at its bias)
                    unsigned o = (unsigned)cur[g][e];                                  // identity block: the block input
                    if (DS) { o = v; v = (unsigned)acs[4 * g + e]; }                   // DS: the shortcut conv hosts the join
                    const unsigned s = (v << a.acc_shl) + (o << a.res_shl);
                    y[g][e] = max((int)s, floor1);
                }
            }
            if (a.out32 && opix_ok) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    v4i o = {y[g][0], y[g][1], y[g][2], y[g][3]};
                    *(v4i*)(a.out32 + i32t_index(m, cot + 8 * g + 4 * lh, COUT)) = o;
                }
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
#ifdef F8_ABL_NOSTORE8_ALL
                if (s0[0] == 0x12345678 && s1[1] == 0x7654321)
#endif
                const v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
                if (STG && k == 0) {                     // 16 bytes of pixel `opix`, column (c & 1) * 64 + wb * 32 + lh * 16 of its 128-byte row
                    const int c16 = (c & 1) * 4 + wb * 2 + lh;
                    *(v4i*)(stgbuf((c >> 1) & 1) + opix * 128 + ((c16 ^ (opix & 7)) << 4)) = o;
                } else if (opix_ok) {
                    *(v4i*)(a.q[k].ptr + (size_t)m * COUT + cot + 16 * lh) = o;
                }
            }
        };
        // deep variant of `chunk`: slot / buffer index k = c % 4 is a compile-time constant of the unrolled body
        auto chunk4 = [&](int c, auto kc) {
            constexpr int k = decltype(kc)::value;
            // newer than res(c) in the queue: per earlier chunk of the window its stores, and the W4 + residual requests of the
            // two later chunks (WL + 4 each); everything older — including the stores of chunk c-4 — has to be back
            const int st = c < 3 ? c : 3;
            int st_sum = 0;                              // stores issued by the (up to three) chunks before this one
            for (int j = 1; j <= st; ++j) st_sum += stores_of(c - j);
#if defined(F8_ABL_P3_NORES)
            wait_vmcnt_dyn(2 * (WL + 0) + st * n_store);
#elif defined(F8_ABL_P3_NOSTORE)
            wait_vmcnt_dyn(2 * (WL + 4));
#elif defined(F8_ABL_P3_NOSTORE8)
            wait_vmcnt_dyn(2 * (WL + 4) + st * (n_store - 1));
#elif defined(F8_ABL_P3_NOSTORE32)
            wait_vmcnt_dyn(2 * (WL + 4) + st * (n_store - 4));
#else
            (void)n_store;
            wait_vmcnt_dyn(2 * (WL + 4) + st_sum);
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                // chunk 0: also "mid2 complete"; slot (c+3)%4 was read in chunk c-1
            const int cot = c * 64 + wb * 32;
            v4i bq4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bq4[g] = *(const v4i*)(bias_lds + cot + 8 * g + 4 * lh);
            asm volatile("" ::: "memory");
            {   // always issued (the tail re-requests the last chunk): the counted wait relies on a fixed count per chunk
                const int cn = c + 3 < NC3 ? c + 3 : NC3 - 1;
                issue_w4d(cn, (k + 3) & 3);
                asm volatile("" ::: "memory");
#ifndef F8_ABL_P3_NORES
                load_res(rq[(k + 3) & 3], cn);
#endif
            }
            if (stage0 && c >= 2 && !(c & 1)) flush_pair(c - 2);   // the pair before this one is complete since this chunk's barrier
            const char* base = w4slot(k);
            if (c == 0) {
#pragma unroll
                for (int kk = 0; kk < KK3; ++kk) xf[kk] = *(const v4i*)(mid2 + SM::off(opix, kk * 2 + lh));
            }
            v16i acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bq4[r >> 2][r & 3];
#pragma unroll
            for (int kk = 0; kk < KK3; ++kk) {
#ifndef F8_ABL_P3_NOMMA
                const v4i wf = *(const v4i*)(base + SM::off(wb * 32 + l31, kk * 2 + lh));
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, xf[kk], acc, 0, 0, 0);
#endif
            }
            int y[4][4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned sres = ((unsigned)acc[4 * g + e] << a.acc_shl) + ((unsigned)rq[k][g][e] << a.res_shl);
                    y[g][e] = max((int)sres, floor1);
                }
#if defined(F8_ABL_P3_NOSTORE) || defined(F8_ABL_P3_NOSTORE32)
            if (y[0][0] == 0x12345678 && y[3][3] == 0x7654321 && y[1][2] == 77 && y[2][1] == 78)
#endif
            if (a.out32 && opix_ok) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    v4i o = {y[g][0], y[g][1], y[g][2], y[g][3]};
                    *(v4i*)(a.out32 + i32t_index(m, cot + 8 * g + 4 * lh, COUT)) = o;
                }
            }
#pragma unroll
            for (int kq = 0; kq < 2; ++kq) {
                if (!a.q[kq].ptr) continue;
                unsigned d[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    d[g] = pack4(requant1(y[g][0], a.q[kq].n, a.q[kq].lo, a.q[kq].hi), requant1(y[g][1], a.q[kq].n, a.q[kq].lo, a.q[kq].hi),
                                 requant1(y[g][2], a.q[kq].n, a.q[kq].lo, a.q[kq].hi), requant1(y[g][3], a.q[kq].n, a.q[kq].lo, a.q[kq].hi)) ^ a.q[kq].bias_xor;
                auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
#if defined(F8_ABL_P3_NOSTORE) || defined(F8_ABL_P3_NOSTORE8)
                if (s0[0] == 0x12345678 && s1[1] == 0x7654321)
#endif
                const v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
                if (STG && kq == 0) {                    // staged: 16 bytes of pixel opix, column (c & 1) * 64 + wb * 32 + lh * 16 of its 128-byte row
                    const int c16 = (c & 1) * 4 + wb * 2 + lh;
                    *(v4i*)(stgbuf((c >> 1) & 1) + opix * 128 + ((c16 ^ (opix & 7)) << 4)) = o;
                } else if (opix_ok) {
                    *(v4i*)(a.q[kq].ptr + (size_t)m * COUT + cot + 16 * lh) = o;
                }
            }
        };
        if constexpr (D3 == 4) {
            static_assert(D3 != 4 || NC3 % 4 == 0, "chunk loop is unrolled by four");
            // (unrolling all 16 chunks with constant chunk numbers measured 1.4 us slower per launch: ~5k more instructions)
            for (int c = 0; c < NC3; c += 4) {
                chunk4(c, std::integral_constant<int, 0>{});
                chunk4(c + 1, std::integral_constant<int, 1>{});
                chunk4(c + 2, std::integral_constant<int, 2>{});
                chunk4(c + 3, std::integral_constant<int, 3>{});
            }
            if (stage0) {                                // the last pair
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                flush_pair(NC3 - 2);
            }
        } else {
            for (int c = 0; c < NC3; c += 2) {
                chunk(c, rv, rn);
                chunk(c + 1, rn, rv);
            }
            if (stage0) {                                // the last pair
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                flush_pair(NC3 - 2);
            }
        }
    }
#ifdef F8_TRACE
    if (a.trace && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tt[5] = __builtin_readcyclecounter();
        unsigned long long* tp = (unsigned long long*)a.trace + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 6; ++i) tp[i] = tt[i];
    }
#endif
}

template <int C, int MID, int W, int R, int COUT = C, bool DS = false>
static hipError_t launch_fused_t(const FusedArgs& a, hipStream_t s) {
    using Cfg = FusedCfg<C, MID, W, R>;
    constexpr int NK1 = C / 64, W2ALL = 9 * (MID / 64) * (MID * 64), W4STAGE = (MID * 64) * (DS ? 2 : 1);      // keep in sync with the kernel
    constexpr int ALLW_END = (NK1 * Cfg::RING > Cfg::PATCH_BYTES + Cfg::MID2_BYTES + W2ALL + 2 * W4STAGE) ? NK1 * Cfg::RING
                                                                                                   : Cfg::PATCH_BYTES + Cfg::MID2_BYTES + W2ALL + 2 * W4STAGE;
    constexpr int LDS = (MID == 64 ? ALLW_END : (MID == 256 ? Cfg::PATCH_BYTES + 4 * Cfg::RING : Cfg::LDS_BYTES)) + (DS ? 2 * COUT * 4 : 0) + ((MID == 256 && !DS) ? COUT * 4 : 0);   // + bias_lds; MID == 256: 4-stage P1 ring
    static_assert(LDS <= 80 * 1024 || MID > 64, "two workgroups per CU");
    static unsigned long long attr_done = 0; int attr_dev = -1;
    if (!dyn_lds_opted_in(&attr_done, &attr_dev)) {   // dynamic LDS above 64 KB must be opted into once per kernel
        hipError_t e = hipFuncSetAttribute((const void*)fused_bottleneck_kernel<C, MID, W, R, COUT, DS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        if (attr_dev >= 0) attr_done |= 1ull << attr_dev;
    }
    const int grid = a.N * a.tiles_per_img;
#ifdef F8_TRACE
    static unsigned long long* tbuf = nullptr; static int count = 0;
    static const int want = [] { const char* e = getenv("F8_TRACE_FUSED"); return e ? atoi(e) : -1; }();
    FusedArgs b = a;
    const bool tracing = (count++ == want);
    if (tracing) { if (!tbuf) (void)hipMalloc((void**)&tbuf, (size_t)1 << 22); (void)hipMemsetAsync(tbuf, 0, (size_t)grid * 64, s); b.trace = tbuf; }
    hipLaunchKernelGGL((fused_bottleneck_kernel<C, MID, W, R, COUT, DS>), dim3(grid), dim3(512), LDS, s, b);
    if (tracing) {
        (void)hipStreamSynchronize(s);
        unsigned long long* h = new unsigned long long[(size_t)grid * 8];
        (void)hipMemcpy(h, tbuf, (size_t)grid * 64, hipMemcpyDeviceToHost);
        double ph[5] = {0, 0, 0, 0, 0}; int n = 0;
        for (int i = 0; i < grid; ++i) { unsigned long long* p = h + (size_t)i * 8; if (!p[5]) continue; ++n; for (int k = 0; k < 5; ++k) ph[k] += (double)(p[k + 1] - p[k]); }
        fprintf(stderr, "[trace fused<%d,%d,%d,%d>] grid %d: avg cycles per WG: P1 loop %.0f | P1 epi %.0f | P2 loop %.0f | P2 epi %.0f | P3 %.0f\n", C, MID, W, R, grid,
                ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n);
        delete[] h;
    }
    return hipGetLastError();
#else
    hipLaunchKernelGGL((fused_bottleneck_kernel<C, MID, W, R, COUT, DS>), dim3(grid), dim3(512), LDS, s, a);
    return hipGetLastError();
#endif
}

hipError_t launch_fused_bottleneck(const FusedArgs& a, hipStream_t s) {
    if (a.wsc) {
        if (a.C == 64 && a.MID == 64 && a.COUT == 256 && a.W == 56 && a.R == 2) return launch_fused_t<64, 64, 56, 2, 256, true>(a, s);
        return hipErrorInvalidValue;
    }
    if (a.C == 256 && a.MID == 64 && a.W == 56 && a.R == 2) return launch_fused_t<256, 64, 56, 2>(a, s);
    if (a.C == 512 && a.MID == 128 && a.W == 28 && a.R == 4) return launch_fused_t<512, 128, 28, 4>(a, s);
    if (a.C == 1024 && a.MID == 256 && a.W == 14 && a.R == 7) return launch_fused_t<1024, 256, 14, 7>(a, s);
    return hipErrorInvalidValue;
}

// stage-opening bottleneck at unchanged resolution (1x1 -> 3x3 -> [1x1 + 1x1 shortcut]): the ResNet-50 stage-0 shape
bool fused_ds_supported(int C, int MID, int COUT, int H, int W, int* R) {
    if (C == 64 && MID == 64 && COUT == 256 && W == 56 && H % 2 == 0) { *R = 2; return true; }
    return false;
}

bool fused_bottleneck_supported(int C, int MID, int H, int W, int imgs_per_launch, int mask, int* R) {
    // mask (Options::fuse_stages): bit s = stage s.  -1: stages 0 and 1 always, stage 2 when one launch fills at least half the chip.
    if ((mask & 1) && C == 256 && MID == 64 && W == 56 && H % 2 == 0) { *R = 2; return true; }
    if ((mask & 2) && C == 512 && MID == 128 && W == 28 && H % 4 == 0) { *R = 4; return true; }
    if (C == 1024 && MID == 256 && W == 14 && H % 7 == 0) {
        // One workgroup per CU (124 KB LDS), two tiles of 98 px per image: measured against the unfused three launches,
        // two concurrent sub-batches, ResNet-50 img/s:  bs 256 (256 workgroups per launch) 74.9k vs 71.6k; bs 128 (128)
        // 71.2k vs 70.7k; bs 64 (64) 58.6k vs 63.6k; bs 32 42.7k vs 49.7k — a launch that leaves most CUs without a
        // workgroup loses to three launches of smaller tiles.  (With the 2-deep P1 / P3 pipelines it lost 4 % at bs 128.)
        const bool fill = imgs_per_launch * (H / 7) >= 128;
        if (mask < 0 ? fill : (mask & 4) != 0) { *R = 7; return true; }
    }
    return false;
}

}  // namespace f8
