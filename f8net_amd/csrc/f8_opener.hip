// f8_opener.hip — one launch for a ResNet bottleneck stage-OPENING block with a stride-2 3x3 (gfx950).
//
//   x8 (int8 NHWC, C ch, H x W) --1x1 C->MID, ReLU--> mid1 (H x W) --3x3 / 2, pad 1, ReLU--> mid2 (H/2 x W/2)
//        --1x1 MID->COUT-->  join  <--1x1 / 2 C->COUT (shortcut)-- x8 at the even pixels
//   join: align shifts, wrapping add, clamp, ReLU -> y32 (I32T) and/or requantised int8 copies
//
// i.e. IntBlock.forward of /root/reference/models/fix_resnet.py:26-77 for the first Bottleneck of stages 1..3
// (:233-319: stride on the 3x3, shortcut conv at the same stride), every int_op_only_fix_quant
// (fix_quant_ops.py:90-114) in place.
//
// Why: unfused the block is three launches (1x1, strided 3x3, dual GEMM) at 0.3 of the HBM roofline; mid1 (the
// block's largest intermediate, H x W x MID) is written and re-read through HBM and the two small convs are
// prologue / epilogue bound.  Fused, the block reads x8 once and writes its outputs once.
//
// Work unit: R output rows x full output width WO = W/2 of ONE image.
//   P1  mid1 for input rows 2*p0-1 .. 2*p0+2R-1 (2R+1 rows: one halo row above, none below), GEMM
//       [(2R+1)*W px] x [MID] x K = C; x8 / W0 stream through a 2-stage LDS ring (LDS-direct DMA).  The result is
//       requantised into an LDS patch whose columns are DE-INTERLEAVED (even input-column phase, then odd): a
//       stride-2 tap then reads consecutive patch entries for consecutive output pixels (conflict-free b128 reads).
//       Patch border (left column, top row of the image) = biased zero.
//   P2  3x3 / 2 from the patch, W2 streams through a 5-stage ring; result -> LDS mid2.
//   P3  chunks of 64 output channels: W4 chunk (K = MID) + shortcut chunk (K = C) stream through a 3-stage ring;
//       the shortcut's x operand (the tile's even pixels, all of C) was fetched into LDS by one DMA gather during P2
//       and sits in registers; epilogue = join + int32 / int8 stores.  No residual stream: the phase is write-bound,
//       so weight tiles are requested two chunks ahead (a chunk waits for the stores issued three chunks ago).
// 512 threads = 8 waves in a 4 (pixel tiles) x 2 (channel tiles) grid, one workgroup per CU (~157 KB LDS).
#include "f8_device.h"
#include <cstdlib>
#include <cstdio>

namespace f8 {

template <int C, int MID, int W, int R, int COUT>
struct OpenerCfg {
    static constexpr int WO = W / 2;
    static constexpr int PR = 2 * R + 1;
    static constexpr int NE = WO + 1, NO = WO;          // patch column pc = input column + 1 in 0..W: even pc first, then odd
    static constexpr int PW = NE + NO;
    static constexpr int P1_PX = PR * W, NP1 = (P1_PX + 31) / 32;
    static constexpr int OUT_PX = R * WO, NPO = (OUT_PX + 31) / 32;
    static constexpr int PATCH_BYTES = (PR * PW * MID + 255) / 256 * 256;
    static constexpr int X1_BYTES = NP1 * 32 * 64, W0_BYTES = MID * 64, STAGE1 = X1_BYTES + W0_BYTES;
    static constexpr int MID2_BYTES = NPO * 32 * MID, XS_BYTES = NPO * 32 * C;
    static constexpr int W2B = MID * 64, NS2 = 5;
    static constexpr int REG_B = MID2_BYTES + XS_BYTES + NS2 * W2B;
    static constexpr int REG_A = 2 * STAGE1 > REG_B ? 2 * STAGE1 : REG_B;       // P1 ring, later mid2 | xs | W2 ring
    static constexpr int BIAS_INTS = 2 * MID + 2 * COUT;                          // b0, b2, b4, bsc
    static constexpr int LDS_BYTES = PATCH_BYTES + REG_A + BIAS_INTS * 4;
    static constexpr int STAGE3 = 64 * MID + 64 * C;                              // one P3 stage: W4 chunk + shortcut chunk
    static constexpr int STG_BYTES = NPO * 32 * 128;                              // int8 staging buffer: [px][128 B]
};

// P12 (round 4): only body.0 and body.2 — mid2 (body.2's output in body.4's int8 input format) goes to HBM (a.q[0], NHWC, MID channels) and
// the launch ends; the join of the block (body.4 + strided shortcut) is then the FIRST BLOCK of the stage's chain launch (f8_chain.hip, TAIL),
// which keeps its result in registers: the block's 205 MB int32 output (per 128 images) is neither written here nor read back there.
// FQ (P12 instances; round 4): both requantisations are ReLU -> unsigned 8-bit right shifts: the bias is the accumulators' start value, the ReLU the clamp's
// lower bound, and the shift runs through the float converter (FQ = 1, bounded accumulators, shifts <= 16: 3 operations per value) or as requant_shr +
// packing (FQ = 2) instead of bias add + max + requant1 + packing (10): P1's epilogue requantises 504 x 128 values per tile.
template <int C, int MID, int W, int R, int COUT, bool STG, bool P12 = false, int FQ = 0>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
fused_opener_kernel(const FusedArgs a) {
    using Cfg = OpenerCfg<C, MID, W, R, COUT>;
    constexpr int WO = Cfg::WO, PR = Cfg::PR, NE = Cfg::NE, PW = Cfg::PW;
    constexpr int P1_PX = Cfg::P1_PX, NP1 = Cfg::NP1, OUT_PX = Cfg::OUT_PX, NPO = Cfg::NPO;
    constexpr int CM = MID / 32, CMW = CM / 2, NP1W = (NP1 + 3) / 4;
    constexpr int NK1 = C / 64, NK2 = 9 * (MID / 64), NC3 = COUT / 64, KK3 = MID / 32, KKS = C / 32, CH = MID / 64;
    constexpr int X1_BYTES = Cfg::X1_BYTES, STAGE1 = Cfg::STAGE1, W2B = Cfg::W2B, NS2 = Cfg::NS2, STAGE3 = Cfg::STAGE3;
    static_assert(NPO == 4 && CM % 2 == 0 && NP1 % 4 == 0, "4 pixel-tile groups x 2 channel-tile groups");
    static_assert(MID % 64 == 0 && C % 64 == 0 && COUT % 128 == 0, "K steps of 64 bytes, chunk pairs");
    static_assert(Cfg::LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(2 * STAGE3 <= Cfg::PATCH_BYTES && STAGE3 <= NS2 * W2B, "P3 ring: two stages over the patch, one over the W2 ring");
    static_assert(!STG || (Cfg::PATCH_BYTES - 2 * STAGE3 >= Cfg::STG_BYTES && NS2 * W2B - STAGE3 >= Cfg::STG_BYTES), "staging buffers");
    constexpr int XL1 = X1_BYTES / 16 / 512;             // DMA instructions per thread: P1 X tile
    constexpr int WL0 = (MID * 4 + 511) / 512;           // W0 / W2 stage (MID rows x 64 B)
    constexpr int WL4 = 64 * MID / 16 / 512, WLS = 64 * C / 16 / 512, XSL = NPO * 32 * C / 16 / 512;
    static_assert(X1_BYTES % 8192 == 0 && (MID * 4) % 512 == 0 && WL4 >= 1 && WLS >= 1, "every wave issues every DMA instruction (compile-time wait counts)");
    constexpr int L1 = XL1 + WL0, L3 = WL4 + WLS;

    if constexpr (FQ == 1) set_fp_round_nearest_even();
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const patch = lds;
    char* const regA = lds + Cfg::PATCH_BYTES;
    char* const mid2 = regA;
    char* const xs = regA + Cfg::MID2_BYTES;
    char* const w2ring = xs + Cfg::XS_BYTES;
    int* const bias_lds = (int*)(regA + Cfg::REG_A);     // b0 | b2 | b4 | bsc
    auto p3slot = [&](int s) -> char* { return s < 2 ? patch + s * STAGE3 : w2ring; };
    auto stgbuf = [&](int b) -> char* { return b == 0 ? patch + 2 * STAGE3 : w2ring + STAGE3; };

    using S64 = Swz<64>;
    using SM = Swz<MID>;
    using SC = Swz<C>;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
    const int wa = wave >> 1, wb = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    int t;
    {   // XCD-aware order: consecutive tiles (vertically adjacent row groups, sharing one halo row) stay on one XCD's L2
        const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
        t = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    }
    const int HO = a.H >> 1;
    const int n = t / a.tiles_per_img, ti = t - n * a.tiles_per_img;
    const int p0 = ti * R;                               // first output row of the tile
    const int gp1 = (n * a.H + 2 * p0 - 1) * W;          // global input pixel of P1 pixel 0 (row 2*p0-1; < 0 only where masked)
    const int m_tile = (n * HO + p0) * WO;               // global output pixel of the tile's pixel 0

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x8, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w0, 0, a.w0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2, 0, a.w2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw4 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w4, 0, a.w4_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwsc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wsc, 0, a.wsc_bytes, 0x00020000);

#ifdef F8_TRACE
    unsigned long long tt[8]; tt[0] = __builtin_readcyclecounter();
#define F8_OT(i) tt[i] = __builtin_readcyclecounter()
#else
#define F8_OT(i)
#endif

    // ---- biases -> LDS (plain loads; complete before P1's first barrier, which waits for vmcnt(0))
    for (int i = tid; i < (P12 ? 2 * MID : Cfg::BIAS_INTS); i += 512) {
        int v;
        if (i < MID) v = a.b0[i];
        else if (i < 2 * MID) v = a.b2[i - MID];
        else if (i < 2 * MID + COUT) v = a.b4[i - 2 * MID];
        else v = a.bsc[i - 2 * MID - COUT];
        bias_lds[i] = v;
    }
    // ---- patch border = biased zero: column 0 (input column -1) of every row; row 0 when it lies above the image
    {
        const unsigned z = a.xor1;
        const v4i zv = {(int)z, (int)z, (int)z, (int)z};
        if (tid < PR * (MID / 16)) *(v4i*)(patch + (tid / (MID / 16)) * PW * MID + (tid % (MID / 16)) * 16) = zv;
        if (p0 == 0)
            for (int o = tid * 16; o < PW * MID; o += 512 * 16) *(v4i*)(patch + o) = zv;
    }

    // ---- gather descriptors
    unsigned xb1[XL1];
#pragma unroll
    for (int i = 0; i < XL1; ++i) {
        const int idx = tid + i * 512;
        const int row = idx >> 2, chunk = (idx & 3) ^ S64::f(row);
        const int pr = row / W;
        const bool ok = row < P1_PX && (2 * p0 - 1 + pr) >= 0;
        xb1[i] = ok ? (unsigned)((gp1 + row) * C + chunk * 16) : kOOB;
    }
    unsigned w0b[WL0], w2b[WL0];
#pragma unroll
    for (int j = 0; j < WL0; ++j) {
        const int idx = tid + j * 512;
        const int row = idx >> 2, chunk = (idx & 3) ^ S64::f(row);
        w0b[j] = (unsigned)(row * C + chunk * 16);
        w2b[j] = (unsigned)(row * (9 * MID) + chunk * 16);
    }
    unsigned w4b[WL4], wsb[WLS], xsb[XSL];
#pragma unroll
    for (int j = 0; j < WL4; ++j) {
        const int idx = tid + j * 512;
        const int row = idx / (MID / 16), c = (idx % (MID / 16)) ^ SM::f(row);
        w4b[j] = (unsigned)(row * MID + c * 16);
    }
#pragma unroll
    for (int j = 0; j < WLS; ++j) {
        const int idx = tid + j * 512;
        const int row = idx / (C / 16), c = (idx % (C / 16)) ^ SC::f(row);
        wsb[j] = (unsigned)(row * C + c * 16);
    }
#pragma unroll
    for (int j = 0; j < XSL; ++j) {
        const int idx = tid + j * 512;
        const int row = idx / (C / 16), c = (idx % (C / 16)) ^ SC::f(row);
        const int orow = row / WO, ocol = row - orow * WO;
        xsb[j] = row < OUT_PX ? (unsigned)(((n * a.H + 2 * (p0 + orow)) * W + 2 * ocol) * C + c * 16) : kOOB;
    }

    auto issue_p1 = [&](int ks, int slot) {
        char* base = regA + slot * STAGE1;
#pragma unroll
        for (int i = 0; i < XL1; ++i) {
            const unsigned off = xb1[i] + (unsigned)(ks * 64);   // kOOB + ks*64 stays beyond any buffer
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(base + i * 8192 + wave * 1024), 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < WL0; ++j) {
            const unsigned woff = w0b[j] + (unsigned)(ks * 64);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw0, (__attribute__((address_space(3))) void*)(base + X1_BYTES + j * 8192 + wave * 1024), 16, woff, 0, 0, 0);
        }
    };
    auto issue_w2 = [&](int j2, int slot) {              // bytes [j2*64, j2*64+64) of every W2 row (tap-major K)
        char* base = w2ring + slot * W2B;
#pragma unroll
        for (int j = 0; j < WL0; ++j) {
            const unsigned woff = w2b[j] + (unsigned)(j2 * 64);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw2, (__attribute__((address_space(3))) void*)(base + j * 8192 + wave * 1024), 16, woff, 0, 0, 0);
        }
    };
    auto issue_xs = [&]() {
#pragma unroll
        for (int j = 0; j < XSL; ++j) {
            const unsigned off = xsb[j];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(xs + j * 8192 + wave * 1024), 16, off, 0, 0, 0);
        }
    };
    auto issue_p3 = [&](int c, int slot) {               // 64 output channels: W4 rows (MID bytes), then shortcut rows (C bytes)
        char* base = p3slot(slot);
#pragma unroll
        for (int j = 0; j < WL4; ++j) {
            const unsigned woff = w4b[j] + (unsigned)(c * 64 * MID);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw4, (__attribute__((address_space(3))) void*)(base + j * 8192 + wave * 1024), 16, woff, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < WLS; ++j) {
            const unsigned woff = wsb[j] + (unsigned)(c * 64 * C);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rwsc, (__attribute__((address_space(3))) void*)(base + 64 * MID + j * 8192 + wave * 1024), 16, woff, 0, 0, 0);
        }
    };

    unsigned cof[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) cof[kk] = (unsigned)(((kk * 2 + lh) ^ S64::f(l31)) << 4);

    // =========================================================================================
    // P1: mid1 = requant(relu(W0 . x8 + b0)) on (2R+1) x W pixels -> patch
    //     wave (wa, wb): px tiles {wa, wa+4, ...}, co tiles {wb*CMW ..}
    // =========================================================================================
    {
        v16i acc[NP1W][CMW];
#pragma unroll
        for (int i = 0; i < CMW; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v4i bv = {0, 0, 0, 0};
                if constexpr (FQ != 0) bv = *(const v4i*)(a.b0 + (wb * CMW + i) * 32 + 8 * g + 4 * lh);     // FQ: bias = the accumulators' start value
#pragma unroll
                for (int j = 0; j < NP1W; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][i][4 * g + e] = bv[e];
            }
        issue_p1(0, 0);
        static_for<NK1>([&](auto kc) {
            constexpr int KS = decltype(kc)::value;
            wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if constexpr (KS + 1 < NK1) issue_p1(KS + 1, (KS + 1) & 1);
            const char* base = regA + (KS & 1) * STAGE1;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                v4i wf[CMW], xf[NP1W];
#pragma unroll
                for (int i = 0; i < CMW; ++i) wf[i] = *(const v4i*)(base + X1_BYTES + ((wb * CMW + i) * 32 + l31) * 64 + cof[kk]);
#pragma unroll
                for (int j = 0; j < NP1W; ++j) xf[j] = *(const v4i*)(base + ((wa + 4 * j) * 32 + l31) * 64 + cof[kk]);
#pragma unroll
                for (int j = 0; j < NP1W; ++j)
#pragma unroll
                    for (int i = 0; i < CMW; ++i) acc[j][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[i], xf[j], acc[j][i], 0, 0, 0);
            }
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // every wave is done with the P1 ring: mid2 / xs / W2 ring take its place
        F8_OT(1);
        if constexpr (!P12) issue_xs();                  // oldest in the queue: any later counted wait covers it
#pragma unroll
        for (int k = 0; k < NS2 - 1; ++k) issue_w2(k, k);

        const int floor0 = a.relu_a ? 0 : INT32_MIN;
        const float sc1 = FQ == 1 ? requant_u8_scale(a.n1) : 0.0f;
        (void)floor0; (void)sc1;
#pragma unroll
        for (int j = 0; j < NP1W; ++j) {
            const int pix = (wa + 4 * j) * 32 + l31;
            const int pr = pix / W, pc = pix - pr * W + 1;
            const bool ok = pix < P1_PX && (2 * p0 - 1 + pr) >= 0;
            const int ent = pr * PW + ((pc & 1) ? NE + (pc >> 1) : (pc >> 1));
#pragma unroll
            for (int i = 0; i < CMW; ++i) {
                const int ct = wb * CMW + i;
                unsigned d[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if constexpr (FQ != 0) {
                        d[g] = requant_u8x4_sel<FQ == 2 ? 2 : 1>(acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3], a.n1, sc1) ^ 0x80808080u;
                    } else {
                        const v4i bv = *(const v4i*)(bias_lds + ct * 32 + 8 * g + 4 * lh);
                        int y[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = requant1(max((int)((unsigned)acc[j][i][4 * g + e] + (unsigned)bv[e]), floor0), a.n1, a.lo1, a.hi1);
                        d[g] = pack4(y[0], y[1], y[2], y[3]) ^ a.xor1;
                    }
                }
                auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
                if (ok) {
                    v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
                    *(v4i*)(patch + SM::off(ent, ct * 2 + lh)) = o;
                }
            }
        }
    }
    F8_OT(2);

    const int opix = wa * 32 + l31;
    const bool opix_ok = opix < OUT_PX;
    const int m = m_tile + opix;

    // =========================================================================================
    // P2: mid2 = requant(relu(conv3x3/2(mid1) + b2)) on R x WO pixels -> mid2
    // =========================================================================================
    {
        v16i acc[CMW];
        const int oc = opix_ok ? opix : OUT_PX - 1;
        const int orow = oc / WO, ocol = oc - orow * WO;
        // patch entry of tap column ts for this lane's output pixel (row 2*orow): even phase ocol, odd phase, even phase ocol+1
        const int pe[3] = {2 * orow * PW + ocol, 2 * orow * PW + NE + ocol, 2 * orow * PW + ocol + 1};
#pragma unroll
        for (int i = 0; i < CMW; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4i bv = *(const v4i*)(bias_lds + MID + (wb * CMW + i) * 32 + 8 * g + 4 * lh);   // visible since P1's barriers
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][4 * g + e] = bv[e];
            }
        static_for<NK2>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            constexpr int BEFORE = J == 0 ? NS2 - 1 : ((J - 1 + NS2 < NK2) ? J - 1 + NS2 : NK2);
            constexpr int AFTER = (J + NS2 < NK2) ? J + NS2 : NK2;
            wait_vmcnt<(BEFORE - 1 - J) * WL0>();        // stage J landed; later stages may fly
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                // step 0: also "patch complete"
            static_for<AFTER - BEFORE>([&](auto ic) { constexpr int S = BEFORE + decltype(ic)::value; issue_w2(S, S % NS2); });
            constexpr int TAP = J / CH, TC = J % CH, TR = TAP / 3, TS = TAP % 3;
            const char* base = w2ring + (J % NS2) * W2B;
            const int ent = pe[TS] + TR * PW;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const v4i xf = *(const v4i*)(patch + SM::off(ent, TC * 4 + kk * 2 + lh));
#pragma unroll
                for (int i = 0; i < CMW; ++i) {
                    const v4i wf = *(const v4i*)(base + ((wb * CMW + i) * 32 + l31) * 64 + cof[kk]);
                    acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, xf, acc[i], 0, 0, 0);
                }
            }
        });
        F8_OT(3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // patch and W2 ring are dead everywhere: the P3 ring takes them over
        if constexpr (!P12) {
            issue_p3(0, 0);
            asm volatile("" ::: "memory");
            issue_p3(1, 1);
            asm volatile("" ::: "memory");
        }

        const int floor0 = a.relu_b ? 0 : INT32_MIN;
        const float sc2 = FQ == 1 ? requant_u8_scale(a.n2) : 0.0f;
        (void)floor0; (void)sc2;
#pragma unroll
        for (int i = 0; i < CMW; ++i) {
            const int ct = wb * CMW + i;
            unsigned d[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if constexpr (FQ != 0) {
                    d[g] = requant_u8x4_sel<FQ == 2 ? 2 : 1>(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3], a.n2, sc2) ^ 0x80808080u;
                } else {
                    int y[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = requant1(max(acc[i][4 * g + e], floor0), a.n2, a.lo2, a.hi2);
                    d[g] = pack4(y[0], y[1], y[2], y[3]) ^ a.xor2;
                }
            }
            auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
            v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
            if constexpr (P12) { if (opix_ok) *(v4i*)(a.q[0].ptr + (size_t)m * MID + ct * 32 + lh * 16) = o; }
            else *(v4i*)(mid2 + SM::off(opix, ct * 2 + lh)) = o;
        }
    }
    F8_OT(4);

    // =========================================================================================
    // P3: y = clamp(((Wsc . x + bsc) << sa) + ((W4 . mid2 + b4) << sr)) [ReLU] -> y32 (I32T) / int8 copies
    //     wave (wa, wb): px tile wa, co tile wb of each 64-channel chunk
    // =========================================================================================
    if constexpr (!P12) {
        const int floor1 = a.relu1 ? 0 : -2147483647 /* the join's clamp_(min=-(2^31-1)) and the ReLU floor are one max */;
        const bool stage0 = STG && a.q[0].ptr != nullptr;
        const int n_direct = (a.out32 ? 4 : 0) + ((!STG && a.q[0].ptr) ? 1 : 0) + (a.q[1].ptr ? 1 : 0);
        auto stores_of = [&](int x) { return x < 0 ? 0 : n_direct + ((stage0 && x >= 2 && !(x & 1)) ? 2 : 0); };
        auto flush_pair = [&](int c0) {                  // chunks c0, c0+1 -> 128-byte lines; exactly two store instructions per wave
            const char* buf = stgbuf((c0 >> 1) & 1);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int tq = tid + rr * 512;
                const int row = tq >> 3, c16 = tq & 7;
                const bool live = row < OUT_PX;
                const int rw = live ? row : OUT_PX - 1;  // dead rows: one lane re-writes the last row (uniform instruction count)
                if (live || lane == 0) {
                    const v4i v = *(const v4i*)(buf + rw * 128 + ((c16 ^ (rw & 7)) << 4));
                    *(v4i*)(a.q[0].ptr + (size_t)(m_tile + rw) * COUT + c0 * 64 + c16 * 16) = v;
                }
            }
        };
        v4i xf[KK3], xsf[KKS];                           // chunk-invariant operands: mid2 and shortcut-x fragments of this lane's pixel
        static_for<NC3>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            // chunk c's weights landed?  Newer in the queue (VMEM retires in order): stores of chunk c-2, the DMA of chunk
            // c+1, stores of chunk c-1 (chunk 0: only the DMA of chunk 1)
            wait_vmcnt_dyn(stores_of(c - 2) + (c + 1 < NC3 ? L3 : 0) + stores_of(c - 1));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                // chunk 0: also "mid2 / xs complete"; slot (c+2)%3 was read in chunk c-1
            if constexpr (c + 2 < NC3) issue_p3(c + 2, (c + 2) % 3);
            asm volatile("" ::: "memory");
            if constexpr (c >= 2 && !(c & 1)) { if (stage0) flush_pair(c - 2); }
            if constexpr (c == 0) {
#pragma unroll
                for (int kk = 0; kk < KK3; ++kk) xf[kk] = *(const v4i*)(mid2 + SM::off(opix, kk * 2 + lh));
#pragma unroll
                for (int kk = 0; kk < KKS; ++kk) xsf[kk] = *(const v4i*)(xs + SC::off(opix, kk * 2 + lh));
            }
            const char* base = p3slot(c % 3);
            const int cot = c * 64 + wb * 32;
            v16i acc, acs;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4i b4v = *(const v4i*)(bias_lds + 2 * MID + cot + 8 * g + 4 * lh);
                const v4i bsv = *(const v4i*)(bias_lds + 2 * MID + COUT + cot + 8 * g + 4 * lh);
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[4 * g + e] = b4v[e]; acs[4 * g + e] = bsv[e]; }
            }
#pragma unroll
            for (int kk = 0; kk < KK3; ++kk) {
                const v4i wf = *(const v4i*)(base + SM::off(wb * 32 + l31, kk * 2 + lh));
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, xf[kk], acc, 0, 0, 0);
            }
#pragma unroll
            for (int kk = 0; kk < KKS; ++kk) {
                const v4i wf = *(const v4i*)(base + 64 * MID + SC::off(wb * 32 + l31, kk * 2 + lh));
                acs = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, xsf[kk], acs, 0, 0, 0);
            }
            int y[4][4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned s = ((unsigned)acs[4 * g + e] << a.acc_shl) + ((unsigned)acc[4 * g + e] << a.res_shl);   // the shortcut conv hosts the join
                    y[g][e] = max((int)s, floor1);
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
                const v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
                if (STG && k == 0) {                     // 16 bytes of pixel opix, column (c & 1) * 64 + wb * 32 + lh * 16 of its 128-byte row
                    const int c16 = (c & 1) * 4 + wb * 2 + lh;
                    *(v4i*)(stgbuf((c >> 1) & 1) + opix * 128 + ((c16 ^ (opix & 7)) << 4)) = o;
                } else if (opix_ok) {
                    *(v4i*)(a.q[k].ptr + (size_t)m * COUT + cot + 16 * lh) = o;
                }
            }
        });
        if (stage0) {                                    // the last pair
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            flush_pair(NC3 - 2);
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

template <int C, int MID, int W, int R, int COUT, bool STG, bool P12 = false, int FQ = 0>
static hipError_t launch_opener_t(const FusedArgs& a, hipStream_t s) {
    using Cfg = OpenerCfg<C, MID, W, R, COUT>;
    static unsigned long long attr_done = 0; int attr_dev = -1;
    if (!dyn_lds_opted_in(&attr_done, &attr_dev)) {
        hipError_t e = hipFuncSetAttribute((const void*)fused_opener_kernel<C, MID, W, R, COUT, STG, P12, FQ>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        if (attr_dev >= 0) attr_done |= 1ull << attr_dev;
    }
    const int grid = a.N * a.tiles_per_img;
#ifdef F8_TRACE
    static unsigned long long* tbuf = nullptr; static int count = 0;
    static const int want = [] { const char* e = getenv("F8_TRACE_OPENER"); return e ? atoi(e) : -1; }();
    FusedArgs b = a;
    const bool tracing = (count++ == want);
    if (tracing) { if (!tbuf) (void)hipMalloc((void**)&tbuf, (size_t)1 << 22); (void)hipMemsetAsync(tbuf, 0, (size_t)grid * 64, s); b.trace = tbuf; }
    hipLaunchKernelGGL((fused_opener_kernel<C, MID, W, R, COUT, STG, P12, FQ>), dim3(grid), dim3(512), Cfg::LDS_BYTES, s, b);
    if (tracing) {
        (void)hipStreamSynchronize(s);
        unsigned long long* h = new unsigned long long[(size_t)grid * 8];
        (void)hipMemcpy(h, tbuf, (size_t)grid * 64, hipMemcpyDeviceToHost);
        double ph[5] = {0, 0, 0, 0, 0}; int n = 0;
        for (int i = 0; i < grid; ++i) { unsigned long long* p = h + (size_t)i * 8; if (!p[5]) continue; ++n; for (int k = 0; k < 5; ++k) ph[k] += (double)(p[k + 1] - p[k]); }
        fprintf(stderr, "[trace opener<%d,%d,%d,%d,%d>] grid %d: avg cycles per WG: P1 loop %.0f | P1 epi %.0f | P2 loop %.0f | P2 epi %.0f | P3 %.0f\n", C, MID, W, R, COUT, grid,
                ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n);
        delete[] h;
    }
    return hipGetLastError();
#else
    hipLaunchKernelGGL((fused_opener_kernel<C, MID, W, R, COUT, STG, P12, FQ>), dim3(grid), dim3(512), Cfg::LDS_BYTES, s, a);
    return hipGetLastError();
#endif
}

hipError_t launch_fused_opener(const FusedArgs& a, hipStream_t s) {
    const int stg = a.stg;
    if (a.C == 256 && a.MID == 128 && a.COUT == 512 && a.W == 56 && a.R == 4) {
        if (a.p12only) {
            if (!(a.q[0].ptr && !a.q[1].ptr && !a.out32)) return hipErrorInvalidValue;
            // FQ: ReLU -> unsigned 8-bit right shifts after both convs (1: float converter where it is provably exact, 2: integer form)
            const bool fqf = a.relu_a && a.relu_b && a.n1 > 0 && a.n2 > 0 && a.n1 <= 30 && a.n2 <= 30 && a.lo1 == 0 && a.lo2 == 0 && a.hi1 == 255 && a.hi2 == 255 &&
                             a.xor1 == 0x80808080u && a.xor2 == 0x80808080u;
#ifndef F8_OPENER_FQ
#define F8_OPENER_FQ 1             // 0 (tuning builds): the general epilogues
#endif
            const int fq = (!fqf || !F8_OPENER_FQ) ? 0 : ((a.rq_int || !a.acc_ok || a.n1 > kRequantU8MaxShift || a.n2 > kRequantU8MaxShift) ? 2 : 1);
            return fq == 1 ? launch_opener_t<256, 128, 56, 4, 512, false, true, 1>(a, s) : fq == 2 ? launch_opener_t<256, 128, 56, 4, 512, false, true, 2>(a, s)
                                                                                               : launch_opener_t<256, 128, 56, 4, 512, false, true, 0>(a, s);
        }
        return stg ? launch_opener_t<256, 128, 56, 4, 512, true>(a, s) : launch_opener_t<256, 128, 56, 4, 512, false>(a, s);
    }
    return hipErrorInvalidValue;
}

// stage-opening bottleneck with a stride-2 3x3 (1x1 -> 3x3/2 -> [1x1 + 1x1/2 shortcut]): the ResNet-50 stage-1 shape
bool fused_opener_supported(int C, int MID, int COUT, int H, int W, int* R) {
    if (C == 256 && MID == 128 && COUT == 512 && W == 56 && H % 8 == 0) { *R = 4; return true; }
    return false;
}

}  // namespace f8
