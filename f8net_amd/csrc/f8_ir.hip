// f8_ir.hip — one launch for a MobileNet-V2 inverted-residual block (gfx950).
//
//   x8 (int8 NHWC, CIN ch) --1x1 CIN->E, ReLU--> e1 --depthwise 3x3 / s, pad 1, ReLU--> e2 --1x1 E->COUT--> [+ x (int32)] -> y
//
// i.e. IntBlock.forward of /root/reference/models/fix_mobilenet_v2.py:20-48 for the blocks built by :168-176 (expand_ratio != 1),
// every int_op_only_fix_quant (fix_quant_ops.py:90-114) in place.  Unfused the block is three launches and the 6x-expanded
// tensors e1 / e2 go through HBM twice (at 112x112x96 that is 154 MB per 128 images, written and read); fused, HBM sees the
// block input once and the block output once, and the late blocks (14x14, 7x7), which are launch-latency bound, become one launch.
//
// The expanded dimension E is processed in CHUNKS of 64 channels: depthwise convolution does not mix channels and the project
// GEMM sums over them, so per chunk
//   P1  expand: e1[chunk] for the tile's input rows (MFMA, K = CIN)          -> requant -> LDS patch (zero border = biased zero)
//   P2  depthwise 3x3 on the patch (v_dot4 over 4 taps of one channel)       -> requant -> LDS mid2
//   P3  project: acc[out px][COUT] += W4[:, chunk] . mid2 (MFMA, K = 64)     accumulators stay in registers across chunks
// and only the chunk's slices of the three weight sets are resident (double-buffered: the next chunk's slices are loaded into
// registers while this chunk computes).  Two barriers per chunk.  Epilogue: bias, [align + int32 residual + clamp], int32 (I32T)
// and / or requantised int8 copies.
//
// Work unit: R output rows x full width of one image (R * Wo <= 128; 256 for the <32, 32> instance), or G whole images when a map has
// <= 64 pixels.  4 or 8 waves (NW below); a wave owns one output pixel tile in P3 / the epilogue and every PXW-th pixel tile in P1.  All
// shapes are run-time values (the kernel is instantiated per (CIN, COUT) channel pair only): the 64x64 test nets run the same code.
#include "f8_device.h"

namespace f8 {

// P2MMA: the depthwise phase on the matrix cores (instances whose project accumulators leave 52 registers for it)
// FQ: both inner requantisations are right shifts into UNSIGNED 8-bit behind a ReLU (every block of MobileNet-V2): the ReLU is the clamp's
// lower bound (requant is monotone and maps 0 to 0), the bias rides in the accumulators' start value, the shift (1 .. 16) is requant_u8x4: 3 vector
// operations per expanded value, packing included, instead of 9 + packing — and the expanded values are what this kernel is bound by (VALU, not memory)
// NW: waves per workgroup.  8 for the <32, 32> instance (stages 1 - 2 of MobileNet-V2: 122 registers, so two workgroups = 16 waves fit a CU):
// the launch is bound by vector work between barriers, and at 4 waves x 2 workgroups a SIMD had two waves to hide them with; its 8 waves
// carry 8 pixel tiles (256 output pixels: more output rows per tile, fewer expand rows computed twice).
// SPLIT (the other 8-wave instances, 14x14 maps: 256 workgroups on 256 CUs, so a 4-wave workgroup leaves ONE wave per SIMD and every phase
// runs at the latency of its own dependency chain): waves 4 - 7 take the second half of the channels of the same four pixel tiles — P1's
// second 32 expanded channels, P3's and the epilogue's upper output-channel tiles; P2 already iterates (channel tile, pixel tile) pairs
#ifndef F8_IR_SPLIT
#define F8_IR_SPLIT 1
#endif
constexpr int ir_nw(int cinS, int coutS) { return (cinS == 32 && coutS == 32) || (F8_IR_SPLIT && coutS <= 96) ? 8 : 4; }
template <int CIN_S, int COUT_S, int FQ, bool P2MMA = (COUT_S <= 96), int NW = ir_nw(CIN_S, COUT_S)>
__global__ void __launch_bounds__(NW * 64, (COUT_S <= 64 ? 4 : COUT_S <= 96 ? 3 : COUT_S <= 160 ? 2 : 1)) fused_ir_kernel(const IRArgs a) {
    constexpr int NT = NW * 64;
    constexpr bool SPLIT = NW == 8 && !(CIN_S == 32 && COUT_S == 32);
    constexpr int PXW = SPLIT ? 4 : NW;                    // waves that own a pixel tile of their own
    constexpr int MID2_CT = PXW * 1024;                    // bytes of one 32-channel plane of mid2: PXW pixel tiles x 32 px x 32 B
    constexpr int KK1 = CIN_S / 32, NCO = COUT_S / 32;
    constexpr int JSPLIT = (NCO + 1) / 2;                  // SPLIT: output-channel tiles [0, JSPLIT) on waves 0 - 3, the rest on waves 4 - 7
    constexpr int W0_BYTES = 64 * CIN_S, W4_BYTES = COUT_S * 64;
    constexpr int W0_SLOTS = W0_BYTES / 16, W4_SLOTS = W4_BYTES / 16, SM_SLOTS = 36 + 16 + 16;   // dw weights (576 B), dw bias, expand bias
    constexpr int W0_L = (W0_SLOTS + NT - 1) / NT, W4_L = (W4_SLOTS + NT - 1) / NT;
    if constexpr (FQ == 1) set_fp_round_nearest_even();
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const X = lds;                                   // [KK1][xp][32 B]
    char* const patch = lds + a.off_patch;                 // [2][G][PR][PW][32 B]: channel-tile planes, so that a wave's fragment read (one
                                                           // 32-channel tile of 32 pixels) is 1 KB contiguous — at [px][64 B] it used half of every bank row
    char* const mid2 = lds + a.off_mid2;                   // [2][NW * 32][32 B]
    char* const wbuf = lds + a.off_w;                      // 2 x { W0 [KK1][64][32] | W4 [2][COUT_S][32] | dw 576 B (+64 pad) | dw bias 256 B | b0 256 B }
    constexpr int OFF_W4 = W0_BYTES, OFF_DW = OFF_W4 + W4_BYTES, OFF_DWB = OFF_DW + 640, OFF_B0 = OFF_DWB + 256, WBUF = OFF_B0 + 256;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & (NW - 1);
    const int l31 = lane & 31, lh = lane >> 5;
    const int pw = SPLIT ? wave & 3 : wave, ch = SPLIT ? wave >> 2 : 0;
    auto my_j = [&](int j) { return !SPLIT || (ch == 0 ? j < JSPLIT : j >= JSPLIT); };
    const int s = a.stride, R = a.R, W = a.W, H = a.H, Wo = a.Wo, PW = W + 2;
    const int PR = (R - 1) * s + 3;
    const int pct = a.G * PR * PW * 32;                    // bytes of one 32-channel plane of the patch
    int t;
    {   // XCD-aware order: vertically adjacent row tiles share their halo rows in one XCD's L2
        const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
        t = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    }
    int n0, p0;
    if (a.G > 1) { n0 = t * a.G; p0 = 0; } else { n0 = t / a.tiles_per_img; p0 = (t - n0 * a.tiles_per_img) * R; }
    const int Geff = (a.N - n0) < a.G ? (a.N - n0) : a.G;
    const int in_row0 = p0 * s - 1;
    const int vr0 = in_row0 < 0 ? 0 : in_row0, vr1 = (in_row0 + PR) > H ? H : (in_row0 + PR), nvr = vr1 - vr0;
    const int P1_PX = Geff * nvr * W, np1 = (P1_PX + 31) >> 5;
    const int RWo = R * Wo, OUT_PX = Geff * RWo;
    const int nchunk = (a.E32 + 63) >> 6;
    // pixel index -> (image g, row, column) without hardware division (host magic numbers).  Row tiles (G == 1) have g == 0;
    // whole-image tiles (G > 1) have nvr == H.
    auto split_in = [&](int px, int& g, int& vr, int& c) {
        g = a.G > 1 ? (int)fast_div((unsigned)px, a.mHW, a.s1HW, a.s2HW) : 0;
        const int r = px - g * nvr * W;
        vr = (int)fast_div((unsigned)r, a.mW, a.s1W, a.s2W);
        c = r - vr * W;
    };
    auto split_out = [&](int op, int& g, int& orow, int& ocol) {
        g = a.G > 1 ? (int)fast_div((unsigned)op, a.mRWo, a.s1RWo, a.s2RWo) : 0;
        const int r = op - g * RWo;
        orow = (int)fast_div((unsigned)r, a.mWo, a.s1Wo, a.s2Wo);
        ocol = r - orow * Wo;
    };

    // ---- block input tile -> X (k-blocked: [kk][px][32 B], so a fragment read is 1 KB contiguous per wave)
    {
        const int nslot = a.xp * KK1 * 2;
        for (int sl = tid; sl < nslot; sl += NT) {
            const int kk = KK1 == 1 ? 0 : sl / (a.xp * 2), rem = sl - kk * (a.xp * 2), px = rem >> 1, half = rem & 1;
            v4i v = {0, 0, 0, 0};
            if (px < P1_PX) {
                int g, vr, c;
                split_in(px, g, vr, c);
                const size_t gpx = ((size_t)(n0 + g) * H + vr0 + vr) * W + c;
                v = *(const v4i*)(a.x8 + gpx * CIN_S + kk * 32 + half * 16);
            }
            *(v4i*)(X + (size_t)sl * 16) = v;
        }
    }
    // ---- patch <- biased zero (border columns, rows outside the image; P1 only ever writes interior pixels)
    {
        const v4i zv = {(int)a.xor1, (int)a.xor1, (int)a.xor1, (int)a.xor1};
        const int pb = a.G * PR * PW * 64;
        for (int o = tid * 16; o < pb; o += NT * 16) *(v4i*)(patch + o) = zv;
    }

    // ---- weight slices of one chunk: global -> registers (early) -> LDS (late)
    v4i rw0[W0_L], rw4[W4_L], rsm;
    auto load_w = [&](int e) {
        const int rows_ok = a.E32 - 64 * e;                // expanded channels left from this chunk on (>= 32)
#pragma unroll
        for (int i = 0; i < W0_L; ++i) {                   // W0 rows 64e .. 64e+63 -> [kk][row][32 B]
            const int sl = tid + i * NT;
            const int kk = sl / 128, row = (sl >> 1) & 63, half = sl & 1;
            v4i v = {0, 0, 0, 0};
            if (sl < W0_SLOTS && row < rows_ok) v = *(const v4i*)(a.w0 + (size_t)(64 * e + row) * CIN_S + kk * 32 + half * 16);
            rw0[i] = v;
        }
#pragma unroll
        for (int i = 0; i < W4_L; ++i) {                   // W4 columns 64e .. 64e+63 of every row -> [kk][row][32 B]
            const int sl = tid + i * NT;
            const int kk = sl / (COUT_S * 2), row = (sl >> 1) % COUT_S, half = sl & 1;
            v4i v = {0, 0, 0, 0};
            if (sl < W4_SLOTS && kk * 32 < rows_ok) v = *(const v4i*)(a.w4 + (size_t)row * a.E32 + 64 * e + kk * 32 + half * 16);
            rw4[i] = v;
        }
        {   // depthwise weights (dot4 image: 36 B per 4-channel quad), depthwise bias, expand bias: 64 channels each
            v4i v = {0, 0, 0, 0};
            if (tid < 36) { if (tid * 16 + 16 <= (rows_ok >= 64 ? 576 : 288)) v = *(const v4i*)(a.wd4 + (size_t)(16 * e) * 36 + tid * 16); }
            else if (tid < 52) { const int i = tid - 36; if (4 * i < rows_ok) v = *(const v4i*)(a.bd4 + 64 * e + 4 * i); }
            else if (tid < SM_SLOTS) { const int i = tid - 52; if (4 * i < rows_ok) v = *(const v4i*)(a.b0 + 64 * e + 4 * i); }
            rsm = v;
        }
    };
    auto store_w = [&](int buf) {
        char* wb = wbuf + buf * WBUF;
#pragma unroll
        for (int i = 0; i < W0_L; ++i) { const int sl = tid + i * NT; if (sl < W0_SLOTS) *(v4i*)(wb + sl * 16) = rw0[i]; }
#pragma unroll
        for (int i = 0; i < W4_L; ++i) { const int sl = tid + i * NT; if (sl < W4_SLOTS) *(v4i*)(wb + OFF_W4 + sl * 16) = rw4[i]; }
        if (tid < 36) *(v4i*)(wb + OFF_DW + tid * 16) = rsm;
        else if (tid < 52) *(v4i*)(wb + OFF_DWB + (tid - 36) * 16) = rsm;
        else if (tid < SM_SLOTS) *(v4i*)(wb + OFF_B0 + (tid - 52) * 16) = rsm;
    };
    load_w(0);
    store_w(0);

    v16i acc3[NCO];
#pragma unroll
    for (int j = 0; j < NCO; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[j][r] = 0;
    const int floor_a = a.relu_a ? 0 : INT32_MIN, floor_b = a.relu_b ? 0 : INT32_MIN;
    const float sc1 = FQ == 1 ? requant_u8_scale(a.n1) : 0.0f, sc2 = FQ == 1 ? requant_u8_scale(a.n2) : 0.0f;   // FQ == 2: integer requantisation (f8_device.h)
    (void)floor_a; (void)sc1; (void)sc2;
    const unsigned padv = a.xor1;

    for (int e = 0; e < nchunk; ++e) {
        const char* wb = wbuf + (e & 1) * WBUF;
        const int nct = (a.E32 - 64 * e) >= 64 ? 2 : 1;    // 32-channel tiles in this chunk (the last chunk may be half)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // this chunk's weight slices are in LDS (chunk 0: also X and the patch border);
                                                           // every wave is done with the previous chunk's P3 (mid2) and P2 (patch)
        if (e + 1 < nchunk) load_w(e + 1);                 // in flight during P1 .. P3
        // ================= P1: expand -> patch
        for (int pt = pw; pt < np1; pt += PXW) {
            v16i acc[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    v4i bv = {0, 0, 0, 0};
                    if constexpr (FQ) bv = *(const v4i*)(wb + OFF_B0 + (i * 32 + 8 * gq + 4 * lh) * 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[i][4 * gq + q] = bv[q];
                }
#pragma unroll
            for (int kk = 0; kk < KK1; ++kk) {
                const v4i xf = *(const v4i*)(X + ((size_t)kk * a.xp + pt * 32 + l31) * 32 + lh * 16);
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    if (i < nct && (!SPLIT || i == ch)) {
                        const v4i wf = *(const v4i*)(wb + (kk * 64 + i * 32 + l31) * 32 + lh * 16);
                        acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, xf, acc[i], 0, 0, 0);
                    }
            }
            const int px = pt * 32 + l31;
            const bool ok = px < P1_PX;
            const int pxc = ok ? px : 0;
            int g, vr, c;
            split_in(pxc, g, vr, c);
            const int ent = (g * PR + (vr0 + vr - in_row0)) * PW + c + 1;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i >= nct || (SPLIT && i != ch)) continue;
                unsigned d[4];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    int y[4];
                    if constexpr (FQ) {
                        d[gq] = requant_u8x4_sel<FQ == 2 ? 2 : 1>(acc[i][4 * gq], acc[i][4 * gq + 1], acc[i][4 * gq + 2], acc[i][4 * gq + 3], a.n1, sc1) ^ 0x80808080u;
                    } else {
                        const v4i bv = *(const v4i*)(wb + OFF_B0 + (i * 32 + 8 * gq + 4 * lh) * 4);
#pragma unroll
                        for (int q = 0; q < 4; ++q) y[q] = requant1(max((int)((unsigned)acc[i][4 * gq + q] + (unsigned)bv[q]), floor_a), a.n1, a.lo1, a.hi1);
                        d[gq] = pack4(y[0], y[1], y[2], y[3]) ^ a.xor1;
                    }
                }
                auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
                if (ok) {
                    const v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
                    *(v4i*)(patch + (size_t)i * pct + (size_t)ent * 32 + lh * 16) = o;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // patch complete
        // ================= P2: depthwise 3x3 on the patch -> mid2
        if constexpr (P2MMA) {
            // ... ON THE MATRIX CORES: for a 32-channel tile the 3x3 depthwise conv is nine v_mfma_i32_32x32x32_i8 with a DIAGONAL
            // weight fragment (A[c][k] = w[tap][c] for k == c, else 0) — the B operand is the patch pixel as it lies in LDS (16 channels
            // per lane half), nothing is byte-transposed, the accumulators start at the bias and come out in the layout the epilogue of
            // every conv in the library turns into a 16-byte row (permlane32 swap).  The VALU version below spends 8 v_perm_b32 + 4 v_dot4
            // per quad and tap group on transposing 4 taps x 4 channels: ~9 vector operations per output against 9 MFMAs per 1024.
            const int npo = (OUT_PX + 31) >> 5;
            for (int pr = wave; pr < npo * nct; pr += NW) {           // (channel tile, pixel tile) pairs over the waves
                const int ctd = pr / npo, pt = pr - ctd * npo;
                v4i wa[9];                                       // the nine diagonal fragments of this 32-channel tile
                {
                    const int cch = ctd * 32 + l31;              // channel (of the chunk) this lane's A row belongs to
                    const unsigned* wq = (const unsigned*)(wb + OFF_DW + (cch >> 2) * 36);      // [wA0..3, wB0..3, wC]: dot4 image
                    const unsigned dA = wq[cch & 3], dB = wq[4 + (cch & 3)], dC = wq[8];
                    const bool mine = (l31 >> 4) == lh;          // K index == row index: rows 0-15 live in K half 0, 16-31 in half 1
                    const int dsel = (l31 & 15) >> 2, bsh = 8 * (l31 & 3);
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const unsigned wv = t < 4 ? (dA >> (8 * t)) & 0xffu : t < 8 ? (dB >> (8 * (t - 4))) & 0xffu : (dC >> (8 * (cch & 3))) & 0xffu;
                        const int piece = mine ? (int)(wv << bsh) : 0;
                        wa[t] = v4i{dsel == 0 ? piece : 0, dsel == 1 ? piece : 0, dsel == 2 ? piece : 0, dsel == 3 ? piece : 0};
                    }
                }
                {
                    const int op = pt * 32 + l31;
                    const bool ok2 = op < OUT_PX;
                    int g, orow, ocol;
                    split_out(ok2 ? op : 0, g, orow, ocol);
                    const char* pp = patch + (size_t)ctd * pct + ((size_t)((g * PR + orow * s) * PW + ocol * s)) * 32 + lh * 16;
                    v16i acc2;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const v4i bv = *(const v4i*)(wb + OFF_DWB + (ctd * 32 + 8 * gq + 4 * lh) * 4);
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc2[4 * gq + q] = bv[q];
                    }
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const v4i xf = *(const v4i*)(pp + ((t / 3) * PW + t % 3) * 32);
                        acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(wa[t], xf, acc2, 0, 0, 0);
                    }
                    unsigned d[4];
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        if constexpr (FQ)
                            d[gq] = requant_u8x4_sel<FQ == 2 ? 2 : 1>(acc2[4 * gq], acc2[4 * gq + 1], acc2[4 * gq + 2], acc2[4 * gq + 3], a.n2, sc2) ^ 0x80808080u;
                        else
                            d[gq] = pack4(requant1(max(acc2[4 * gq], floor_b), a.n2, a.lo2, a.hi2), requant1(max(acc2[4 * gq + 1], floor_b), a.n2, a.lo2, a.hi2),
                                          requant1(max(acc2[4 * gq + 2], floor_b), a.n2, a.lo2, a.hi2), requant1(max(acc2[4 * gq + 3], floor_b), a.n2, a.lo2, a.hi2)) ^ a.xor2;
                    }
                    auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
                    auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
                    if (ok2) {
                        const v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
                        *(v4i*)(mid2 + ctd * MID2_CT + op * 32 + lh * 16) = o;
                    }
                }
            }
        } else {
        //                  (VALU: one item = one output pixel x 16 channels)
        for (int it = tid; it < OUT_PX * 4; it += NT) {
            const int op = it >> 2, cg = it & 3;
            if (cg >= nct * 2) continue;
            int g, orow, ocol;
            split_out(op, g, orow, ocol);
            const char* pp = patch + (size_t)(cg >> 1) * pct + ((size_t)((g * PR + orow * s) * PW + ocol * s)) * 32 + (cg & 1) * 16;
            v4i xw[3][3];
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) xw[rr][cc] = *(const v4i*)(pp + (rr * PW + cc) * 32);
            unsigned outw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {                  // 4-channel quad inside the 16
                const unsigned* wq = (const unsigned*)(wb + OFF_DW + (cg * 4 + k) * 36);       // [wA0..3, wB0..3, wC]
                const v4i bv = *(const v4i*)(wb + OFF_DWB + (cg * 4 + k) * 16);
                unsigned tp[9];
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) tp[rr * 3 + cc] = (unsigned)xw[rr][cc][k];
                int ac[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int grp = 0; grp < 2; ++grp) {        // taps 0-3, 4-7: byte-transpose 4 taps x 4 channels, then one dot4 per channel
                    const unsigned t0 = tp[grp * 4], t1 = tp[grp * 4 + 1], t2 = tp[grp * 4 + 2], t3 = tp[grp * 4 + 3];
                    const unsigned lo01 = __builtin_amdgcn_perm(t1, t0, 0x05010400u), hi01 = __builtin_amdgcn_perm(t1, t0, 0x07030602u);
                    const unsigned lo23 = __builtin_amdgcn_perm(t3, t2, 0x05010400u), hi23 = __builtin_amdgcn_perm(t3, t2, 0x07030602u);
                    const unsigned c0 = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u), c1 = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);
                    const unsigned c2 = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u), c3 = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u);
                    ac[0] = __builtin_amdgcn_sdot4((int)c0, (int)wq[grp * 4 + 0], ac[0], false);
                    ac[1] = __builtin_amdgcn_sdot4((int)c1, (int)wq[grp * 4 + 1], ac[1], false);
                    ac[2] = __builtin_amdgcn_sdot4((int)c2, (int)wq[grp * 4 + 2], ac[2], false);
                    ac[3] = __builtin_amdgcn_sdot4((int)c3, (int)wq[grp * 4 + 3], ac[3], false);
                }
                const unsigned wC = wq[8];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    ac[q] = (int)((unsigned)ac[q] + (unsigned)((int)(signed char)(tp[8] >> (8 * q)) * (int)(signed char)(wC >> (8 * q))));
                outw[k] = pack4(requant1(max(ac[0], floor_b), a.n2, a.lo2, a.hi2), requant1(max(ac[1], floor_b), a.n2, a.lo2, a.hi2),
                                requant1(max(ac[2], floor_b), a.n2, a.lo2, a.hi2), requant1(max(ac[3], floor_b), a.n2, a.lo2, a.hi2)) ^ a.xor2;
            }
            const v4i o = {(int)outw[0], (int)outw[1], (int)outw[2], (int)outw[3]};
            *(v4i*)(mid2 + (cg >> 1) * MID2_CT + op * 32 + (cg & 1) * 16) = o;
        }
        }
        (void)padv;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // mid2 complete
        // ================= P3: project, accumulate over the chunks (wave = output pixel tile)
        if (pw * 32 < OUT_PX) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if (kk >= nct) continue;
                const v4i xf = *(const v4i*)(mid2 + kk * MID2_CT + (pw * 32 + l31) * 32 + lh * 16);
#pragma unroll
                for (int j = 0; j < NCO; ++j) {
                    if (!my_j(j)) continue;
                    const v4i wf = *(const v4i*)(wb + OFF_W4 + ((kk * COUT_S) + j * 32 + l31) * 32 + lh * 16);
                    acc3[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, xf, acc3[j], 0, 0, 0);
                }
            }
        }
        if (e + 1 < nchunk) store_w((e + 1) & 1);          // the other buffer was last read in the previous chunk (barriers above)
    }

    // ================= epilogue: bias, [align + int32 residual + clamp], int32 (I32T) and / or int8 copies
    const int opx = pw * 32 + l31;
    if (pw * 32 >= OUT_PX) return;
    const bool ok = opx < OUT_PX;
    const int oc = ok ? opx : 0;
    int g, orow, ocol;
    split_out(oc, g, orow, ocol);
    const int m = ((n0 + g) * a.Ho + p0 + orow) * Wo + ocol;
    const int floor0 = a.relu0 ? 0 : INT32_MIN, floor1 = a.relu1 ? 0 : -2147483647;
#pragma unroll
    for (int j = 0; j < NCO; ++j) {
        if (!my_j(j)) continue;
        const int cot = j * 32;
        int y[4][4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const v4i bv = *(const v4i*)(a.b4 + cot + 8 * gq + 4 * lh);
            v4i rv = {0, 0, 0, 0};
            if (a.xr && ok) rv = *(const v4i*)(a.xr + i32t_index(m, cot + 8 * gq + 4 * lh, COUT_S));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int v = max((int)((unsigned)acc3[j][4 * gq + q] + (unsigned)bv[q]), floor0);
                if (a.xr) {
                    const unsigned sres = ((unsigned)v << a.acc_shl) + ((unsigned)rv[q] << a.res_shl);
                    v = max((int)sres, floor1);
                }
                y[gq][q] = v;
            }
        }
        if (a.out32 && ok) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const v4i o = {y[gq][0], y[gq][1], y[gq][2], y[gq][3]};
                *(v4i*)(a.out32 + i32t_index(m, cot + 8 * gq + 4 * lh, COUT_S)) = o;
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!a.q[k].ptr) continue;
            unsigned d[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                d[gq] = pack4(requant1(y[gq][0], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(y[gq][1], a.q[k].n, a.q[k].lo, a.q[k].hi),
                              requant1(y[gq][2], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(y[gq][3], a.q[k].n, a.q[k].lo, a.q[k].hi)) ^ a.q[k].bias_xor;
            auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
            if (ok) {
                const v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
                *(v4i*)(a.q[k].ptr + (size_t)m * COUT_S + cot + 16 * lh) = o;
            }
        }
    }
}

static bool ir_instance(int cinS, int coutS) {
    static const int pairs[][2] = {{32, 32}, {32, 64}, {64, 64}, {64, 96}, {96, 96}, {96, 160}, {160, 160}, {160, 320}};
    for (auto& p : pairs) if (p[0] == cinS && p[1] == coutS) return true;
    return false;
}

// output pixels of a tile: one 32-pixel tile per wave in P3 (8 waves for the <32, 32> instance, else 4)
static int ir_max_px(int cinS, int coutS) { return cinS == 32 && coutS == 32 ? 256 : 128; }

// LDS layout of a tile (bytes); false if it does not fit
static bool ir_layout(int cinS, int coutS, int H, int W, int stride, int R, int G, IRArgs* a, int* lds_bytes) {
    const int PR = (R - 1) * stride + 3, PW = W + 2;
    const int rows = PR < H ? PR : H;                       // valid input rows of a tile are at most this many
    const int xp = (G * rows * W + 31) / 32 * 32;
    const int x_bytes = xp * cinS;
    const int patch = (G * PR * PW * 64 + 255) / 256 * 256;
    const int wbuf = 64 * cinS + coutS * 64 + 640 + 256 + 256;
    const int mid2 = 2 * ir_max_px(cinS, coutS) * 32;
    const int total = x_bytes + patch + mid2 + 2 * wbuf;
    if (a) { a->xp = xp; a->off_patch = x_bytes; a->off_mid2 = x_bytes + patch; a->off_w = x_bytes + patch + mid2; }
    if (lds_bytes) *lds_bytes = total;
    return total <= 160 * 1024;
}

// Tile choice: R output rows x full width with R * Wo <= 128 (R divides Ho), or G whole images for small maps.
bool fused_ir_config(int cinS, int coutS, int H, int W, int stride, int* R, int* G) {
    if (!ir_instance(cinS, coutS) || (stride != 1 && stride != 2) || H < 1 || W < 1) return false;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const int cap = ir_max_px(cinS, coutS);
    if (Wo > cap) return false;
    int r = cap / Wo;
    if (r > Ho) r = Ho;
    while (r > 1 && Ho % r != 0) --r;
    int g = 1;
    if (r == Ho) { g = 128 / (Ho * Wo); if (g < 1) g = 1; if (g > 8) g = 8; }
    // the 8-wave instance wants two workgroups per CU (16 waves at occupancy 4): prefer tiles of at most 80 KB
    int lds = 0;
    while (!ir_layout(cinS, coutS, H, W, stride, r, g, nullptr, &lds) || (cap > 128 && lds > 80 * 1024 && g == 1 && r > 1)) {
        if (g > 1) { --g; continue; }
        if (r <= 1) return false;
        --r;
        while (r > 1 && Ho % r != 0) --r;
    }
    *R = r; *G = g;
    return true;
}

template <int CIN_S, int COUT_S, int FQ>
static hipError_t launch_ir_t(const IRArgs& a, int lds, hipStream_t s) {
    // dynamic LDS above 64 KB must be opted into per kernel AND per device (a process may drive several GPUs): keep the maximum per device
    static int attr_lds[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (dev < 0 || lds > attr_lds[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)fused_ir_kernel<CIN_S, COUT_S, FQ>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        if (dev >= 0) attr_lds[dev] = lds;
    }
    const int grid = a.G > 1 ? (a.N + a.G - 1) / a.G : a.N * a.tiles_per_img;
    hipLaunchKernelGGL((fused_ir_kernel<CIN_S, COUT_S, FQ>), dim3(grid), dim3(ir_nw(CIN_S, COUT_S) * 64), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_fused_ir(const IRArgs& a0, int cinS, int coutS, hipStream_t s) {
    IRArgs a = a0;
    int lds = 0;
    if (!ir_layout(cinS, coutS, a.H, a.W, a.stride, a.R, a.G, &a, &lds)) return hipErrorInvalidValue;
    // FQ: ReLU + right shift into unsigned 8-bit after the expand AND the depthwise conv (the VALU depthwise path keeps its general epilogue)
    // 1: through the float converter (bounded accumulators, shifts <= 16); 2: the integer form (option requant_float = 0, or where 1 is not provably exact)
    const bool fqf = a.relu_a && a.relu_b && a.n1 > 0 && a.n2 > 0 && a.n1 <= 30 && a.n2 <= 30 && a.lo1 == 0 && a.lo2 == 0 && a.hi1 == 255 && a.hi2 == 255 &&
                     a.xor1 == 0x80808080u && a.xor2 == 0x80808080u && coutS <= 96;
    const int fq = !fqf ? 0 : ((a.rq_int || !a.acc_ok || a.n1 > kRequantU8MaxShift || a.n2 > kRequantU8MaxShift) ? 2 : 1);
#define F8_IR(C_, O_) if (cinS == C_ && coutS == O_) return fq == 1 ? launch_ir_t<C_, O_, 1>(a, lds, s) : fq == 2 ? launch_ir_t<C_, O_, 2>(a, lds, s) : launch_ir_t<C_, O_, 0>(a, lds, s);
    F8_IR(32, 32) F8_IR(32, 64) F8_IR(64, 64) F8_IR(64, 96) F8_IR(96, 96) F8_IR(96, 160) F8_IR(160, 160) F8_IR(160, 320)
#undef F8_IR
    return hipErrorInvalidValue;
}

}  // namespace f8
