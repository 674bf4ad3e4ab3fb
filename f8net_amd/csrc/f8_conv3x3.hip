// f8_conv3x3.hip — 3x3 / stride 1 / pad 1 convolution with the input patch resident in LDS (gfx950 only).
//
// Why a second conv kernel: a CU's load path (global -> LDS / VGPR) sustains ~14 B/clk, ~8.6 TB/s over the chip
// (profiles/, DESIGN.md "load path").  The implicit-GEMM kernel re-stages every activation row once per tap
// (9x) and every weight row once per pixel tile, and the late 3x3 layers of the ResNets (14x14x256, 7x7x512)
// run at exactly that limit.  Here a workgroup owns R full output rows of one image (or IMGS whole small images):
//   * the (R+2) x (W+2) x CIN input patch is fetched ONCE by LDS-direct DMA (out-of-image pixels come back
//     as zeros through the buffer range check; the border-class bias table repairs them, f8_net.cpp),
//   * a tap is a constant LDS offset, so the K loop streams only the BN x 64-byte weight tiles through a ring,
//   * 8 waves = 4 pixel tiles x 2 cout groups; weights = MFMA A operand, patch pixels = B operand.
// Staged bytes per layer drop 2-2.4x against the implicit GEMM at these shapes.
//
// Arithmetic is that of conv_igemm_kernel: wrapping int32 accumulate, class bias, ReLU, optional residual join,
// int32 (I32T) and / or requantised int8 outputs (reference: models/fix_quant_ops.py:99-112, fix_resnet.py:40-54).
#include "f8_device.h"
#include "f8_block.h"
#include <cstdlib>
#ifdef F8_TRACE
#include <cstdio>
#endif

namespace f8 {

template <int CIN, int W, int R, int IMGS, int BN, int KB>
struct PatchCfg {
    static constexpr int PW = W + 2, PR = R + 2;
    static constexpr int IMG_PX = PR * PW;                      // patch pixels per image
    static constexpr int PATCH_PX = IMGS * IMG_PX;
    static constexpr int OUT_PX = IMGS * R * W;
    static constexpr int NPO = (OUT_PX + 31) / 32;
    static constexpr int CPR = CIN / 16;
    static constexpr int PSLOTS = PATCH_PX * CPR;               // 16-byte slots of the patch
    static constexpr int PL = (PSLOTS + 511) / 512;             // DMA instructions per thread for the patch
    static constexpr int PATCH_BYTES = PL * 8192;               // whole wave pieces (tail slots are written as zeros)
    static constexpr int W_BYTES = BN * KB;                     // one weight stage: KB bytes of K for BN couts
    static constexpr int NS = 3;                                // weight ring depth (tools/ubench_ldsdma: 3 == 4 > 8)
    // K-split exchange buffer (reuses the patch / ring space): every wave parks the partial sums of the tiles it
    // does not finish itself: 8 waves x (4 - 4/KS) tiles x 4 KB, KS = 8 / (BN / 32)
    static constexpr int RED_BYTES = 8 * (4 - 4 / (8 / (2 * (BN / 64)))) * 4096;
    static constexpr int LDS_BYTES = PATCH_BYTES + NS * W_BYTES > RED_BYTES ? PATCH_BYTES + NS * W_BYTES : RED_BYTES;
    static constexpr int NK = 9 * CIN / KB;
    static_assert((9 * CIN) % KB == 0 && (KB == 64 || KB == 128 || KB == 256), "stage width");
    static constexpr int CMW = BN / 64;                         // cout tiles per wave
    static_assert(NPO == 4, "4 pixel-tile groups");
    static_assert(CIN % 64 == 0 && (BN == 64 || BN == 128), "shapes");
    static_assert((NPO - 1) * 32 < OUT_PX, "every pixel tile has live lanes");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <int CIN, int W, int R, int IMGS, int BN, int KB, bool HAS_RES>
__global__ void __launch_bounds__(512) conv3x3_patch_kernel(const ConvArgs a) {
    using Cfg = PatchCfg<CIN, W, R, IMGS, BN, KB>;
    constexpr int PW = Cfg::PW, IMG_PX = Cfg::IMG_PX, OUT_PX = Cfg::OUT_PX, CPR = Cfg::CPR;
    constexpr int PSLOTS = Cfg::PSLOTS, PL = Cfg::PL, PATCH_BYTES = Cfg::PATCH_BYTES, W_BYTES = Cfg::W_BYTES;
    constexpr int NS = Cfg::NS, NK = Cfg::NK;
    constexpr int WS = BN * KB / 16;                            // 16-byte slots of a weight stage
    constexpr int WL = (WS + 511) / 512, WCPR = KB / 16;
    // Wave roles.  The workgroup tile is 4 pixel tiles x NCO cout tiles of 32x32.  A wave accumulates a 2x2 BLOCK
    // of them (two x fragments + two w fragments from LDS feed four MFMAs: LDS read bandwidth, 128 B/clk per CU,
    // is what bounds one-tile-per-wave schemes), and the KS waves that share a block split the K slices of every
    // stage between them; their partial sums meet in LDS after the loop (integer adds: order-free, exact).
    constexpr int NCO = BN / 32, NB = 2 * (NCO / 2), KS = 8 / NB, NSL = KB / 32, NF = 4 / KS;
    static_assert(NCO == 2 || NCO == 4, "cout tiles");
    using SK = Swz<KB>;
    using SC = Swz<CIN>;

    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const patch = lds;
    char* const ring = lds + PATCH_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
    const int blk = wave % NB, ks = wave / NB;                  // 2x2 block, K-split index
    const int bp = blk & 1, bc = blk >> 1;                      // pixel-tile pair, cout-tile pair
    const int l31 = lane & 31, lh = lane >> 5;
    // tiles this wave finishes in the epilogue: pixel tile fpx, cout tiles fco .. fco + NF - 1
    const int fpx = 2 * bp + (ks & 1);
    const int fco = 2 * bc + (KS == 4 ? (ks >> 1) : 0);

    // ---- tile: XCD-aware order, cout tile fastest (the workgroups sharing a patch sit on one XCD's L2)
    const int tilesN = (a.coutP + BN - 1) / BN;
    int wg;
    {
        const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
        wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    }
    const int tile_n = wg % tilesN, t = wg / tilesN;
    const int co0 = tile_n * BN;
    const int NIMG = a.M / a.PQ;
    const int tiles_per_img = a.H / R;                          // IMGS > 1: R == H, one tile = IMGS whole images
    const int n0 = (IMGS > 1) ? t * IMGS : t / tiles_per_img;
    const int p0 = (IMGS > 1) ? 0 : (t - n0 * tiles_per_img) * R;
    const int m_tile = (n0 * a.H + p0) * W;                     // global output pixel of this tile's pixel 0
    const int live_px = (IMGS > 1) ? ((NIMG - n0) < IMGS ? (NIMG - n0) : IMGS) * R * W : R * W;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);

#ifdef F8_TRACE
    unsigned long long tt[6]; tt[0] = __builtin_readcyclecounter();
#define F8_TT(i) tt[i] = __builtin_readcyclecounter()
#else
#define F8_TT(i)
#endif
    // patch pixel of tap (0,0) for output pixel `op` of the tile (padding lanes: a valid pixel, result unused)
    auto patch_px = [&](int op, int* row, int* col) {
        const int oc = op < OUT_PX ? op : OUT_PX - 1;
        const int oimg = oc / (R * W), orem = oc - oimg * (R * W);
        const int orow = orem / W, ocol = orem - orow * W;
        *row = orow; *col = ocol;
        return oimg * IMG_PX + orow * PW + ocol;
    };
    int dr, dc;
    const int bpxA = patch_px((2 * bp) * 32 + l31, &dr, &dc);   // main loop: pixel tiles 2bp, 2bp + 1
    const int bpxB = patch_px((2 * bp + 1) * 32 + l31, &dr, &dc);
    // epilogue pixel of this lane
    const int opix = fpx * 32 + l31;
    const bool opix_ok = opix < live_px;
    const int m = m_tile + opix;
    int orow, ocol;
    (void)patch_px(opix, &orow, &ocol);

    // border class of this lane's pixel: two dependent byte loads, requested first and consumed after the DMA issue
    int cls_r = 0, cls_c = 0;
    if (a.ncc > 0) { cls_r = a.rowcls[p0 + orow]; cls_c = a.colcls[ocol]; }   // IMGS > 1: p0 == 0

    // ---- patch: every 16-byte slot once
#pragma unroll
    for (int i = 0; i < PL; ++i) {
        const int s = tid + i * 512;
        const int ppx = s / CPR, phys = s - ppx * CPR;
        const int img = ppx / IMG_PX, rem = ppx - img * IMG_PX;
        const int pr = rem / PW, pc = rem - pr * PW;
        const int h = p0 - 1 + pr, w = pc - 1, n = n0 + img;
        const bool ok = s < PSLOTS && h >= 0 && h < a.H && w >= 0 && w < W && n < NIMG;
        const unsigned off = ok ? (unsigned)(((n * a.H + h) * W + w) * CIN + ((phys ^ SC::f(ppx)) << 4)) : kOOB;
        if ((i * 512 + wave * 64) < PSLOTS)                     // wave-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(patch + i * 8192 + wave * 1024), 16, off, 0, 0, 0);
    }

    // ---- weight ring: stage j = bytes [j*KB, j*KB + KB) of BN cout rows (K is tap-major, then channel)
    unsigned wbs[WL];
#pragma unroll
    for (int i = 0; i < WL; ++i) {
        const int sl = tid + i * 512;
        const int row = sl / WCPR, chunk = (sl % WCPR) ^ SK::f(row);
        wbs[i] = (sl < WS) ? (unsigned)((co0 + row) * a.ktot + chunk * 16) : kOOB;   // rows past coutP: outside the buffer -> zeros
    }
    int ldw = 0;                                                // weight-stage DMA instructions of THIS wave
#pragma unroll
    for (int i = 0; i < WL; ++i) ldw += ((i * 512 + wave * 64) < WS) ? 1 : 0;
    auto issue_w = [&](int j, int slot) {
#pragma unroll
        for (int i = 0; i < WL; ++i) {
            const unsigned woff = wbs[i] + (unsigned)(j * KB);
            if ((i * 512 + wave * 64) < WS)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(ring + slot * W_BYTES + i * 8192 + wave * 1024), 16, woff, 0, 0, 0);
        }
    };
#pragma unroll
    for (int st = 0; st < NS; ++st)
        if (st < NK) issue_w(st, st);
    asm volatile("" ::: "memory");

    // ---- residual operand and class biases: behind the prologue DMA in the (in-order) VMEM queue, which only
    //      makes the first counted waits conservative; they are consumed after the K loop
    v4i rv[HAS_RES ? NF : 1][4];
    if (HAS_RES) {
        const int mc = opix_ok ? m : m_tile;
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = co0 + (fco + i) * 32 + 8 * g + 4 * lh;
                v4i z = {0, 0, 0, 0};
                rv[i][g] = (c < a.coutP) ? *(const v4i*)(a.res + i32t_index(mc, c, a.coutP)) : z;
            }
    }
    v4i bq[NF][4];
    {
        const int32_t* bias = a.bias;
        bias += (size_t)(cls_r * a.ncc + cls_c) * (size_t)a.coutP;      // ncc == 0: single class
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = co0 + (fco + i) * 32 + 8 * g + 4 * lh;
                v4i z = {0, 0, 0, 0};
                bq[i][g] = (c < a.coutP) ? *(const v4i*)(bias + c) : z;
            }
    }

    F8_TT(1);
    v16i acc[2][2];                                             // [cout tile of the pair][pixel tile of the pair]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    constexpr int CH = CIN / 64;                                // 64-byte channel pieces per tap
    constexpr int SUB = KB / 64;                                // 64-byte K pieces per stage
    constexpr int NQ = (NSL / KS) > 0 ? (NSL / KS) : 1;         // K slices of a stage this wave owns
    const int wf_sw = SK::f(l31);                               // == f(row) for both rows (row = 32k + l31, 32 % 16 == 0)
    struct Frag { v4i xa, xb, w0, w1; };
    // The fragments of stage j+1 are fetched from LDS WHILE the MFMAs of stage j run (register double buffer):
    // with one barrier per stage, read-then-multiply inside a stage would serialise the LDS latency, the LDS
    // transfer (1 KB per MFMA) and the matrix pipe on every step.
    auto read_frags = [&](int j, Frag (&f)[NQ]) {
        const char* base = ring + (j % NS) * W_BYTES;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = ks + i * KS;                          // this wave's 32-byte K slice of the stage
            if (q < NSL) {                                      // wave-uniform (only the CIN = 64 shape has idle waves)
                const int u = q >> 1, kk = q & 1;
                const int piece = j * SUB + u;                  // 64-byte K piece: tap = piece / CH, channel piece = piece % CH
                const int tap = piece / CH, tc = piece - tap * CH;
                const int tr = (tap * 11) >> 5, ts = tap - tr * 3;        // tap / 3 for tap < 9
                const int xch = tc * 4 + kk * 2 + lh;
                f[i].xa = *(const v4i*)(patch + SC::off(bpxA + tr * PW + ts, xch));
                f[i].xb = *(const v4i*)(patch + SC::off(bpxB + tr * PW + ts, xch));
                const unsigned wch = (unsigned)(((u * 4 + kk * 2 + lh) ^ wf_sw) << 4);
                f[i].w0 = *(const v4i*)(base + ((2 * bc) * 32 + l31) * KB + wch);
                f[i].w1 = *(const v4i*)(base + ((2 * bc + 1) * 32 + l31) * KB + wch);
            }
        }
    };
    auto stage = [&](int j, Frag (&cur)[NQ], Frag (&nxt)[NQ]) {
        // stage j+1 landed (its fragments are read below); up to NS-2 later stages stay in flight
        const int rem = NK - 2 - j;                             // stages after j+1
        const int ahead = rem < 0 ? 0 : (rem < (NS - 2) ? rem : (NS - 2));
        const int nfl = ahead * ldw;
        if (nfl == 0) wait_vmcnt<0>();
        else if (nfl == 1) wait_vmcnt<1>();
        else wait_vmcnt<2>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // `cur` is in registers; every wave is done with slot j % NS
        __builtin_amdgcn_s_barrier();
        if (j + NS < NK) issue_w(j + NS, j % NS);
        if (j + 1 < NK) read_frags(j + 1, nxt);
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            if (ks + i * KS < NSL) {
                acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur[i].w0, cur[i].xa, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur[i].w0, cur[i].xb, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur[i].w1, cur[i].xa, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur[i].w1, cur[i].xb, acc[1][1], 0, 0, 0);
            }
        }
    };
    Frag fa[NQ], fb[NQ];
    // prologue: patch + stage 0 landed everywhere, then this wave's stage-0 fragments
    {
        const int inflight = ((NK < NS ? NK : NS) - 1) * ldw;
        if (inflight == 0) wait_vmcnt<0>();
        else if (inflight == 1) wait_vmcnt<1>();
        else if (inflight == 2) wait_vmcnt<2>();
        else if (inflight == 3) wait_vmcnt<3>();
        else wait_vmcnt<4>();
        __builtin_amdgcn_s_barrier();
#ifdef F8_TRACE
        tt[2] = __builtin_readcyclecounter();
#endif
        read_frags(0, fa);
    }
    for (int j = 0; j < NK; j += 2) {
        stage(j, fa, fb);
        if (j + 1 < NK) stage(j + 1, fb, fa);
    }
    F8_TT(3);

    // ---- K-split exchange through LDS (patch / ring are dead now), then the fused epilogue on this wave's NF tiles
    v4i fin[NF][4];
    block_exchange<KS, NB, NF>(acc, (v4i*)lds, wave, blk, ks, lane, fin);
    block_finish<NF, HAS_RES>(a, fin, bq, rv, co0 + fco * 32, m, opix_ok, lh);
#ifdef F8_TRACE
    if (a.trace && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tt[4] = __builtin_readcyclecounter();
        unsigned long long* tp = (unsigned long long*)a.trace + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 5; ++i) tp[i] = tt[i];
    }
#endif
}

template <int CIN, int W, int R, int IMGS, int BN, int KB>
static hipError_t launch_patch_t(const ConvArgs& a, hipStream_t s) {
    using Cfg = PatchCfg<CIN, W, R, IMGS, BN, KB>;
    static unsigned long long attr_done = 0; int attr_dev = -1;
    if (!dyn_lds_opted_in(&attr_done, &attr_dev)) {   // dynamic LDS above 64 KB must be opted into once per kernel
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_patch_kernel<CIN, W, R, IMGS, BN, KB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_patch_kernel<CIN, W, R, IMGS, BN, KB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        if (attr_dev >= 0) attr_done |= 1ull << attr_dev;
    }
    const int nimg = a.M / a.PQ;
    const int tiles = (IMGS > 1) ? (nimg + IMGS - 1) / IMGS : nimg * (a.H / R);
    const int grid = tiles * ((a.coutP + BN - 1) / BN);
#ifdef F8_TRACE
    static unsigned long long* tbuf = nullptr; static int count = 0;
    static const int want = [] { const char* e = getenv("F8_TRACE_PATCH"); return e ? atoi(e) : -1; }();
    ConvArgs b = a;
    const bool tracing = (count++ == want);
    if (tracing) { if (!tbuf) (void)hipMalloc((void**)&tbuf, (size_t)1 << 22); (void)hipMemsetAsync(tbuf, 0, (size_t)grid * 64, s); b.trace = tbuf; }
    if (a.res) hipLaunchKernelGGL((conv3x3_patch_kernel<CIN, W, R, IMGS, BN, KB, true>), dim3(grid), dim3(512), Cfg::LDS_BYTES, s, b);
    else hipLaunchKernelGGL((conv3x3_patch_kernel<CIN, W, R, IMGS, BN, KB, false>), dim3(grid), dim3(512), Cfg::LDS_BYTES, s, b);
    if (tracing) {
        (void)hipStreamSynchronize(s);
        unsigned long long* h = new unsigned long long[(size_t)grid * 8];
        (void)hipMemcpy(h, tbuf, (size_t)grid * 64, hipMemcpyDeviceToHost);
        double ph[4] = {0, 0, 0, 0}; int n = 0; unsigned long long lo = ~0ull, hi = 0;
        for (int i = 0; i < grid; ++i) { unsigned long long* p = h + (size_t)i * 8; if (!p[4]) continue; ++n; for (int k = 0; k < 4; ++k) ph[k] += (double)(p[k + 1] - p[k]);
            if (p[0] < lo) lo = p[0]; if (p[4] > hi) hi = p[4]; }
        fprintf(stderr, "[trace patch<%d,%d,%d,%d,%d,%d>] grid %d span %llu: avg cycles per WG: issue %.0f | first wait %.0f | loop %.0f | epilogue %.0f\n", CIN, W, R, IMGS, BN, KB, grid,
                hi - lo, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n);
        delete[] h;
    }
    return hipGetLastError();
#else
    if (a.res) hipLaunchKernelGGL((conv3x3_patch_kernel<CIN, W, R, IMGS, BN, KB, true>), dim3(grid), dim3(512), Cfg::LDS_BYTES, s, a);
    else hipLaunchKernelGGL((conv3x3_patch_kernel<CIN, W, R, IMGS, BN, KB, false>), dim3(grid), dim3(512), Cfg::LDS_BYTES, s, a);
    return hipGetLastError();
#endif
}

// Shapes with an instance: (cin, input width) -> rows per tile / images per tile / cout tile.
bool conv3x3_patch_config(int cin, int H, int W, int coutP, int* R, int* IMGS, int* BN) {
    if (coutP < 64) return false;
    if (cin == 128 && W == 28 && H % 4 == 0) { *R = 4; *IMGS = 1; *BN = 128; return true; }
    if (cin == 256 && W == 14 && H % 7 == 0) { *R = 7; *IMGS = 1; *BN = 128; return true; }
#ifndef F8_P3_BN7
#define F8_P3_BN7 64              // cout tile of the 7x7x512 instance (64: two workgroups share an image pair's patch; tuning builds)
#endif
    if (cin == 512 && W == 7 && H == 7) { *R = 7; *IMGS = 2; *BN = F8_P3_BN7; return true; }
    return false;
}

hipError_t launch_conv3x3_patch(const ConvArgs& a, int cin, hipStream_t s) {
    if (cin == 128 && a.W == 28) return launch_patch_t<128, 28, 4, 1, 128, 128>(a, s);
    if (cin == 256 && a.W == 14) return launch_patch_t<256, 14, 7, 1, 128, 128>(a, s);
    if (cin == 512 && a.W == 7) return launch_patch_t<512, 7, 7, 2, F8_P3_BN7, F8_P3_BN7 == 64 ? 256 : 128>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace f8
