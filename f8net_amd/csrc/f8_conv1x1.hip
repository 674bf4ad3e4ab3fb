// f8_conv1x1.hip — 1x1 convolution (stride 1 / 2, no padding) / linear as an 8-wave block GEMM (gfx950 only).
//
// Same arithmetic contract as conv_igemm_kernel (f8_kernels.hip); this kernel serves the 1x1 layers with long K and
// few pixels (ResNet stages 2-3), where the 4-wave kernel spends its life in read-then-multiply K steps:
//   * 512 threads, tile 128 pixels x BN couts (BN = 64 / 128), K stage = 128 bytes per row: 24-32 KB per stage
//     through a 3-slot LDS-direct DMA ring (one barrier + counted wait per stage costs ~340 cycles whatever the
//     stage size, tools/ubench_ldsdma.hip: make stages big);
//   * each wave accumulates a 2x2 block of 32x32 tiles (1 KB of LDS reads per MFMA instead of 1.5-2), the waves that
//     share a block split the K slices of every stage, partial sums meet once through LDS (f8_block.h);
//   * the fragments of stage j+1 are read from LDS while the MFMAs of stage j run (register double buffer).
// DUAL: a second (x2, w2, bias2) product over the same output pixels runs after the first through the same ring and
// takes the place of the residual operand (bottleneck downsample join: body.4 + shortcut), as in conv_igemm_kernel.
#include "f8_device.h"
#include "f8_block.h"
#include <cstdlib>

namespace f8 {

template <int BN>
struct C1Cfg {
    static constexpr int BM = 128, KB = 128;
    static constexpr int X_BYTES = BM * KB, W_BYTES = BN * KB, STAGE = X_BYTES + W_BYTES;
    static constexpr int NS = 3;
    static constexpr int NCO = BN / 32, NB = 2 * (NCO / 2), KS = 8 / NB, NF = 4 / KS;
    static constexpr int RED_BYTES = 8 * (4 - NF) * 4096;
    static constexpr int LDS_BYTES = NS * STAGE > RED_BYTES ? NS * STAGE : RED_BYTES;
};

template <int BN, bool HAS_RES, bool DUAL>
__global__ void __launch_bounds__(512) conv1x1_block_kernel(const ConvArgs a) {
    using Cfg = C1Cfg<BN>;
    constexpr int BM = Cfg::BM, KB = Cfg::KB, X_BYTES = Cfg::X_BYTES, STAGE = Cfg::STAGE, NS = Cfg::NS;
    constexpr int NB = Cfg::NB, KS = Cfg::KS, NF = Cfg::NF;
    constexpr int CPR = KB / 16;                                // 16-byte chunks per row
    constexpr int XS = BM * CPR, WS = BN * CPR;                 // slots per stage
    constexpr int XL = XS / 512, WL = WS / 512;                 // DMA instructions per thread per stage (exact: BM, BN multiples of 64)
    constexpr int LD = XL + WL;
    constexpr int NSL = KB / 32, NQ = NSL / KS;                 // K slices per stage / per wave
    static_assert(!DUAL || HAS_RES, "the second product is the residual operand");
    static_assert(XS % 512 == 0 && WS % 512 == 0 && NQ >= 1, "shapes");
    using SK = Swz<KB>;

    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
    const int blk = wave % NB, ks = wave / NB;                  // 2x2 block, K-split index
    const int bp = blk & 1, bc = blk >> 1;                      // pixel-tile pair, cout-tile pair
    const int l31 = lane & 31, lh = lane >> 5;
    const int fpx = 2 * bp + (ks & 1);                          // tiles this wave finishes (f8_block.h)
    const int fco = 2 * bc + (KS == 4 ? (ks >> 1) : 0);

    // ---- tile: XCD-aware order, cout tile fastest (the workgroups re-reading one X tile sit on one XCD's L2)
    const int tilesN = (a.coutP + BN - 1) / BN;
    int wg;
    {
        const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
        wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    }
    const int tile_n = wg % tilesN, tile_m = wg / tilesN;
    const int m0 = tile_m * BM, co0 = tile_n * BN;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL ? a.x2 : a.x), 0, DUAL ? a.x2_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL ? a.w2 : a.w), 0, DUAL ? a.w2_bytes : 0u, 0x00020000);

    // ---- gather descriptors: thread t owns slots t + 512*i; row = slot / CPR, physical chunk = slot % CPR
    unsigned xb[XL], xb2[DUAL ? XL : 1], wb[WL], wb2[DUAL ? WL : 1];
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        const int sl = tid + i * 512;
        const int row = sl / CPR, chunk = (sl % CPR) ^ SK::f(row);
        const int m = m0 + row;
        xb[i] = kOOB;
        if (DUAL) xb2[i] = kOOB;
        if (m < a.M) {
            const int n = (int)fast_div((unsigned)m, a.mPQ, a.s1PQ, a.s2PQ), rem = m - n * a.PQ;
            const int p = (int)fast_div((unsigned)rem, a.mQ, a.s1Q, a.s2Q), q = rem - p * a.Q;
            xb[i] = (unsigned)(n * a.sN + p * a.sP + q * a.sQ + a.origin + chunk * 16);
            if (DUAL) xb2[i] = (unsigned)(n * a.sN2 + p * a.sP2 + q * a.sQ2 + chunk * 16);
        }
    }
#pragma unroll
    for (int i = 0; i < WL; ++i) {
        const int sl = tid + i * 512;
        const int row = sl / CPR, chunk = (sl % CPR) ^ SK::f(row);
        wb[i] = (unsigned)((co0 + row) * a.ktot + chunk * 16);  // rows past coutP fall outside the buffer: zeros
        if (DUAL) wb2[i] = (unsigned)((co0 + row) * a.ktot2 + chunk * 16);
    }

    const int nk1 = a.ktot / KB;
    const int nk = nk1 + (DUAL ? a.ktot2 / KB : 0);
    auto issue = [&](int j, int slot) {
        char* base = lds + slot * STAGE;
        if (DUAL && j >= nk1) {                                  // wave-uniform: stages of the second product
            const unsigned ko = (unsigned)((j - nk1) * KB);
#pragma unroll
            for (int i = 0; i < XL; ++i) {
                const unsigned off = xb2[i] == kOOB ? kOOB : xb2[i] + ko;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx2, (__attribute__((address_space(3))) void*)(base + i * 8192 + wave * 1024), 16, off, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < WL; ++i) {
                const unsigned off = wb2[i] + ko;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw2, (__attribute__((address_space(3))) void*)(base + X_BYTES + i * 8192 + wave * 1024), 16, off, 0, 0, 0);
            }
            return;
        }
        const unsigned ko = (unsigned)(j * KB);
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const unsigned off = xb[i] == kOOB ? kOOB : xb[i] + ko;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(base + i * 8192 + wave * 1024), 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WL; ++i) {
            const unsigned off = wb[i] + ko;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(base + X_BYTES + i * 8192 + wave * 1024), 16, off, 0, 0, 0);
        }
    };
#pragma unroll
    for (int st = 0; st < NS; ++st)
        if (st < nk) issue(st, st);
    asm volatile("" ::: "memory");

    // ---- this lane's output pixel (epilogue), residual operand and biases: queued behind the prologue DMA, which only
    //      makes the first counted waits conservative; consumed after the K loop
    const int m = m0 + fpx * 32 + l31;
    const bool pix_ok = m < a.M;
    v4i rv[HAS_RES ? NF : 1][4];
    if (HAS_RES && !DUAL) {
        const int mc = pix_ok ? m : m0;
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = co0 + (fco + i) * 32 + 8 * g + 4 * lh;
                v4i z = {0, 0, 0, 0};
                rv[i][g] = (c < a.coutP) ? *(const v4i*)(a.res + i32t_index(mc, c, a.coutP)) : z;
            }
    }
    v4i bq[NF][4], bq2[DUAL ? NF : 1][4];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = co0 + (fco + i) * 32 + 8 * g + 4 * lh;
            v4i z = {0, 0, 0, 0};
            bq[i][g] = (c < a.coutP) ? *(const v4i*)(a.bias + c) : z;
            if (DUAL) bq2[i][g] = (c < a.coutP) ? *(const v4i*)(a.bias2 + c) : z;
        }
    constexpr int EXTRA = (HAS_RES && !DUAL ? NF * 4 : 0) + NF * 4 * (DUAL ? 2 : 1);   // loads queued behind the prologue
    (void)EXTRA;

    v16i acc[2][2], acc2[DUAL ? 2 : 1][DUAL ? 2 : 1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0; if (DUAL) acc2[i][j][r] = 0; }

    const int wf_sw = SK::f(l31);                               // == f(row) for rows 32k + l31
    struct Frag { v4i xa, xb, w0, w1; };
    auto read_frags = [&](int j, Frag (&f)[NQ]) {
        const char* base = lds + (j % NS) * STAGE;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = ks + i * KS;                          // this wave's 32-byte K slice of the stage
            const unsigned ch = (unsigned)(((q * 2 + lh) ^ wf_sw) << 4);
            f[i].xa = *(const v4i*)(base + ((2 * bp) * 32 + l31) * KB + ch);
            f[i].xb = *(const v4i*)(base + ((2 * bp + 1) * 32 + l31) * KB + ch);
            f[i].w0 = *(const v4i*)(base + X_BYTES + ((2 * bc) * 32 + l31) * KB + ch);
            f[i].w1 = *(const v4i*)(base + X_BYTES + ((2 * bc + 1) * 32 + l31) * KB + ch);
        }
    };
    auto mma = [&](Frag (&f)[NQ], v16i (&ac)[2][2]) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            ac[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[i].w0, f[i].xa, ac[0][0], 0, 0, 0);
            ac[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[i].w0, f[i].xb, ac[0][1], 0, 0, 0);
            ac[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[i].w1, f[i].xa, ac[1][0], 0, 0, 0);
            ac[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[i].w1, f[i].xb, ac[1][1], 0, 0, 0);
        }
    };
    auto wait_n = [&](int n) {                                  // at most n newer VMEM operations in flight
        switch (n) {
            case 0: wait_vmcnt<0>(); break;  case 1: wait_vmcnt<1>(); break;  case 2: wait_vmcnt<2>(); break;
            case 3: wait_vmcnt<3>(); break;  case 4: wait_vmcnt<4>(); break;  case 5: wait_vmcnt<5>(); break;
            case 6: wait_vmcnt<6>(); break;  case 7: wait_vmcnt<7>(); break;  default: wait_vmcnt<8>(); break;
        }
    };
    auto stage = [&](int j, Frag (&cur)[NQ], Frag (&nxt)[NQ]) {
        // stage j+1 landed (its fragments are read below); up to NS-2 later stages stay in flight
        const int rem = nk - 2 - j;
        const int ahead = rem < 0 ? 0 : (rem < (NS - 2) ? rem : (NS - 2));
        wait_n(ahead * LD);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // `cur` is in registers; every wave is done with slot j % NS
        __builtin_amdgcn_s_barrier();
        if (j + NS < nk) issue(j + NS, j % NS);
        if (j + 1 < nk) read_frags(j + 1, nxt);
        if (DUAL && j >= nk1) { if constexpr (DUAL) mma(cur, acc2); }
        else mma(cur, acc);
    };
    Frag fa[NQ], fb[NQ];
    {   // prologue: stage 0 landed everywhere (everything issued after it may stay in flight), then its fragments
        const int later = ((nk < NS ? nk : NS) - 1) * LD + EXTRA;
        wait_n(later > 8 ? 8 : later);
        __builtin_amdgcn_s_barrier();
        read_frags(0, fa);
    }
    for (int j = 0; j < nk; j += 2) {
        stage(j, fa, fb);
        if (j + 1 < nk) stage(j + 1, fb, fa);
    }

    // ---- K-split exchange (the ring is dead now), then the fused epilogue on this wave's NF tiles
    v4i fin[NF][4];
    block_exchange<KS, NB, NF>(acc, (v4i*)lds, wave, blk, ks, lane, fin);
    if constexpr (DUAL) {
        v4i fin2[NF][4];
        block_exchange<KS, NB, NF>(acc2, (v4i*)lds, wave, blk, ks, lane, fin2);
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) rv[i][g][e] = (int)((unsigned)fin2[i][g][e] + (unsigned)bq2[i][g][e]);
    }
    block_finish<NF, HAS_RES>(a, fin, bq, rv, co0 + fco * 32, m, pix_ok, lh);
}

template <int BN>
static hipError_t launch_c1_t(const ConvArgs& a, hipStream_t s) {
    using Cfg = C1Cfg<BN>;
    static bool attr_set = false;
    if (!attr_set) {   // dynamic LDS above 64 KB must be opted into once per kernel
        hipError_t e = hipFuncSetAttribute((const void*)conv1x1_block_kernel<BN, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv1x1_block_kernel<BN, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv1x1_block_kernel<BN, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int grid = ((a.M + Cfg::BM - 1) / Cfg::BM) * ((a.coutP + BN - 1) / BN);
    if (a.x2) hipLaunchKernelGGL((conv1x1_block_kernel<BN, true, true>), dim3(grid), dim3(512), Cfg::LDS_BYTES, s, a);
    else if (a.res) hipLaunchKernelGGL((conv1x1_block_kernel<BN, true, false>), dim3(grid), dim3(512), Cfg::LDS_BYTES, s, a);
    else hipLaunchKernelGGL((conv1x1_block_kernel<BN, false, false>), dim3(grid), dim3(512), Cfg::LDS_BYTES, s, a);
    return hipGetLastError();
}

// Which 1x1 convs take this kernel: K (and K2) multiples of 128, at least 64 couts.  `M` = pixels of one launch.
// mode: 0 plain, 1 residual-carrying, 2 dual.  Returns the cout tile (64 / 128) or 0.
int conv1x1_block_config(int M, int coutP, int ktot, int ktot2, int mode) {
    static const int on = [] { const char* e = getenv("F8_BLOCK1X1"); return e ? atoi(e) : 0; }();   // bit per mode; OFF by default: measured slower than conv_igemm_kernel
    // on every ResNet-50 layer but the 14x14 body.0 convs (28.6 vs 30 us): these launches are bound by fixed latencies (launch, first load,
    // store drain) and by having ONE resident workgroup per CU (96 KB LDS), not by K-loop throughput; kept as a tuning experiment
    if (!((on >> mode) & 1) || coutP < 64 || ktot % 128 != 0 || ktot2 % 128 != 0) return 0;
    static const int force = [] { const char* e = getenv("F8_BLOCK1X1_BN"); return e ? atoi(e) : 0; }();
    if (force == 64 || force == 128) return force;
    const long tiles128 = (long)((M + 127) / 128) * ((coutP + 127) / 128);
    return (coutP >= 128 && tiles128 >= 192) ? 128 : 64;
}

hipError_t launch_conv1x1_block(const ConvArgs& a, int bn, hipStream_t s) {
    if (bn == 128) return launch_c1_t<128>(a, s);
    if (bn == 64) return launch_c1_t<64>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace f8
