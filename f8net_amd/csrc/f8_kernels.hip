// f8_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the fixed-point-8 integer forward.
//
// Written for CDNA4 only: 64-wide wavefronts, v_mfma_i32_32x32x32_i8, 160 KB LDS, buffer loads
// with hardware range checking, v_permlane32_swap.  No portability layer.
//
// Arithmetic contract (bit-exact with the reference's int32 CPU path):
//   conv / linear : wrapping int32 accumulate of int8 x int8 products + int32 bias
//   requant       : /root/reference/models/fix_quant_ops.py:99-112 (shift, round-half-even, clamp)
//   residual      : /root/reference/models/fix_resnet.py:40-54 (align shift, wrapping add, clamp)
// Unsigned (0..255) activations meet a signed-only MFMA through the offset identity
//   sum w*x = sum w*(x-128) + 128*sum w     (x-128 == x ^ 0x80 as int8)
// Unsigned int8 tensors are STORED biased (x ^ 0x80) by their producers, so operands go from HBM to
// LDS untouched; 128*sum(w) over the taps that lie inside the image is folded into a per-border-class
// bias table on the host (f8_net.cpp: pack_conv_weights).  Everything is mod 2^32, so the identity
// is exact under wrap-around.
#include "f8_device.h"
#include <cstdlib>
#ifdef F8_TRACE
#include <algorithm>
#include <cstdio>
#include <vector>
#endif

namespace f8 {

// Fused epilogue of one BM x BN tile: class bias -> ReLU -> [align + residual + clamp -> ReLU] -> int32 (I32T)
// and / or up to two requantised int8 (NHWC) outputs.  ReLUs are branch-free floors.
template <int BM, int BN, int WPX, int WCO, bool HAS_RES, int TCO, int TPX>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, v16i (&acc)[TCO][TPX], v4i (&rv)[HAS_RES ? TCO : 1][HAS_RES ? TPX : 1][4],
                                              int m0, int co0, int wpx, int wco, int l31, int lh) {
    const int floor0 = a.relu0 ? 0 : INT32_MIN, floor1 = a.relu1 ? 0 : -2147483647 /* the join's clamp_(min=-(2^31-1)) and the ReLU floor are one max */;
#pragma unroll
    for (int j = 0; j < TPX; ++j) {
        const int m = m0 + wpx * (BM / WPX) + j * 32 + l31;
        const bool ok = m < a.M;
        const size_t rowo = (size_t)m * (size_t)a.coutP;
        // border class of this lane's pixel (which taps fell outside the image)
        const int32_t* bias = a.bias;
        if (a.ncc > 0 && ok) {
            const int n = (int)fast_div((unsigned)m, a.mPQ, a.s1PQ, a.s2PQ), rem = m - n * a.PQ;
            const int p = (int)fast_div((unsigned)rem, a.mQ, a.s1Q, a.s2Q), q = rem - p * a.Q;
            bias += (size_t)((int)a.rowcls[p] * a.ncc + (int)a.colcls[q]) * (size_t)a.coutP;
        }
#pragma unroll
        for (int i = 0; i < TCO; ++i) {
            const int cot = co0 + wco * (BN / WCO) + i * 32;   // first cout of this 32-wide MFMA tile
            if (cot >= a.coutP) continue;                      // wave-uniform
            int y[4][4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4i bv = *(const v4i*)(bias + cot + 8 * g + 4 * lh);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int v = max((int)((unsigned)acc[i][j][4 * g + e] + (unsigned)bv[e]), floor0);
                    if (HAS_RES) {
                        const unsigned s = ((unsigned)v << a.acc_shl) + ((unsigned)rv[i][j][g][e] << a.res_shl);
                        v = max((int)s, floor1);
                    }
                    y[g][e] = v;
                }
            }
            if (a.out32 && (m - l31 < a.M)) {              // I32T: 4 x 1 KB contiguous per wave
                int32_t* op = a.out32 + i32t_index(m, cot, a.coutP) + 4 * 32 * lh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    v4i o = {y[g][0], y[g][1], y[g][2], y[g][3]};
                    *(v4i*)(op + g * 256) = o;
                }
            }
            // int8 rows: lanes l and l+32 hold interleaved 4-channel groups of one pixel
            //   lower: d[0]=c0-3  d[1]=c8-11  d[2]=c16-19 d[3]=c24-27
            //   upper: d[0]=c4-7  d[1]=c12-15 d[2]=c20-23 d[3]=c28-31
            // two half-swaps give each lane 16 contiguous channel bytes (lower c0-15, upper c16-31).
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!a.q[k].ptr) continue;                 // wave-uniform
                unsigned d[4];
                const int qn = a.q[k].n, qlo = a.q[k].lo, qhi = a.q[k].hi;
                if (qn > 0) {                              // wave-uniform: the common case, right shift
                    const unsigned half = 1u << (qn - 1), mask = (half << 1) - 1u;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        d[g] = pack4(requant_shr(y[g][0], qn, half, mask, qlo, qhi), requant_shr(y[g][1], qn, half, mask, qlo, qhi),
                                     requant_shr(y[g][2], qn, half, mask, qlo, qhi), requant_shr(y[g][3], qn, half, mask, qlo, qhi)) ^ a.q[k].bias_xor;
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        d[g] = pack4(requant_shl(y[g][0], qn, qlo, qhi), requant_shl(y[g][1], qn, qlo, qhi),
                                     requant_shl(y[g][2], qn, qlo, qhi), requant_shl(y[g][3], qn, qlo, qhi)) ^ a.q[k].bias_xor;
                }
                auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
                if (ok) {
                    v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
                    *(v4i*)(a.q[k].ptr + rowo + cot + 16 * lh) = o;
                }
            }
        }
    }
}

// DUAL: a second 1x1 conv (x2, w2, bias2; its own strides, no padding) is accumulated after the first in the
// same DMA ring and takes the place of the residual operand:  out = join(acc + bias, acc2 + bias2).  This is the
// downsample block's `body.4 + shortcut` pair: the int32 tensor between them never exists in HBM.
// (Minimum occupancy for the 128 x 128 tile: left alone the compiler parks the 64 accumulator registers in AGPRs AND keeps ~94 VGPRs —
// 158 registers, 2 - 3 waves per SIMD; bounded it needs 94 - 162 without scratch.  Same finding as f8_ir.hip, round 3.)
template <int BM, int BN, int BK, int WPX, int WCO, bool HAS_PAD, bool HAS_RES, int STAGES, bool DUAL = false>
__global__ void __launch_bounds__(256, BM * BN >= 128 * 128 ? (DUAL ? 2 : HAS_RES ? 3 : 4) : 1) conv_igemm_kernel(const ConvArgs a) {
    static_assert(WPX * WCO == 4, "4 waves");
    static_assert(!DUAL || (HAS_RES && !HAS_PAD), "dual GEMM: the second product is the residual operand; 1x1 convs only");
    static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
    constexpr int CPR = BK / 16;                  // chunks per row
    constexpr int RPB = 256 / BK;                 // rows per 256-byte bank row
    constexpr int XCH = BM * CPR, WCH = BN * CPR; // 16-byte slots per tile
    constexpr int XL = (XCH + 255) / 256, WL = (WCH + 255) / 256;
    constexpr int NLD = XL + WL;                  // DMA instructions per thread per stage (upper bound)
    constexpr int TPX = BM / WPX / 32, TCO = BN / WCO / 32;
    constexpr int KK = BK / 32;
    constexpr int XBYTES = BM * BK, TILE = (BM + BN) * BK;
    static_assert(TPX >= 1 && TCO >= 1, "wave tile");
    static_assert(STAGES * TILE <= 65536, "static LDS");

    __shared__ __attribute__((aligned(16))) char lds[STAGES * TILE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 3;
    const int wpx = wave / WCO, wco = wave % WCO;

    // XCD-aware tile order: consecutive tiles (cout-tile fastest, then pixel-tile) stay on one XCD,
    // so a pixel tile's X rows and 3x3 halos are re-read from that XCD's L2.
    const int tilesN = (a.coutP + BN - 1) / BN;
    const int nwg = gridDim.x;
    int wg;
    {
        const int bid = blockIdx.x, xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
        wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    }
    const int tile_n = wg % tilesN, tile_m = wg / tilesN;
    const int m0 = tile_m * BM, co0 = tile_n * BN;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);

    const int l31 = lane & 31, lh = lane >> 5;

    // ---- residual operand: ALL of the wave tile's int32 rows are requested before anything else, so
    // their HBM latency overlaps the operand staging and the MFMAs (the in-place out32 store of the
    // same addresses comes later from the same lane).  I32T layout: 1 KB contiguous per access.
#ifdef F8_TRACE
    unsigned long long t_start = __builtin_readcyclecounter(), t_pro = 0, t_first = 0, t_loop = 0;
#endif
    v4i rv[HAS_RES ? TCO : 1][HAS_RES ? TPX : 1][4];
    if (HAS_RES && !DUAL) {
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int j = 0; j < TPX; ++j) {
                const int cot = co0 + wco * (BN / WCO) + i * 32;
                const int m = m0 + wpx * (BM / WPX) + j * 32 + l31;
                // whole 32-pixel tiles exist in the (row-padded) I32T buffer: guard per tile, not per lane
                const bool ld = (m - l31 < a.M) && (cot < a.coutP);
                const int32_t* rp = a.res + i32t_index(m, cot, a.coutP) + 4 * 32 * lh;   // + g*256
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    v4i z = {0, 0, 0, 0};
                    rv[i][j][g] = ld ? *(const v4i*)(rp + g * 256) : z;
                }
            }
    }

    // ---- per-thread gather descriptors (K-loop invariant).  Thread t owns LDS slots t + 256*i:
    // row = slot / CPR, physical chunk = slot % CPR, which holds LOGICAL chunk (phys ^ f(row)).
    unsigned xbase[XL], wbase[WL];
    int xh0[XL], xw0[XL];   // top-left input coordinate of each gathered row (HAS_PAD only)
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        const int idx = tid + i * 256;
        const int row = idx / CPR, chunk = (idx % CPR) ^ ((row / RPB) % CPR);
        const int m = m0 + row;
        xh0[i] = xw0[i] = -(1 << 24);
        xbase[i] = kOOB;
        if (idx < XCH && m < a.M) {
            const int n = (int)fast_div((unsigned)m, a.mPQ, a.s1PQ, a.s2PQ), rem = m - n * a.PQ;
            const int p = (int)fast_div((unsigned)rem, a.mQ, a.s1Q, a.s2Q), q = rem - p * a.Q;
            xbase[i] = (unsigned)(n * a.sN + p * a.sP + q * a.sQ + a.origin + chunk * 16);
            xh0[i] = p * a.stride - a.pad;
            xw0[i] = q * a.stride - a.pad;
        }
    }
#pragma unroll
    for (int j = 0; j < WL; ++j) {
        const int idx = tid + j * 256;
        const int row = idx / CPR, chunk = (idx % CPR) ^ ((row / RPB) % CPR);
        // rows past coutP fall outside the buffer and read as 0
        wbase[j] = (idx < WCH) ? (unsigned)((co0 + row) * a.ktot + chunk * 16) : kOOB;
    }

    // second operand pair (DUAL): same tile, its own image strides
    const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL ? a.x2 : a.x), 0, DUAL ? a.x2_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL ? a.w2 : a.w), 0, DUAL ? a.w2_bytes : 0u, 0x00020000);
    unsigned xbase2[DUAL ? XL : 1], wbase2[DUAL ? WL : 1];
    if (DUAL) {
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / CPR, chunk = (idx % CPR) ^ ((row / RPB) % CPR);
            const int m = m0 + row;
            xbase2[i] = kOOB;
            if (idx < XCH && m < a.M) {
                const int n = (int)fast_div((unsigned)m, a.mPQ, a.s1PQ, a.s2PQ), rem = m - n * a.PQ;
                const int p = (int)fast_div((unsigned)rem, a.mQ, a.s1Q, a.s2Q), q = rem - p * a.Q;
                xbase2[i] = (unsigned)(n * a.sN2 + p * a.sP2 + q * a.sQ2 + chunk * 16);
            }
        }
#pragma unroll
        for (int j = 0; j < WL; ++j) {
            const int idx = tid + j * 256;
            const int row = idx / CPR, chunk = (idx % CPR) ^ ((row / RPB) % CPR);
            wbase2[j] = (idx < WCH) ? (unsigned)((co0 + row) * a.ktot2 + chunk * 16) : kOOB;
        }
    }

    // ---- per-lane fragment addresses
    const int fl = (l31 / RPB) % CPR;
    unsigned coff[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) coff[kk] = (unsigned)(((kk * 2 + lh) ^ fl) << 4);
    const unsigned xfrag0 = (unsigned)((wpx * (BM / WPX) + l31) * BK);
    const unsigned wfrag0 = (unsigned)(XBYTES + (wco * (BN / WCO) + l31) * BK);

    v16i acc[TCO][TPX];
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int j = 0; j < TPX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    const int nk1 = a.ktot / BK;
    const int nk = nk1 + (DUAL ? a.ktot2 / BK : 0);
    // K-step state of the NEXT stage to issue (wave-uniform): tap row/col, channel offset in the tap
    int tr = 0, ts = 0, c0 = 0, kiss = 0;

    // Every thread issues exactly NLD DMA instructions per stage (waves without a slot for some
    // instruction issue it anyway with an out-of-range offset into a scratch-free no-op? no: they skip
    // it, and the counted waits below use the per-wave instruction count).
    auto issue_stage = [&](int slot) {
        char* base = lds + slot * TILE;
        const unsigned koffx = (unsigned)(tr * a.tapH + ts * a.tapW + c0);
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            unsigned off = xbase[i] + koffx;
            if (HAS_PAD)   // out-of-image tap: fetched through the range check (DMA writes 0)
                off = ((unsigned)(xh0[i] + tr) < (unsigned)a.H && (unsigned)(xw0[i] + ts) < (unsigned)a.W) ? off : kOOB;
            if ((i * 256 + wave * 64) < XCH)      // wave-uniform
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(base + i * 4096 + wave * 1024),
                                                         16, off, 0, 0, 0);
        }
        const unsigned koffw = (unsigned)(kiss * BK);
#pragma unroll
        for (int j = 0; j < WL; ++j) {
            const unsigned woff = wbase[j] + koffw;   // (a captured-array element passed directly to the builtin
                                                      //  makes hipcc 7.2 drop the kernel's host stub: keep it a scalar)
            if ((j * 256 + wave * 64) < WCH)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(base + XBYTES + j * 4096 + wave * 1024),
                                                         16, woff, 0, 0, 0);
        }
        ++kiss;
        c0 += BK;
        if (c0 == a.CK) {
            c0 = 0; ++ts;
            if (ts == a.kw) { ts = 0; ++tr; }
        }
    };
    // DUAL (two 1x1 convs: no tap state): STATELESS issue of global stage g.  The mutable captures of issue_stage end
    // up in scratch memory once a second code path touches them, and a scratch load queues behind every DMA in flight
    // (VMEM returns in order): the ring would drain on every K step.
    auto issue_dual = [&](int slot, int g) {
        char* base = lds + slot * TILE;
        const bool second = g >= nk1;                            // wave-uniform
        const unsigned koff = (unsigned)((second ? g - nk1 : g) * BK);
        const __amdgpu_buffer_rsrc_t qx = second ? rx2 : rx, qw = second ? rw2 : rw;
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const unsigned b = second ? xbase2[DUAL ? i : 0] : xbase[i];
            const unsigned off = b == kOOB ? kOOB : b + koff;
            if ((i * 256 + wave * 64) < XCH)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(qx, (__attribute__((address_space(3))) void*)(base + i * 4096 + wave * 1024), 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < WL; ++j) {
            const unsigned b = second ? wbase2[DUAL ? j : 0] : wbase[j];
            const unsigned woff = b == kOOB ? kOOB : b + koff;
            if ((j * 256 + wave * 64) < WCH)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(qw, (__attribute__((address_space(3))) void*)(base + XBYTES + j * 4096 + wave * 1024), 16, woff, 0, 0, 0);
        }
    };
    // DMA instructions THIS wave issues per stage (compile-time per wave index is not available, so
    // take the count for wave 0 when all waves are equal, else the exact per-wave value at run time)
    constexpr bool UNIFORM_LD = (XCH % 256 == 0) && (WCH % 256 == 0);
    int my_ld = 0;
    if (!UNIFORM_LD) {
#pragma unroll
        for (int i = 0; i < XL; ++i) my_ld += ((i * 256 + wave * 64) < XCH) ? 1 : 0;
#pragma unroll
        for (int j = 0; j < WL; ++j) my_ld += ((j * 256 + wave * 64) < WCH) ? 1 : 0;
    }
    // wait until at most `ahead` later stages of this wave's DMA are still in flight
    auto wait_ahead = [&](int ahead) {
        if (UNIFORM_LD) {
            if (ahead >= 2 && STAGES >= 4) wait_vmcnt<2 * NLD>();
            else if (ahead >= 1 && STAGES >= 3) wait_vmcnt<1 * NLD>();
            else wait_vmcnt<0>();
        } else {
            // ragged tiles (some waves own fewer slots): counts differ per wave; my_ld in {0,1,2,..}
            const int n = ahead * my_ld;
            if (n >= 4) wait_vmcnt<4>(); else if (n == 3) wait_vmcnt<3>(); else if (n == 2) wait_vmcnt<2>();
            else if (n == 1) wait_vmcnt<1>(); else wait_vmcnt<0>();
        }
    };

    // prologue: STAGES-1 tiles in flight
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) { if constexpr (DUAL) issue_dual(s, s); else issue_stage(s); }
#ifdef F8_TRACE
    t_pro = __builtin_readcyclecounter();
#endif

    auto k_step = [&](int ks, v16i (&ac)[TCO][TPX]) {
        // stages issued so far: min(nk, ks + STAGES - 1); stage ks must have landed
        const int issued = (ks + STAGES - 1 < nk) ? ks + STAGES - 1 : nk;
        wait_ahead(issued - 1 - ks);
        __builtin_amdgcn_s_barrier();     // all waves' DMA for stage ks landed; slot (ks-1)%STAGES is free
#ifdef F8_TRACE
        if (ks == 0) t_first = __builtin_readcyclecounter();
#endif
#ifndef F8_ABL_NO_DMA
        if (ks + STAGES - 1 < nk) { if constexpr (DUAL) issue_dual((ks + STAGES - 1) % STAGES, ks + STAGES - 1); else issue_stage((ks + STAGES - 1) % STAGES); }
#endif
        const char* base = lds + (ks % STAGES) * TILE;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            v4i wf[TCO], xf[TPX];
#pragma unroll
            for (int i = 0; i < TCO; ++i) wf[i] = *(const v4i*)(base + wfrag0 + i * 32 * BK + coff[kk]);
#pragma unroll
            for (int j = 0; j < TPX; ++j) xf[j] = *(const v4i*)(base + xfrag0 + j * 32 * BK + coff[kk]);
#pragma unroll
            for (int i = 0; i < TCO; ++i)
#pragma unroll
                for (int j = 0; j < TPX; ++j)
#ifndef F8_ABL_NO_MFMA
                    ac[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[i], xf[j], ac[i][j], 0, 0, 0);
#else
                    ac[i][j][0] += wf[i].x ^ xf[j].x;
#endif
        }
    };
    for (int ks = 0; ks < nk1; ++ks) k_step(ks, acc);
    if constexpr (DUAL) {
        v16i acc2[TCO][TPX];
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int j = 0; j < TPX; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0;
        for (int ks = nk1; ks < nk; ++ks) k_step(ks, acc2);
        // the second product (+ its bias) is the residual operand; fragment order == I32T order
#pragma unroll
        for (int i = 0; i < TCO; ++i) {
            const int cot = co0 + wco * (BN / WCO) + i * 32;
            if (cot >= a.coutP) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4i bv = *(const v4i*)(a.bias2 + cot + 8 * g + 4 * lh);
#pragma unroll
                for (int j = 0; j < TPX; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) rv[i][j][g][e] = (int)((unsigned)acc2[i][j][4 * g + e] + (unsigned)bv[e]);
            }
        }
    }
#ifdef F8_ABL_NO_EPI
    if (a.M > 0) {   // ablation: one dummy store per lane keeps the accumulators live, nothing else
        int t = 0;
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int j = 0; j < TPX; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t ^= acc[i][j][r];
        if (t == 0x12345678) a.q[0].ptr[tid] = (int8_t)t;
        return;
    }
#endif

#ifdef F8_TRACE
    t_loop = __builtin_readcyclecounter();
#endif
    conv_epilogue<BM, BN, WPX, WCO, HAS_RES, TCO, TPX>(a, acc, rv, m0, co0, wpx, wco, l31, lh);
#ifdef F8_TRACE
    if (a.trace && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = __builtin_readcyclecounter();
        unsigned long long* tp = (unsigned long long*)a.trace + (size_t)blockIdx.x * 8;
        tp[0] = t_start; tp[1] = t_pro; tp[2] = t_first; tp[3] = t_loop; tp[4] = t_end;
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        tp[5] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// Depthwise 3x3 (VALU).  One thread = one output pixel x 4 channels (one dword of NHWC int8).
// Unsigned inputs are multiplied as unsigned bytes directly: no offset trick needed here.
// ---------------------------------------------------------------------------------------------
template <bool SIGNED_IN>
__global__ void __launch_bounds__(256) dwconv3x3_kernel(const DwArgs a) {
    const int cgs = a.Cs >> 2;
    const size_t total = (size_t)a.N * a.P * a.Q * cgs;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % cgs);
        size_t m = idx / cgs;
        const int q = (int)(m % a.Q); m /= a.Q;
        const int p = (int)(m % a.P);
        const int n = (int)(m / a.P);
        const int c = cg << 2;
        int acc[4];
        {
            const v4i b = *(const v4i*)(a.bias + c);
            acc[0] = b.x; acc[1] = b.y; acc[2] = b.z; acc[3] = b.w;
        }
        const int h0 = p * a.stride - a.pad, w0 = q * a.stride - a.pad;
        const unsigned in_xor = SIGNED_IN ? 0u : 0x80808080u;   // unsigned tensors are stored biased
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int h = h0 + r;
            if ((unsigned)h >= (unsigned)a.H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int w = w0 + s;
                if ((unsigned)w >= (unsigned)a.W) continue;
                const unsigned xv = *(const unsigned*)(a.x + (((size_t)n * a.H + h) * a.W + w) * a.Cs + c) ^ in_xor;
                const unsigned wv = *(const unsigned*)(a.w + (r * 3 + s) * a.Cs + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int xe = SIGNED_IN ? (int)(signed char)(xv >> (8 * e)) : (int)((xv >> (8 * e)) & 0xffu);
                    const int we = (int)(signed char)(wv >> (8 * e));
                    acc[e] = (int)((unsigned)acc[e] + (unsigned)(xe * we));
                }
            }
        }
        if (a.relu0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = max(acc[e], 0);
        }
        const int mo = (n * a.P + p) * a.Q + q;
        const size_t o = (size_t)mo * a.Cs + c;
        if (a.out32) { v4i v = {acc[0], acc[1], acc[2], acc[3]}; *(v4i*)(a.out32 + i32t_index(mo, c, a.Cs)) = v; }
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (a.q[k].ptr)
                *(unsigned*)(a.q[k].ptr + o) =
                    pack4(requant1(acc[0], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(acc[1], a.q[k].n, a.q[k].lo, a.q[k].hi),
                          requant1(acc[2], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(acc[3], a.q[k].n, a.q[k].lo, a.q[k].hi)) ^ a.q[k].bias_xor;
    }
}

// ---------------------------------------------------------------------------------------------
// Depthwise 3x3 on v_dot4_i32_i8.  One thread = 16 channels x PIX adjacent output pixels.
// Depthwise products do not reduce across channels, so the 4 bytes a dot4 reduces over must be 4 TAPS of
// one channel: per 4-channel quad the dwords of taps t0..t3 are byte-transposed with 8 v_perm_b32 and
// consumed by 4 dot4 (weights are stored pre-transposed); taps 0-3 and 4-7 take this route, tap 8 is a
// scalar multiply-add.  ~0.9 VALU per MAC instead of ~2.5 for byte-extract + mad.
// Everything stays in the stored domain: unsigned activations are biased (x - 128 as int8), out-of-image
// taps are replaced by the biased zero, and 128 * sum(w) sits in the bias (pack_dw_weights).
// ---------------------------------------------------------------------------------------------
template <int S, int PIX>
__global__ void __launch_bounds__(256) dwconv3x3_dot4_kernel(const DwArgs a) {
    constexpr int NCOL = (PIX - 1) * S + 3;
    const int cgs = a.Cs >> 4;                       // 16-channel groups
    const int QS = (a.Q + PIX - 1) / PIX;
    const size_t total = (size_t)a.N * a.P * QS * cgs;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cg = (int)(idx % cgs);
    size_t st = idx / cgs;
    const int qs = (int)(st % QS); st /= QS;
    const int p = (int)(st % a.P);
    const int n = (int)(st / a.P);
    const int c = cg << 4, q0 = qs * PIX;
    const unsigned padv = a.in_signed ? 0u : 0x80808080u;
    const int h0 = p * S - 1, w0 = q0 * S - 1;

    // input window: 3 rows x NCOL columns x 16 channels
    v4i x[3][NCOL];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int h = h0 + r;
#pragma unroll
        for (int cc = 0; cc < NCOL; ++cc) {
            const int w = w0 + cc;
            v4i v = {(int)padv, (int)padv, (int)padv, (int)padv};
            if ((unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W)
                v = *(const v4i*)(a.x + (((size_t)n * a.H + h) * a.W + w) * a.Cs + c);
            x[r][cc] = v;
        }
    }
    const int floor0 = a.relu0 ? 0 : INT32_MIN;
    unsigned outw[PIX][2][4];                        // packed int8 results [pixel][format][quad]
#pragma unroll
    for (int k = 0; k < 4; ++k) {                    // 4-channel quad inside the 16
        const unsigned* wq = (const unsigned*)a.w + (size_t)((c >> 2) + k) * 9;    // [quad][wA0..3, wB0..3, wC]
        unsigned wA[4], wB[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { wA[e] = wq[e]; wB[e] = wq[4 + e]; }
        const unsigned wC = wq[8];
        const v4i bv = *(const v4i*)(a.bias + c + 4 * k);
#pragma unroll
        for (int j = 0; j < PIX; ++j) {
            // taps 0..8 of this quad for pixel j
            unsigned t[9];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s2 = 0; s2 < 3; ++s2) t[r * 3 + s2] = (unsigned)x[r][j * S + s2][k];
            int acc[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int grp = 0; grp < 2; ++grp) {
                const unsigned t0 = t[grp * 4], t1 = t[grp * 4 + 1], t2 = t[grp * 4 + 2], t3 = t[grp * 4 + 3];
                const unsigned lo01 = __builtin_amdgcn_perm(t1, t0, 0x05010400u), hi01 = __builtin_amdgcn_perm(t1, t0, 0x07030602u);
                const unsigned lo23 = __builtin_amdgcn_perm(t3, t2, 0x05010400u), hi23 = __builtin_amdgcn_perm(t3, t2, 0x07030602u);
                const unsigned c0 = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u), c1 = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);
                const unsigned c2 = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u), c3 = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u);
                const unsigned* ww = grp == 0 ? wA : wB;
                acc[0] = __builtin_amdgcn_sdot4((int)c0, (int)ww[0], acc[0], false);
                acc[1] = __builtin_amdgcn_sdot4((int)c1, (int)ww[1], acc[1], false);
                acc[2] = __builtin_amdgcn_sdot4((int)c2, (int)ww[2], acc[2], false);
                acc[3] = __builtin_amdgcn_sdot4((int)c3, (int)ww[3], acc[3], false);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc[e] = (int)((unsigned)acc[e] + (unsigned)((int)(signed char)(t[8] >> (8 * e)) * (int)(signed char)(wC >> (8 * e))));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = max(acc[e], floor0);
#pragma unroll
            for (int f = 0; f < 2; ++f)
                if (a.q[f].ptr)
                    outw[j][f][k] = pack4(requant1(acc[0], a.q[f].n, a.q[f].lo, a.q[f].hi), requant1(acc[1], a.q[f].n, a.q[f].lo, a.q[f].hi),
                                          requant1(acc[2], a.q[f].n, a.q[f].lo, a.q[f].hi), requant1(acc[3], a.q[f].n, a.q[f].lo, a.q[f].hi)) ^ a.q[f].bias_xor;
        }
    }
#pragma unroll
    for (int j = 0; j < PIX; ++j) {
        if (q0 + j >= a.Q) continue;
        const size_t o = ((((size_t)n * a.P + p) * a.Q + q0 + j)) * a.Cs + c;
#pragma unroll
        for (int f = 0; f < 2; ++f)
            if (a.q[f].ptr) {
                v4i v = {(int)outw[j][f][0], (int)outw[j][f][1], (int)outw[j][f][2], (int)outw[j][f][3]};
                *(v4i*)(a.q[f].ptr + o) = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------
// Max-pool (NHWC).  int32 input: exact max, then any of {int32, two requantised int8} outputs.
// int8 input (already in the single consumer format; requant is monotone so pooling commutes with
// it exactly): per-byte signed max — unsigned tensors are stored biased (x ^ 0x80), which preserves order.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) maxpool_kernel(const PoolArgs a) {
    const int cgs = a.Cs >> 2;
    const size_t total = (size_t)a.N * a.P * a.Q * cgs;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % cgs);
        size_t m = idx / cgs;
        const int q = (int)(m % a.Q); m /= a.Q;
        const int p = (int)(m % a.P);
        const int n = (int)(m / a.P);
        const int c = cg << 2;
        int mx[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
        const int h0 = p * a.stride - a.pad, w0 = q * a.stride - a.pad;
        for (int r = 0; r < a.k; ++r) {
            const int h = h0 + r;
            if ((unsigned)h >= (unsigned)a.H) continue;
            for (int s = 0; s < a.k; ++s) {
                const int w = w0 + s;
                if ((unsigned)w >= (unsigned)a.W) continue;
                const int mi = (n * a.H + h) * a.W + w;
                if (a.in_is_i8) {
                    const unsigned xv = *(const unsigned*)((const int8_t*)a.x + (size_t)mi * a.Cs + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        mx[e] = max(mx[e], (int)(signed char)(xv >> (8 * e)));   // biased u8 keeps its order as s8
                    }
                } else {
                    const v4i xv = *(const v4i*)((const int32_t*)a.x + i32t_index(mi, c, a.Cs));
                    mx[0] = max(mx[0], xv.x); mx[1] = max(mx[1], xv.y);
                    mx[2] = max(mx[2], xv.z); mx[3] = max(mx[3], xv.w);
                }
            }
        }
        const int mo = (n * a.P + p) * a.Q + q;
        const size_t o = (size_t)mo * a.Cs + c;
        if (a.in_is_i8) {
            *(unsigned*)(a.q[0].ptr + o) = pack4(mx[0], mx[1], mx[2], mx[3]);
        } else {
            if (a.out32) { v4i v = {mx[0], mx[1], mx[2], mx[3]}; *(v4i*)(a.out32 + i32t_index(mo, c, a.Cs)) = v; }
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (a.q[k].ptr)
                    *(unsigned*)(a.q[k].ptr + o) =
                        pack4(requant1(mx[0], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(mx[1], a.q[k].n, a.q[k].lo, a.q[k].hi),
                              requant1(mx[2], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(mx[3], a.q[k].n, a.q[k].lo, a.q[k].hi)) ^ a.q[k].bias_xor;
        }
    }
}

// int8 -> int8 max-pool, 16 channels (one dwordx4) per thread.  Bytes are made unsigned-comparable (stored ^ 0x80 is
// order-preserving for plain s8 and for biased u8 alike), split into even / odd bytes and reduced with packed 16-bit
// max: 6 VALU operations per dword and tap instead of 12, a quarter of the memory instructions.
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) maxpool_i8x16_kernel(const PoolArgs a) {
    const int cgs = a.Cs >> 4;
    const size_t total = (size_t)a.N * a.P * a.Q * cgs;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % cgs);
        size_t m = idx / cgs;
        const int q = (int)(m % a.Q); m /= a.Q;
        const int p = (int)(m % a.P);
        const int n = (int)(m / a.P);
        us2 ev[4], od[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) { ev[d] = us2{0, 0}; od[d] = us2{0, 0}; }
        const int h0 = p * a.stride - a.pad, w0 = q * a.stride - a.pad;
        for (int r = 0; r < a.k; ++r) {
            const int h = h0 + r;
            if ((unsigned)h >= (unsigned)a.H) continue;
            for (int s = 0; s < a.k; ++s) {
                const int w = w0 + s;
                if ((unsigned)w >= (unsigned)a.W) continue;
                const size_t mi = ((size_t)n * a.H + h) * a.W + w;
                const v4i xv = *(const v4i*)((const int8_t*)a.x + mi * a.Cs + cg * 16);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const unsigned u = (unsigned)xv[d] ^ 0x80808080u;
                    const unsigned e = u & 0x00ff00ffu, o = (u >> 8) & 0x00ff00ffu;
                    ev[d] = __builtin_elementwise_max(ev[d], __builtin_bit_cast(us2, e));
                    od[d] = __builtin_elementwise_max(od[d], __builtin_bit_cast(us2, o));
                }
            }
        }
        v4i out;
#pragma unroll
        for (int d = 0; d < 4; ++d)
            out[d] = (int)((__builtin_bit_cast(unsigned, ev[d]) | (__builtin_bit_cast(unsigned, od[d]) << 8)) ^ 0x80808080u);
        const size_t mo = ((size_t)n * a.P + p) * a.Q + q;
        *(v4i*)(a.q[0].ptr + mo * a.Cs + cg * 16) = out;
    }
}

// FXQAvgPool2d int branch: int64 sum over H*W, truncate to int32 (fix_quant_ops.py:130-133).
// Eight lanes per (image, 4 channels): lane j sums the pixels j, j + 8, ... (consecutive lanes read consecutive 16-byte pieces of an
// I32T tile), three xor-shuffles add the partial sums.  (One thread per (image, 4 channels) walking all 49 pixels was 64 workgroups
// and a chain of dependent loads for ResNet-18's 12.8 MB: 14 us.)  int32 wrapping adds: what the reference's int64 sum becomes when
// its result is narrowed (fix_quant_ops.py: FXQAvgPool2d).
__global__ void __launch_bounds__(256) avgpool_kernel(const AvgArgs a) {
    const int cgs = a.Cs >> 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = idx & 7, item = idx >> 3;
    const bool live = item < a.N * cgs;                  // (a whole group of 8 lanes is live or not: the shuffles below stay inside it)
    const int it = live ? item : 0;
    const int cg = it % cgs, n = it / cgs, c = cg << 2;
    unsigned t[4] = {0u, 0u, 0u, 0u};
    for (int i = j; i < a.HW; i += 8) {
        const v4i v = *(const v4i*)(a.x + i32t_index(n * a.HW + i, c, a.Cs));
        t[0] += (unsigned)v.x; t[1] += (unsigned)v.y; t[2] += (unsigned)v.z; t[3] += (unsigned)v.w;
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] += (unsigned)__shfl_xor((int)t[e], m);
    if (!live || j != 0) return;
    const size_t o = (size_t)n * a.Cs + c;
    if (a.out32) { v4i v = {(int)t[0], (int)t[1], (int)t[2], (int)t[3]}; *(v4i*)(a.out32 + i32t_index(n, c, a.Cs)) = v; }
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (a.q[k].ptr)
            *(unsigned*)(a.q[k].ptr + o) =
                pack4(requant1((int)t[0], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1((int)t[1], a.q[k].n, a.q[k].lo, a.q[k].hi),
                      requant1((int)t[2], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1((int)t[3], a.q[k].n, a.q[k].lo, a.q[k].hi)) ^ a.q[k].bias_xor;
}

// Stand-alone residual join (only when it cannot ride in a conv epilogue) and stand-alone requant
// (b == nullptr; third and later int8 formats of one tensor).  One thread = one pixel x 4 channels.
__global__ void __launch_bounds__(256) add_kernel(const AddArgs a) {
    const int cgs = a.Cs >> 2;
    const size_t total = (size_t)a.M * cgs;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cgs) << 2, m = (int)(idx / cgs);
        const size_t it = i32t_index(m, c, a.Cs);
        const v4i av = *(const v4i*)(a.a + it);
        int y[4] = {av.x, av.y, av.z, av.w};
        if (a.b) {
            const v4i bv = *(const v4i*)(a.b + it);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = clamp_sym31((int)(((unsigned)av[e] << a.a_shl) + ((unsigned)bv[e] << a.b_shl)));
                if (a.relu) y[e] = max(y[e], 0);
            }
        }
        if (a.out32) { v4i v = {y[0], y[1], y[2], y[3]}; *(v4i*)(a.out32 + it) = v; }
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (a.q[k].ptr)
                *(unsigned*)(a.q[k].ptr + (size_t)m * a.Cs + c) =
                    pack4(requant1(y[0], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(y[1], a.q[k].n, a.q[k].lo, a.q[k].hi),
                          requant1(y[2], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(y[3], a.q[k].n, a.q[k].lo, a.q[k].hi)) ^ a.q[k].bias_xor;
    }
}

// Network input: int32 NCHW -> zero-haloed NHWC4 int8 (stem) / NHWC int8 / NHWC int32.
// One thread per (n, h, w); C planes are read coalesced along w.
// Input quantisation of forward_loss (fix_train.py:683-692) on one fp32 value: rint(x * scale) (round half to even,
// one IEEE multiply: 255 or 2^fl), clamped to [lo, hi] (normalize: +-127 or [0,255], fix_quant_ops.py:64-87;
// the u8 path has no clamp in the reference: lo / hi = int32 range there).
__device__ __forceinline__ int quant_in(float x, float scale, int lo, int hi) {
    const float r = rintf(__fmul_rn(x, scale));
    return (int)fminf(fmaxf(r, (float)lo), (float)hi);
}

__global__ void __launch_bounds__(256) quantize_input_kernel(const float* x, int32_t* y, size_t n, float scale, int lo, int hi) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = quant_in(x[i], scale, lo, hi);
}

// ImageNet scoring of forward_loss (fix_train.py:697-704): correct[k][n] = 1 if target n is among the k largest logits.
// Rank of the target = #{logit > logit[target]} + #{logit == logit[target], index < target} (ties by lower index, the
// order a stable descending sort gives); one wave per image.
__global__ void __launch_bounds__(256) topk_correct_kernel(const float* logits, const int64_t* target, int N, int C, const TopkKs ks, int nk, float* correct) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int t = (int)target[n];
    const float* row = logits + (size_t)n * C;
    const bool valid = t >= 0 && t < C;
    const float lt = valid ? row[t] : 0.f;
    int rank = 0;
    for (int c = lane; c < C; c += 64) {
        const float v = row[c];
        rank += (v > lt || (v == lt && c < t)) ? 1 : 0;
    }
    for (int off = 32; off > 0; off >>= 1) rank += __shfl_down(rank, off, 64);
    if (lane == 0)
        for (int k = 0; k < nk; ++k) correct[(size_t)k * N + n] = (valid && rank < ks.k[k]) ? 1.f : 0.f;
}

// Stem form only, W % 4 == 0: one thread = 4 consecutive pixels (3 dwordx4 plane reads, one dwordx4 write).  KIND: 0 int32 planes,
// 1 fp32 planes (quantised here), 2 uint8 NCHW planes, 3 uint8 NHWC pixels (both through the LUT); C is a compile-time 3 for the
// image case so the channel loop unrolls and the pixel values stay in registers (the runtime-C form indexed a private array with a
// loop counter: 2.5 TB/s on a pure streaming kernel).
template <int KIND, int CT>
__global__ void __launch_bounds__(256) input_stem4_kernel(const InArgs a) {
    const int W4 = a.W >> 2;
    const int C = CT ? CT : a.C;
    const size_t total = (size_t)a.N * a.H * W4;
    const size_t plane = (size_t)a.H * a.W;
    unsigned bad = 0;                                     // KIND 0: an int32 value outside the 8-bit format it is narrowed to
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(idx % W4) * 4;
        size_t t = idx / W4;
        const int h = (int)(t % a.H);
        const int n = (int)(t / a.H);
        const size_t pix0 = ((size_t)n * C * a.H + h) * a.W + w;
        int v[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool live = CT ? c < CT : c < C;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[c][j] = 0;
            if (!live) continue;
            if constexpr (KIND == 3) {
                const int16_t* lut = a.lut + (c < 3 ? c : 2) * 256;
                const uint8_t* p = a.xu8 + ((((size_t)n * a.H + h) * a.W) + w) * C + c;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[c][j] = lut[p[(size_t)j * C]];
            } else if constexpr (KIND == 2) {
                // uint8 pixels (1 B / px / channel instead of 4): one dword of the plane
                const int16_t* lut = a.lut + (c < 3 ? c : 2) * 256;
                const unsigned d = *(const unsigned*)(a.xu8 + pix0 + c * plane);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[c][j] = lut[(d >> (8 * j)) & 0xffu];
            } else if constexpr (KIND == 1) {
                const v4f f = *(const v4f*)(a.xf + pix0 + c * plane);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[c][j] = quant_in(f[j], a.scale, a.qlo, a.qhi);
            } else {
                const v4i x = __builtin_nontemporal_load((const v4i*)(a.x + pix0 + c * plane));
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[c][j] = x[j]; bad |= (unsigned)(x[j] - a.chk_lo) > (unsigned)(a.chk_hi - a.chk_lo) ? 1u : 0u; }
            }
        }
        v4i o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (int)(pack4(v[0][j], v[1][j], v[2][j], v[3][j]) ^ a.xor8);
        int8_t* dst = a.stem + ((((size_t)n * a.Hp + h + a.pad) * a.Wp) + w + a.pad) * 4;
        if ((((size_t)dst) & 15) == 0) *(v4i*)dst = o;                 // the halo may leave rows 4-byte aligned only
        else { ((int*)dst)[0] = o[0]; ((int*)dst)[1] = o[1]; ((int*)dst)[2] = o[2]; ((int*)dst)[3] = o[3]; }
    }
    if constexpr (KIND == 0) { if (a.err && bad) atomicOr(a.err, 1u); }
}

__global__ void __launch_bounds__(256) input_kernel(const InArgs a) {
    const size_t total = (size_t)a.N * a.H * a.W;
    unsigned bad = 0;                                     // an int32 value outside the 8-bit format it is narrowed to
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(idx % a.W);
        size_t t = idx / a.W;
        const int h = (int)(t % a.H);
        const int n = (int)(t / a.H);
        const size_t pix0 = ((size_t)n * a.C * a.H + h) * a.W + w;
        const size_t plane = (size_t)a.H * a.W;
        // fp32 images are quantised on the fly (f8_net_run_f32): the int32 tensor of the reference never exists
        auto ld = [&](int c) -> int {
            if (a.xu8) return a.lut[(c < 3 ? c : 2) * 256 + (a.u8_nhwc ? a.xu8[((((size_t)n * a.H + h) * a.W) + w) * a.C + c] : a.xu8[pix0 + c * plane])];
            if (a.xf) return quant_in(a.xf[pix0 + c * plane], a.scale, a.qlo, a.qhi);
            const int v = a.x[pix0 + c * plane];
            bad |= (unsigned)(v - a.chk_lo) > (unsigned)(a.chk_hi - a.chk_lo) ? 1u : 0u;
            return v;
        };
        if (a.stem) {
            int v[4] = {0, 0, 0, 0};
            for (int c = 0; c < a.C; ++c) v[c] = ld(c);
            *(unsigned*)(a.stem + ((((size_t)n * a.Hp + h + a.pad) * a.Wp) + w + a.pad) * 4) = pack4(v[0], v[1], v[2], v[3]) ^ a.xor8;
        }
        if (a.out8) {
            int8_t* o = a.out8 + (((size_t)n * a.H + h) * a.W + w) * a.Cs8;
            const int8_t bx = (int8_t)(a.xor8 & 0xff);
            for (int c = 0; c < a.C; ++c) o[c] = (int8_t)(ld(c)) ^ bx;
            for (int c = a.C; c < a.Cs8; ++c) o[c] = bx;
        }
        if (a.out32) {
            const int m = (n * a.H + h) * a.W + w;
            for (int c = 0; c < a.Cs32; ++c) a.out32[i32t_index(m, c, a.Cs32)] = c < a.C ? ld(c) : 0;
        }
    }
    if (a.err && bad) atomicOr(a.err, 1u);                // (err is only passed when an 8-bit form is written by plain narrowing)
}

// Network output: NHWC int32 (row stride Cs) -> NCHW int32 / float32.
__global__ void __launch_bounds__(256) output_kernel(const OutArgs a) {
    const size_t total = (size_t)a.N * a.C * a.HW;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % a.HW);
        size_t t = idx / a.HW;
        const int c = (int)(t % a.C);
        const int n = (int)(t / a.C);
        const int v = a.x[i32t_index(n * a.HW + i, c, a.Cs)];
        const bool bad = a.err != nullptr && (*a.err >> 8) == a.epoch;     // f8_fc.hip: a chain launch of this run gave up a halo wait
        if (a.as_float) ((float*)a.out)[idx] = bad ? __builtin_nanf("") : (float)v;
        else ((int*)a.out)[idx] = bad ? INT32_MIN : v;
    }
}

// ---- op-level element-wise kernels on flat int32 tensors (reference tensor format)
__global__ void __launch_bounds__(256) requant_i32_kernel(const int32_t* src, int32_t* dst, size_t n, int sh, int lo, int hi) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = requant1(src[i], sh, lo, hi);
}
__global__ void __launch_bounds__(256) relu_i32_kernel(int32_t* x, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        x[i] = max(x[i], 0);
}
__global__ void __launch_bounds__(256) add_align_i32_kernel(int32_t* res, const int32_t* x, size_t n, int res_shl, int x_shl) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        res[i] = clamp_sym31((int)(((unsigned)res[i] << res_shl) + ((unsigned)x[i] << x_shl)));
}

// ---------------------------------------------------------------------------------------------
// Host launchers
// ---------------------------------------------------------------------------------------------
static inline int grid_for(size_t work, int block = 256, int cap = 256 * 8 * 4) {
    size_t g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > (size_t)cap) g = cap;
    return (int)g;
}

bool pick_conv_tile(int M, int coutP, int ck, bool has_res, bool bk128, ConvTile* t) {
    if (ck % 32 != 0 || coutP % 32 != 0) return false;
    t->bk = (ck % 64 == 0) ? 64 : 32;
    if (bk128 && ck % 128 == 0 && coutP > 32) t->bk = 128;
    t->bn = coutP >= 96 ? 128 : (coutP > 32 ? 64 : 32);
    t->bm = 128;
    // residual-carrying epilogues hold the int32 operand in registers: the 64-wide tile keeps
    // 4 waves/SIMD resident instead of 2, which is what a streaming epilogue needs
    if (has_res && t->bn == 128) t->bn = 64;
    // not enough workgroups for 256 CUs: shrink the tile (bm first: keeps cout reuse of X rows)
    auto tiles = [&](int bm, int bn) { return (long)((M + bm - 1) / bm) * ((coutP + bn - 1) / bn); };
    if (tiles(t->bm, t->bn) < 512 && t->bn >= 64) t->bm = 64;
    if (tiles(t->bm, t->bn) < 512 && t->bn == 128) t->bn = 64;
    if (t->bm == 64 && t->bn == 32) t->bn = 64;
    return true;
}

// K loops of at least ConvArgs::deep_nk (Options::deep_nk, default 7) steps get the deepest ring that fits 64 KB of static LDS (else 2
// slots).  Measured (ResNet-50, bs 128, A/B/A/B): threshold 16 -> 64.0-64.5 k, 7 -> 64.7-64.9 k, 5 -> 65.2 k img/s; 7 takes in the
// 7-step stem (80 -> 75 us) and the 9-step 3x3s without touching the 4-step residual-carrying 1x1s.
int conv_grid(const ConvTile& t, int M, int coutP) {
    return ((M + t.bm - 1) / t.bm) * ((coutP + t.bn - 1) / t.bn);
}

static int g_num_cu = 0;
static int num_cus() {
    if (g_num_cu == 0) {
        int dev = 0; hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) g_num_cu = p.multiProcessorCount;
        if (g_num_cu <= 0) g_num_cu = 256;
    }
    return g_num_cu;
}

template <int BM, int BN, int BK, int WPX, int WCO>
static hipError_t launch_conv_t(const ConvArgs& a, int grid, hipStream_t s) {
#ifdef F8_FORCE_STAGES
    constexpr int ST = F8_FORCE_STAGES;
#else
    // measured on ResNet-50 (profiles/): ring depth 2/3/4 = 49.9k/49.5k/48.5k img/s for the one-tile-per-workgroup
    // kernel — bound by per-workgroup instruction issue and start-up, not by steady-state latency, so the
    // smaller LDS footprint (more resident workgroups) wins there
    constexpr int ST = 2;
#endif
    constexpr int TILE = (BM + BN) * BK;
    static_assert(ST * TILE <= 65536, "static LDS");
    const bool pad = a.pad > 0, res = a.res != nullptr;
    // long K loops (3x3 of the late stages: 36-72 steps of ~64-256 MFMA cycles against an ~800-cycle DMA round
    // trip) want a deeper ring; short ones want the smaller LDS footprint
    const int deep_nk = a.deep_nk > 0 ? a.deep_nk : 7;
    constexpr int DST = (4 * TILE <= 65536) ? 4 : ((3 * TILE <= 65536) ? 3 : 2);
    const bool deep = (DST > ST) && ((a.ktot + a.ktot2) / BK >= deep_nk);
    if (a.x2) {   // dual GEMM (downsample join): 1x1 / no padding; 64-wide cout tiles, or 128x128 for the wide late stages
        if constexpr (BK == 64 && (BN == 64 || (BN == 128 && BM == 128))) {
            if (pad || res) return hipErrorInvalidValue;
            if (deep) hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WPX, WCO, false, true, DST, true>), dim3(grid), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WPX, WCO, false, true, ST, true>), dim3(grid), dim3(256), 0, s, a);
            return hipGetLastError();
        } else return hipErrorInvalidValue;
    }
#define F8_LAUNCH(PAD_, RES_) \
    do { if (deep) hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WPX, WCO, PAD_, RES_, DST>), dim3(grid), dim3(256), 0, s, a); \
         else hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WPX, WCO, PAD_, RES_, ST>), dim3(grid), dim3(256), 0, s, a); } while (0)
    if (pad && res) F8_LAUNCH(true, true);
    else if (pad) F8_LAUNCH(true, false);
    else if (res) F8_LAUNCH(false, true);
    else F8_LAUNCH(false, false);
#undef F8_LAUNCH
    return hipGetLastError();
}

#ifdef F8_TRACE
static int g_trace_launch = 0;
static void* g_trace_buf = nullptr;
#endif

hipError_t launch_conv(const ConvArgs& a0, const ConvTile& t, hipStream_t s) {
    ConvArgs a = a0;
    const int grid = conv_grid(t, a.M, a.coutP);
#ifdef F8_TRACE
    static const int want = [] { const char* e = getenv("F8_TRACE_LAUNCH"); return e ? atoi(e) : -1; }();
    const bool tracing = (g_trace_launch++ == want);
    if (tracing) {
        if (!g_trace_buf) (void)hipMalloc((void**)&g_trace_buf, (size_t)1 << 24);
        (void)hipMemsetAsync(g_trace_buf, 0, (size_t)grid * 64, s);
        a.trace = g_trace_buf;
    }
    struct Dump { bool on; int grid; hipStream_t s; ~Dump() {
        if (!on) return;
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> h((size_t)grid * 8);
        (void)hipMemcpy(h.data(), g_trace_buf, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0; double pro = 0, first = 0, loop = 0, epi = 0; int n = 0;
        for (int i = 0; i < grid; ++i) { auto* p = &h[(size_t)i * 8]; if (!p[4]) continue; ++n; t0 = std::min(t0, p[0]); t1 = std::max(t1, p[4]);
            pro += p[1] - p[0]; first += p[2] - p[1]; loop += p[3] - p[2]; epi += p[4] - p[3]; }
        fprintf(stderr, "[trace] grid %d traced %d span %llu cyc | avg per WG: prologue %.0f  first-wait %.0f  loop %.0f  epilogue+drain %.0f  (cycles)\n",
                grid, n, t1 - t0, pro / n, first / n, loop / n, epi / n);
        // start-time histogram (in 1/8 of span)
        int hist[8] = {0}, hend[8] = {0};
        for (int i = 0; i < grid; ++i) { auto* p = &h[(size_t)i * 8]; if (!p[4]) continue; hist[std::min<unsigned long long>(7, (p[0] - t0) * 8 / (t1 - t0 + 1))]++; hend[std::min<unsigned long long>(7, (p[4] - t0) * 8 / (t1 - t0 + 1))]++; }
        fprintf(stderr, "[trace] WG starts per eighth of span:"); for (int k = 0; k < 8; ++k) fprintf(stderr, " %d", hist[k]);
        fprintf(stderr, " | ends:"); for (int k = 0; k < 8; ++k) fprintf(stderr, " %d", hend[k]); fprintf(stderr, "\n");
    } } dump{tracing, grid, s};
#endif
#define F8_CASE(BM_, BN_, BK_, WPX_, WCO_) \
    if (t.bm == BM_ && t.bn == BN_ && t.bk == BK_) return launch_conv_t<BM_, BN_, BK_, WPX_, WCO_>(a, grid, s);
    F8_CASE(128, 128, 128, 2, 2)
    F8_CASE(128, 64, 128, 4, 1)
    F8_CASE(64, 128, 128, 2, 2)
    F8_CASE(64, 64, 128, 2, 2)
    F8_CASE(128, 128, 64, 2, 2)
    F8_CASE(128, 64, 64, 4, 1)
    F8_CASE(128, 32, 64, 4, 1)
    F8_CASE(64, 128, 64, 2, 2)
    F8_CASE(64, 64, 64, 2, 2)
    F8_CASE(128, 128, 32, 2, 2)
    F8_CASE(128, 64, 32, 4, 1)
    F8_CASE(128, 32, 32, 4, 1)
    F8_CASE(64, 128, 32, 2, 2)
    F8_CASE(64, 64, 32, 2, 2)
#undef F8_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_dwconv(const DwArgs& a, hipStream_t s) {
    // dot4 kernel: int8 outputs only, stride 1 / 2, pad 1, channel stride a multiple of 16 (always: Cs % 32 == 0)
    if (a.use_mma && dwconv_mma_supported(a)) return launch_dwconv_mma(a, s);
    if (a.use_dot4 && !a.out32 && a.w4 && a.pad == 1 && (a.stride == 1 || a.stride == 2)) {
        const size_t work = (size_t)a.N * a.P * ((a.Q + 1) / 2) * (a.Cs >> 4);
        const unsigned grid = (unsigned)((work + 255) / 256);
        DwArgs b = a; b.w = a.w4; b.bias = a.bias4;
        if (a.stride == 1) hipLaunchKernelGGL((dwconv3x3_dot4_kernel<1, 2>), dim3(grid), dim3(256), 0, s, b);
        else hipLaunchKernelGGL((dwconv3x3_dot4_kernel<2, 2>), dim3(grid), dim3(256), 0, s, b);
        return hipGetLastError();
    }
    const size_t work = (size_t)a.N * a.P * a.Q * (a.Cs >> 2);
    if (a.in_signed) hipLaunchKernelGGL(dwconv3x3_kernel<true>, dim3(grid_for(work)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(dwconv3x3_kernel<false>, dim3(grid_for(work)), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_maxpool(const PoolArgs& a, hipStream_t s) {
    if (a.in_is_i8 && (a.Cs & 15) == 0 && a.pad < a.k) {     // every window holds at least one in-image tap (pad < k)
        hipLaunchKernelGGL(maxpool_i8x16_kernel, dim3(grid_for((size_t)a.N * a.P * a.Q * (a.Cs >> 4), 256, 1 << 20)), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    const size_t work = (size_t)a.N * a.P * a.Q * (a.Cs >> 2);
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(work)), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_avgpool(const AvgArgs& a, hipStream_t s) {
    const size_t work = (size_t)a.N * (a.Cs >> 2) * 8;
    hipLaunchKernelGGL(avgpool_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_add(const AddArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(add_kernel, dim3(grid_for((size_t)a.M * (a.Cs >> 2))), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_input(const InArgs& a, hipStream_t s) {
    if (a.stem && !a.out8 && !a.out32 && (a.W & 3) == 0 && a.C <= 4) {
        const dim3 grid(grid_for((size_t)a.N * a.H * (a.W >> 2), 256, 1 << 20));
        const int kind = a.xu8 ? (a.u8_nhwc ? 3 : 2) : a.xf ? 1 : 0;
#define F8_IN(K, CT) hipLaunchKernelGGL((input_stem4_kernel<K, CT>), grid, dim3(256), 0, s, a)
        if (a.C == 3) { if (kind == 0) F8_IN(0, 3); else if (kind == 1) F8_IN(1, 3); else if (kind == 2) F8_IN(2, 3); else F8_IN(3, 3); }
        else          { if (kind == 0) F8_IN(0, 0); else if (kind == 1) F8_IN(1, 0); else if (kind == 2) F8_IN(2, 0); else F8_IN(3, 0); }
#undef F8_IN
        return hipGetLastError();
    }
    hipLaunchKernelGGL(input_kernel, dim3(grid_for((size_t)a.N * a.H * a.W)), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_quantize_input(const float* x, int32_t* y, size_t n, float scale, int lo, int hi, hipStream_t s) {
    hipLaunchKernelGGL(quantize_input_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, y, n, scale, lo, hi);
    return hipGetLastError();
}
hipError_t launch_topk_correct(const float* logits, const int64_t* target, int N, int C, TopkKs ks, int nk, float* correct, hipStream_t s) {
    hipLaunchKernelGGL(topk_correct_kernel, dim3((N + 3) / 4), dim3(256), 0, s, logits, target, N, C, ks, nk, correct);
    return hipGetLastError();
}
hipError_t launch_output(const OutArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(output_kernel, dim3(grid_for((size_t)a.N * a.C * a.HW)), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_requant_i32(const int32_t* src, int32_t* dst, size_t n, int sh, int lo, int hi, hipStream_t s) {
    hipLaunchKernelGGL(requant_i32_kernel, dim3(grid_for(n)), dim3(256), 0, s, src, dst, n, sh, lo, hi);
    return hipGetLastError();
}
hipError_t launch_relu_i32(int32_t* x, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(relu_i32_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, n);
    return hipGetLastError();
}
hipError_t launch_add_align_i32(int32_t* res, const int32_t* x, size_t n, int res_shl, int x_shl, hipStream_t s) {
    hipLaunchKernelGGL(add_align_i32_kernel, dim3(grid_for(n)), dim3(256), 0, s, res, x, n, res_shl, x_shl);
    return hipGetLastError();
}

}  // namespace f8
