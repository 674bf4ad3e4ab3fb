// f8_wstat.hip — weight-stationary 1x1 convolution (gfx950): the weights live in registers, the pixels stream through LDS.
//
// The 1x1 convs around the stage-opening blocks of ResNet-50's stages 2 / 3 and the closing 1x1 of the 7x7 blocks (`layer_(res)` of
// /root/reference/models/fix_resnet.py:34 for body.0 / body.4 / shortcut.0, the join of :40-54) are GEMMs with 6 k - 100 k pixels,
// K = 256 - 1536 bytes and 256 - 2048 output channels.  On conv_igemm_kernel (and on f8_wreg.hip, which re-streams the weight matrix
// per 64 pixels) they ran at 0.6 - 0.9 POP/s: every tile pays a prologue, a K loop of 4 - 16 barriers and an epilogue, and re-reads
// its weights from L2.  Here a workgroup keeps ONE weight slice for its whole life:
//   * wave w of workgroup (mg, ng) owns output channels [32 (ng NW + w), +32): its K0 (+ K1) x 32 weights are loaded ONCE, in MFMA
//     fragment order (host: pack_frag_weights), into K/8 VGPRs — the A operand of every v_mfma_i32_32x32x32_i8 it ever issues;
//   * the workgroup walks its share of the 32-pixel tiles: a tile's rows (32 x K bytes; for a dual GEMM the rows of both inputs, the
//     shortcut's gathered at its stride) arrive by LDS-direct DMA in an S-stage ring, ONE barrier per tile; all NW waves read the same
//     B fragments (ds_read_b128, XOR-swizzled rows);
//   * epilogue per tile, straight from the accumulators: bias, ReLU, [dual: second product + bias as the join's other operand |
//     residual: int32 I32T operand prefetched at the top of the tile], align-add-clamp, ReLU, int32 I32T store (4 x 1 KB per wave)
//     and up to two requantised int8 rows (16-byte stores).  All global accesses of the loop are buffer operations whose out-of-range
//     lanes carry an out-of-bounds offset instead of being branched around: every wave then issues the SAME number of VMEM operations
//     per tile, which is what lets the ring use a counted `s_waitcnt vmcnt(N)` (VMEM retires in order; loads, DMAs and stores share
//     the counter) — tile i's rows are waited for while the DMAs of tiles i+1 .. i+S-2 and the stores of the last S-1 tiles fly.
// Bound: per tile a wave issues K/32 MFMAs (32 cycles each, two waves per SIMD) and reads 32 x K bytes from LDS; with 8 waves the LDS
// (128 B/clk) and the matrix pipe saturate together — the kernel runs at the LDS rate, about the MFMA peak, instead of at the launch
// + prologue rate of a tile-per-workgroup GEMM.
#include "f8_device.h"
#ifndef F8_WS_PF
#define F8_WS_PF 2
#endif
#ifdef F8_TRACE
#include <cstdio>
#include <cstdlib>
#endif

namespace f8 {

template <int K0, int K1, int NW>
struct WstatCfg {
    static constexpr int KT = K0 + K1;
    static constexpr int STAGE = 32 * KT;                          // bytes of one 32-pixel tile (both inputs)
    static constexpr int S = 4 * STAGE <= 128 * 1024 ? 4 : 3;      // ring depth
    static constexpr int WG_PER_CU = 1;
    static constexpr int LDS_BYTES = S * STAGE;
    static constexpr int L0 = K0 / (32 * NW), L1 = K1 / (32 * NW); // DMA instructions per thread and tile
    static_assert(K0 % (32 * NW) == 0 && K1 % (32 * NW) == 0 && K0 >= 256 && (K1 == 0 || K1 >= 256), "whole wave instructions; Swz rows >= 256 B");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <int K0, int K1, int NW, bool RES, bool OUT32, int NQ, bool FAST>
__global__ void __launch_bounds__(NW * 64) conv1x1_wstat_kernel(const ConvArgs a, const int T, const int MG, const int NG, const int dense) {
    using Cfg = WstatCfg<K0, K1, NW>;
    constexpr int S = Cfg::S, STAGE = Cfg::STAGE, L0 = Cfg::L0, L1 = Cfg::L1;
    constexpr int NK0 = K0 / 32, NK1 = K1 / 32;
    constexpr bool DUAL = K1 > 0;
    static_assert(!(DUAL && RES), "the join has one other operand");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
#ifdef F8_TRACE
    unsigned long long tt0 = __builtin_readcyclecounter(), tta = 0, ttw = 0, ttc = 0, tte = 0, ttl = 0, ttv = 0, ttb = 0;
#define F8_WT(var) do { const unsigned long long now_ = __builtin_readcyclecounter(); var += now_ - tta; tta = now_; } while (0)
#else
#define F8_WT(var)
#endif
    int mg, ng;
    {   // XCD-aware order: the NG channel groups of one pixel range run on one XCD (its rows are fetched into ONE L2)
        const int bid = blockIdx.x, xcd = bid & 7, r = bid >> 3;
        ng = r % NG; mg = (r / NG) * 8 + xcd;
    }
    if (mg >= MG) return;
    const int t0 = (int)((long long)mg * T / MG), nt = (int)((long long)(mg + 1) * T / MG) - t0;
    const int ct = ng * NW + wave;                       // this wave's cout tile
    const int ctiles = a.coutP >> 5;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL ? a.x2 : a.x), 0, DUAL ? a.x2_bytes : a.x_bytes, 0x00020000);
    const unsigned i32_bytes = (unsigned)T * 32u * (unsigned)a.coutP * 4u;
    const __amdgpu_buffer_rsrc_t ro32 = __builtin_amdgcn_make_buffer_rsrc((void*)a.out32, 0, a.out32 ? i32_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, RES ? i32_bytes : 0u, 0x00020000);
    const unsigned i8_bytes = (unsigned)a.M * (unsigned)a.coutP;
    const __amdgpu_buffer_rsrc_t rq0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.q[0].ptr, 0, a.q[0].ptr ? i8_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rq1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.q[1].ptr, 0, a.q[1].ptr ? i8_bytes : 0u, 0x00020000);

    // Source offsets of this thread's LDN DMA pieces of a tile.  Piece p = wave instruction j = wave + NW p' of its input: lane l
    // carries chunk (l % CPR) ^ f(row) of row j RPI + l / CPR (the swizzle is applied on the source side).  The per-tile part must be
    // a handful of instructions: a wave runs ONE instruction stream, and at ~4-5 cycles per issued instruction the first version's
    // scalar decode with its range branches (~450 SALU instructions per tile) cost more than the MFMAs and the epilogue together.
    //   dense input (stride 1): offset = tile 32 K + (row K + chunk 16): a per-lane constant plus one scalar; rows beyond M (and
    //     whole tiles beyond T) lie beyond the buffer's range: the DMA writes zeros, no explicit check;
    //   strided input: per-lane (n, p, q) decoding on the VALU (two magic-number divisions), branch-free.
    constexpr int LDN = L0 + L1;
    unsigned dconst[LDN];                                // dense: row K + swizzled chunk 16; strided: swizzled chunk 16
    unsigned drow[LDN];                                  // row of the tile this lane fetches in piece p
    {
        auto init_rows = [&](auto kc, auto lc, unsigned* dc, unsigned* dr, bool is_dense) {
            constexpr int K = decltype(kc)::value, L = decltype(lc)::value, CPR = K / 16, RPI = 64 / CPR;
            static_assert(CPR <= 64 && RPI * CPR == 64, "a wave instruction covers whole rows");
#pragma unroll
            for (int i = 0; i < L; ++i) {
                const int row = (wave + NW * i) * RPI + lane / CPR;
                const unsigned sw = (unsigned)(((lane % CPR) ^ Swz<K>::f(row)) * 16);
                dr[i] = (unsigned)row;
                dc[i] = is_dense ? (unsigned)(row * K) + sw : sw;
            }
        };
        init_rows(std::integral_constant<int, K0>{}, std::integral_constant<int, L0>{}, dconst, drow, (dense & 1) != 0);
        if constexpr (DUAL) init_rows(std::integral_constant<int, (DUAL ? K1 : 256)>{}, std::integral_constant<int, L1>{}, dconst + L0, drow + L0, (dense & 2) != 0);
    }
    auto dma_offsets = [&](int tile, unsigned* doff) {
        auto rows = [&](auto kc, auto lc, int p0, int sN, int sP, int sQ, bool is_dense) {
            constexpr int K = decltype(kc)::value, L = decltype(lc)::value;
            if (is_dense) {                                  // wave-uniform
                const unsigned tb = (unsigned)tile * (unsigned)(32 * K);
#pragma unroll
                for (int i = 0; i < L; ++i) doff[p0 + i] = tb + dconst[p0 + i];
            } else {
#pragma unroll
                for (int i = 0; i < L; ++i) {
                    const unsigned m = (unsigned)tile * 32u + drow[p0 + i];
                    const unsigned n = fast_div(m, a.mPQ, a.s1PQ, a.s2PQ), rem = m - n * (unsigned)a.PQ;
                    const unsigned pr = fast_div(rem, a.mQ, a.s1Q, a.s2Q), q = rem - pr * (unsigned)a.Q;
                    const unsigned o = n * (unsigned)sN + pr * (unsigned)sP + q * (unsigned)sQ + dconst[p0 + i];
                    doff[p0 + i] = m < (unsigned)a.M ? o : kOOB;
                }
            }
        };
        rows(std::integral_constant<int, K0>{}, std::integral_constant<int, L0>{}, 0, a.sN, a.sP, a.sQ, (dense & 1) != 0);
        if constexpr (DUAL) rows(std::integral_constant<int, (DUAL ? K1 : 256)>{}, std::integral_constant<int, L1>{}, L0, a.sN2, a.sP2, a.sQ2, (dense & 2) != 0);
    };
    // piece p: wave instruction j = wave + NW p' of its input deposits 1 KB (RPI rows x CPR chunks) at LDS slot j of that input's image
    auto dma_piece = [&](auto pc, int slot, const unsigned* doff) {
        constexpr int p = decltype(pc)::value;
#ifndef F8_ABL_WS_NODMA
        char* const base = smem + slot * STAGE;
        if constexpr (p < L0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(base + (wave + NW * p) * 1024), 16, doff[p], 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rx2, (__attribute__((address_space(3))) void*)(base + 32 * K0 + (wave + NW * (p - L0)) * 1024), 16, doff[p], 0, 0, 0);
#endif
    };
    auto issue = [&](int tile, int slot) {               // all pieces of a tile at once (prologue)
        unsigned doff[LDN];
        dma_offsets(tile, doff);
        static_for<LDN>([&](auto pc) { dma_piece(pc, slot, doff); });
    };

    // ---- the stationary operand: this wave's weight slice(s) and biases.  All workgroups of a channel group want the same lines at
    // the same moment: four start offsets (by pixel group) spread them over the L2 channels (f8_p12.hip: K-order rotation)
    v4i w0[NK0], w1[DUAL ? NK1 : 1];
    {
        const v4i* const wp = (const v4i*)a.w + (size_t)ct * NK0 * 64 + lane;
        const v4i* const wq = (const v4i*)(DUAL ? a.w2 : a.w) + (size_t)ct * (DUAL ? NK1 : NK0) * 64 + lane;
        auto load_w = [&](auto rc) {
            constexpr int R = decltype(rc)::value;
#pragma unroll
            for (int k = 0; k < NK0; ++k) { constexpr int dummy = 0; (void)dummy; const int kk = (k + R * (NK0 / 4)) % NK0; w0[kk] = wp[(size_t)kk * 64]; }
            if constexpr (DUAL) {
#pragma unroll
                for (int k = 0; k < NK1; ++k) { const int kk = (k + R * (NK1 / 4)) % NK1; w1[kk] = wq[(size_t)kk * 64]; }
            }
        };
        switch (mg & 3) {
            case 0: load_w(std::integral_constant<int, 0>{}); break;
            case 1: load_w(std::integral_constant<int, 1>{}); break;
            case 2: load_w(std::integral_constant<int, 2>{}); break;
            default: load_w(std::integral_constant<int, 3>{}); break;
        }
    }
    v16i b0, b1;                                         // biases in accumulator layout: the C operand of a chain's first MFMA
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const v4i t0v = *(const v4i*)(a.bias + ct * 32 + 8 * g + 4 * lh);
        const v4i t1v = DUAL ? *(const v4i*)(a.bias2 + ct * 32 + 8 * g + 4 * lh) : t0v;
#pragma unroll
        for (int e = 0; e < 4; ++e) { b0[4 * g + e] = t0v[e]; b1[4 * g + e] = t1v[e]; }
    }
    // fragment addresses: byte (32 k + 16 lh) ^ (f << 4) of row l31, f = l31 % 16 (Swz): the XOR touches bits 4..7 only, so with
    // k = 8 kh + kl the address is [row base + ((16 lh) ^ (f0 << 4))] + [(32 kl) ^ (fh << 4)] + 256 kh — eight per-lane values per
    // input for the whole kernel, 256 kh as the instruction's immediate offset (instead of an XOR and an add per fragment)
    unsigned fa0[8], fa1[DUAL ? 8 : 1];
    {
        const unsigned f = (unsigned)(l31 & 15), f0 = f & 1u, fh = (f & ~1u) << 4;
#pragma unroll
        for (int kl = 0; kl < 8; ++kl) {
            fa0[kl] = (unsigned)(l31 * K0) + (((unsigned)lh ^ f0) << 4) + ((32u * kl) ^ fh);
            if constexpr (DUAL) fa1[kl] = (unsigned)(32 * K0 + l31 * K1) + (((unsigned)lh ^ f0) << 4) + ((32u * kl) ^ fh);
        }
    }
#ifdef F8_ABL_WS_NOSTORE
    constexpr int nst = 0;
#else
    constexpr int nst = (OUT32 ? 4 : 0) + NQ;          // stores per thread and tile: the outputs are template parameters so that the
                                                       // epilogue is branch-free and shares a basic block with the next tile's MFMAs
#endif
#ifdef F8_ABL_WS_NODMA
    constexpr int NRES = RES ? 4 : 0, LD = 0;
#else
    constexpr int NRES = RES ? 4 : 0, LD = L0 + L1;
#endif
    const int floor0 = a.relu0 ? 0 : INT32_MIN, floor1 = a.relu1 ? 0 : -2147483647;      // the join's clamp_(min) and its ReLU are one max

    // int32 byte offset of this lane's accumulator group 0 of a tile (I32T): ((tile ctiles + ct) 1024 + lh 128 + l31 4) ints; + g KB
    auto i32off = [&](int tile) -> unsigned { return (((unsigned)tile * (unsigned)ctiles + (unsigned)ct) * 1024u + (unsigned)(lh * 128 + l31 * 4)) * 4u; };
    auto load_res = [&](int tile, v4i (&rr)[RES ? 4 : 1]) {
        if constexpr (RES) {
#pragma unroll
            for (int g = 0; g < 4; ++g) rr[g] = __builtin_amdgcn_raw_buffer_load_b128(rres, i32off(tile) + g * 1024u, 0, 0);
        }
    };
    // One pass over the NKT MFMA slots of a tile.  DO_MMA: the K loop of the tile in `slot` into nA / nB (biases ride in the start
    // values: wrapping adds commute); DO_EPI: the epilogue of the PREVIOUS tile (`etile`, accumulators pA / pB, residual prr), cut
    // into 16 per-output steps that are dealt out behind the MFMAs.  A scheduling barrier closes every slot, so the instruction
    // stream is: fragment read (two slots ahead) | MFMA | a share of the epilogue's VALU work | ... — the matrix pipe and the VALU
    // of one wave overlap.  (Left alone the scheduler issues all MFMAs first and the epilogue after them; sched_group_barrier
    // pipelines were ignored in this block.)
    constexpr int NKT = NK0 + NK1;
    auto pass = [&](auto do_mma, auto do_epi, int slot, int etile, v16i& nA, v16i& nB, const v16i& pA, const v16i& pB, const v4i (&prr)[RES ? 4 : 1], int fillslot, const unsigned* doff) {
        constexpr bool DO_MMA = decltype(do_mma)::value, DO_EPI = decltype(do_epi)::value;
        const char* const xb = smem + slot * STAGE;
        // MFMA slot k -> (which conv, K step): non-dual: step k into chain k & 1; dual: A0 B0 A1 B1 ... then the longer one's rest
        auto which = [](int k) constexpr -> int { constexpr int m = NK0 < NK1 ? NK0 : NK1; return !DUAL ? 0 : (k < 2 * m ? (k & 1) : (NK0 > NK1 ? 0 : 1)); };
        auto kstep = [](int k) constexpr -> int { constexpr int m = NK0 < NK1 ? NK0 : NK1; return !DUAL ? k : (k < 2 * m ? k / 2 : k - m); };
        auto frag = [&](auto kc) -> v4i {
            constexpr int k = decltype(kc)::value;
            constexpr int ks = kstep(k);
            if constexpr (which(k) == 0) return *(const v4i*)(xb + fa0[ks & 7] + 256 * (ks >> 3));
            else return *(const v4i*)(xb + fa1[DUAL ? (ks & 7) : 0] + 256 * (ks >> 3));
        };
        constexpr int PF = F8_WS_PF;                        // fragment reads in flight ahead of the MFMA that consumes them
        v4i xf[PF + 1];
        if constexpr (DO_MMA) {
            static_for<PF>([&](auto jc) { constexpr int j = decltype(jc)::value; if constexpr (j < NKT) xf[j] = frag(std::integral_constant<int, (j < NKT ? j : 0)>{}); });
        }
        const int m = etile * 32 + l31;
        const unsigned o32 = i32off(etile);
        int y[4][4];
        int rq[NQ ? NQ : 1][4];
        unsigned d[NQ ? NQ : 1][4];
        static_for<NKT>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (DO_MMA) {
                // the DMAs of the tile S-1 ahead ride between the MFMAs too: issuing one costs the wave 100+ cycles (a burst of
                // them after the barrier: ~500 idle cycles per tile on every SIMD)
                static_for<LDN>([&](auto pc) { if constexpr (decltype(pc)::value * NKT / LDN == k) dma_piece(pc, fillslot, doff); });
                if constexpr (k + PF < NKT) xf[(k + PF) % (PF + 1)] = frag(std::integral_constant<int, (k + PF < NKT ? k + PF : 0)>{});
                constexpr int ks = kstep(k);
#ifdef F8_ABL_WS_NOMMA
                nA[0] += xf[k % (PF + 1)][0] + w0[ks % NK0][0];
#else
                // a chain's first MFMA takes the bias vector as its C operand (wrapping adds commute: no accumulator initialisation)
                if constexpr (which(k) == 0) nA = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0[ks], xf[k % (PF + 1)], ks == 0 ? b0 : nA, 0, 0, 0);
                else                         nB = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1[DUAL ? ks : 0], xf[k % (PF + 1)], ks == 0 ? b1 : nB, 0, 0, 0);
#endif
            }
#ifdef F8_ABL_WS_NOEPI
            if constexpr (false) {
#else
            if constexpr (DO_EPI) {
#endif
                constexpr int T0 = k * 16 / NKT, T1 = (k + 1) * 16 / NKT;
                static_for<T1 - T0>([&](auto jc) {
                    constexpr int t = T0 + decltype(jc)::value, g = t / 4, e = t % 4;
                    int v = FAST ? pA[t] : max(pA[t], floor0);
                    if constexpr (DUAL) v = max((int)(((unsigned)v << a.acc_shl) + ((unsigned)pB[t] << a.res_shl)), floor1);
                    if constexpr (RES) v = max((int)(((unsigned)v << a.acc_shl) + ((unsigned)prr[g][e] << a.res_shl)), floor1);
                    y[g][e] = v;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        if constexpr (FAST) rq[q][e] = requant_shr(v, a.q[q].n, 1u << (a.q[q].n - 1), 0u, (a.relu0 && a.q[q].lo < 0) ? 0 : a.q[q].lo, a.q[q].hi);
                        else rq[q][e] = requant1(v, a.q[q].n, a.q[q].lo, a.q[q].hi);
                    }
                    if constexpr (e == 3) {
#pragma unroll
                        for (int q = 0; q < NQ; ++q) d[q][g] = pack4(rq[q][0], rq[q][1], rq[q][2], rq[q][3]) ^ a.q[q].bias_xor;
                        if constexpr (OUT32) {               // I32T rows are padded to 32: the whole tile is written
                            const v4i o = {y[g][0], y[g][1], y[g][2], y[g][3]};
#ifndef F8_ABL_WS_NOSTORE
                            __builtin_amdgcn_raw_buffer_store_b128(o, ro32, o32 + g * 1024u, 0, 0);
#else
                            if (o[0] == 0x12345678 && o[1] == 0x7654321) __builtin_amdgcn_raw_buffer_store_b128(o, ro32, o32 + g * 1024u, 0, 0);
#endif
                        }
                    }
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
#ifdef F8_ABL_WS_NOEPI
        if constexpr (false) {
#else
        if constexpr (DO_EPI) {
#endif
            const unsigned rowo = m < a.M ? (unsigned)m * (unsigned)a.coutP + (unsigned)(ct * 32 + 16 * lh) : kOOB;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                auto s0 = __builtin_amdgcn_permlane32_swap(d[q][0], d[q][2], false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(d[q][1], d[q][3], false, false);
                const v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
#ifndef F8_ABL_WS_NOSTORE
                __builtin_amdgcn_raw_buffer_store_b128(o, q ? rq1 : rq0, rowo, 0, 0);
#else
                if (o[0] == 0x12345678 && o[1] == 0x7654321) __builtin_amdgcn_raw_buffer_store_b128(o, q ? rq1 : rq0, rowo, 0, 0);
#endif
            }
        }
    };
    using yes = std::true_type; using no = std::false_type;

#pragma unroll
    for (int i = 0; i < S - 1; ++i) issue(i < nt ? t0 + i : T, i);
    __builtin_amdgcn_sched_barrier(0);
#ifdef F8_TRACE
    tta = __builtin_readcyclecounter(); ttl = tta;
#endif

    // The loop is software-pipelined by one tile.  Per iteration i >= 1, in program order:
    // residual loads (i) | wait + barrier | { MFMAs (i), DMAs (i+S-1), epilogue + stores (i-1) } interleaved.
    // VMEM operations per iteration: c0 = NRES + LD in iteration 0 (no epilogue yet), c1 = NRES + LD + nst afterwards.  Tile i's DMAs
    // were issued in iteration i-S+1 (or the prologue): the wait lets the operations of the iterations after that one (and this
    // iteration's residual loads) stay in flight.
    v16i pA, pB, nA, nB;
    v4i prr[RES ? 4 : 1];
    unsigned doff[LDN];
    load_res(t0, prr);
    dma_offsets(S - 1 < nt ? t0 + S - 1 : T, doff);
    __builtin_amdgcn_sched_barrier(0);
    wait_vmcnt_dyn((S - 2) * LD + NRES);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    pass(yes{}, no{}, 0, 0, pA, pB, pA, pB, prr, S - 1, doff);
    int slot = 1, fill = 0;
    constexpr int C0 = NRES + LD, C1 = NRES + LD + nst;
#pragma unroll 1
    for (int i = 1; i < nt; ++i) {
        const int tile = t0 + i;
        v4i nrr[RES ? 4 : 1];
        load_res(tile, nrr);
        dma_offsets(i + S - 1 < nt ? tile + S - 1 : T, doff);
        __builtin_amdgcn_sched_barrier(0);
#ifndef F8_ABL_WS_NOWAIT
        if (i >= S - 1) wait_vmcnt<(NRES + (S - 2) * C1 < 48 ? NRES + (S - 2) * C1 : 0)>();      // steady state: a constant
        else wait_vmcnt_dyn(NRES + (S - 2 - i) * LD + C0 + (i - 1) * C1);
#endif
        F8_WT(ttv);
#ifndef F8_ABL_WS_NOBAR
        __builtin_amdgcn_s_barrier();                    // everybody's rows landed; everybody is done with tile i-1: its slot is free
#endif
        __builtin_amdgcn_sched_barrier(0);
        F8_WT(ttb);
        pass(yes{}, yes{}, slot, tile - 1, nA, nB, pA, pB, prr, fill, doff);
        pA = nA;
        if constexpr (DUAL) pB = nB;
        if constexpr (RES) {
#pragma unroll
            for (int g = 0; g < 4; ++g) prr[g] = nrr[g];
        }
        slot = slot + 1 == S ? 0 : slot + 1;
        fill = fill + 1 == S ? 0 : fill + 1;
        F8_WT(ttc);
    }
    pass(no{}, yes{}, 0, t0 + nt - 1, nA, nB, pA, pB, prr, 0, doff);
    F8_WT(tte);
#ifdef F8_TRACE
    if (a.trace && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* tp = (unsigned long long*)a.trace + (size_t)blockIdx.x * 8;
        const unsigned long long end = __builtin_readcyclecounter();
        tp[0] = ttl - tt0; tp[1] = ttw; tp[2] = ttc; tp[3] = tte; tp[4] = end - tta; tp[5] = (unsigned long long)nt; tp[6] = end - tt0; tp[7] = (ttv << 32) | (ttb & 0xffffffffull);
    }
#endif
}

// ---- host side
// (K0, K1, has_res): the instances.  K1 > 0: dual GEMM (the join's other operand is a second 1x1 conv over the same output pixels).
bool conv1x1_wstat_supported(int k0, int k1, int coutP, bool has_res) {
    if (k1 == 0 && !has_res) return (k0 == 512 || k0 == 1024) && coutP % 256 == 0;
    if (k1 == 0 && has_res) return k0 == 512 && coutP % 256 == 0;
    if (has_res) return false;
    if (k0 == 512 && k1 == 256) return coutP % 256 == 0;      // host = the strided shortcut, other operand = body.4 (ResNet-50 stage 2)
    if (k0 == 1024 && k1 == 512) return coutP % 128 == 0;     // ... stage 3, on 4 waves (192 VGPRs of weights per wave)
    return false;
}
int conv1x1_wstat_waves(int k0, int k1) { return k0 + k1 > 1024 ? 4 : 8; }

template <int K0, int K1, int NW, bool RES, bool OUT32, int NQ, bool FAST>
static hipError_t launch_wstat_t(const ConvArgs& a, int num_cu, hipStream_t s) {
    using Cfg = WstatCfg<K0, K1, NW>;
    static unsigned long long attr_done = 0; int attr_dev = -1;
    if (!dyn_lds_opted_in(&attr_done, &attr_dev)) {
        hipError_t e = hipFuncSetAttribute((const void*)conv1x1_wstat_kernel<K0, K1, NW, RES, OUT32, NQ, FAST>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        if (attr_dev >= 0) attr_done |= 1ull << attr_dev;
    }
    const int T = (a.M + 31) / 32, NG = a.coutP / (32 * NW);
    int MG = num_cu * Cfg::WG_PER_CU / NG; if (MG < 1) MG = 1; if (MG > T) MG = T;
    const int grid = (MG + 7) / 8 * 8 * NG;
    // bit 0 / 1: the rows of x / x2 are the output pixels in order (stride 1): row m starts at m K, no (n, p, q) decoding
    const int dense0 = ((a.sQ == K0 && a.sP == a.Q * K0 && a.sN == a.PQ * K0) ? 1 : 0) |
                       ((K1 > 0 && a.sQ2 == K1 && a.sP2 == a.Q * K1 && a.sN2 == a.PQ * K1) ? 2 : 0);
#ifdef F8_TRACE
    static unsigned long long* tbuf = nullptr; static int count = 0;
    static const int want = [] { const char* e = getenv("F8_TRACE_WSTAT"); return e ? atoi(e) : -1; }();
    ConvArgs b = a;
    const bool tracing = (count++ == want);
    if (tracing) { if (!tbuf) (void)hipMalloc((void**)&tbuf, (size_t)1 << 20); (void)hipMemsetAsync(tbuf, 0, (size_t)grid * 64, s); b.trace = tbuf; }
    hipLaunchKernelGGL((conv1x1_wstat_kernel<K0, K1, NW, RES, OUT32, NQ, FAST>), dim3(grid), dim3(NW * 64), Cfg::LDS_BYTES, s, b, T, MG, NG, dense0);
    if (tracing) {
        (void)hipStreamSynchronize(s);
        unsigned long long* hb = new unsigned long long[(size_t)grid * 8];
        (void)hipMemcpy(hb, tbuf, (size_t)grid * 64, hipMemcpyDeviceToHost);
        double ph[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; int n = 0;
        for (int i = 0; i < grid; ++i) { unsigned long long* p = hb + (size_t)i * 8; if (!p[6]) continue; ++n; for (int k = 0; k < 7; ++k) ph[k] += (double)p[k]; ph[7] += (double)(p[7] >> 32); ph[8] += (double)(p[7] & 0xffffffffull); }
        fprintf(stderr, "[trace wstat<%d,%d,%d,%d,%d,%d,%d>] grid %d T %d MG %d NG %d: avg cycles per WG: prologue %.0f | per tile (%.2f tiles): vmcnt %.0f barrier %.0f (unused %.0f) pass %.0f  (last epilogue %.0f) | drain %.0f | total %.0f\n",
                K0, K1, NW, (int)RES, (int)OUT32, NQ, (int)FAST, grid, T, MG, NG, ph[0] / n, ph[5] / n, ph[7] / ph[5], ph[8] / ph[5], ph[1] / ph[5], ph[2] / ph[5], ph[3] / ph[5], ph[4] / n, ph[6] / n);
        delete[] hb;
    }
    return hipGetLastError();
#else
    hipLaunchKernelGGL((conv1x1_wstat_kernel<K0, K1, NW, RES, OUT32, NQ, FAST>), dim3(grid), dim3(NW * 64), Cfg::LDS_BYTES, s, a, T, MG, NG, dense0);
    return hipGetLastError();
#endif
}

template <int K0, int K1, int NW, bool RES, bool FAST>
static hipError_t launch_wstat_o(const ConvArgs& a, int num_cu, hipStream_t s) {
    // the int8 outputs are q[0 .. nq-1] (fill_out fills them in order)
    const int nq = (a.q[0].ptr ? 1 : 0) + (a.q[1].ptr ? 1 : 0);
    if (a.q[1].ptr && !a.q[0].ptr) return hipErrorInvalidValue;
    if (a.out32) {
        if (nq == 0) return launch_wstat_t<K0, K1, NW, RES, true, 0, FAST>(a, num_cu, s);
        if (nq == 1) return launch_wstat_t<K0, K1, NW, RES, true, 1, FAST>(a, num_cu, s);
        return launch_wstat_t<K0, K1, NW, RES, true, 2, FAST>(a, num_cu, s);
    }
    if (nq == 1) return launch_wstat_t<K0, K1, NW, RES, false, 1, FAST>(a, num_cu, s);
    if (nq == 2) return launch_wstat_t<K0, K1, NW, RES, false, 2, FAST>(a, num_cu, s);
    return hipErrorInvalidValue;
}
// FAST epilogue: every int8 output shifts right (the usual case: 4-operation requantisation) and the ReLU behind the conv is either
// absent or can ride in the clamp's lower bound (no int32 output, no join: requant is monotonic and maps 0 to 0, so
// requant(max(v, 0)) = max(requant(v), 0))
bool conv1x1_wstat_fast(const ConvArgs& a) {
    if (a.no_fast) return false;
    for (int k = 0; k < 2; ++k) if (a.q[k].ptr && a.q[k].n <= 0) return false;
    const bool join = a.res != nullptr || a.x2 != nullptr;
    return !a.relu0 || (!join && !a.out32);
}
template <int K0, int K1, int NW, bool RES>
static hipError_t launch_wstat_f(const ConvArgs& a, int num_cu, hipStream_t s) {
    return conv1x1_wstat_fast(a) ? launch_wstat_o<K0, K1, NW, RES, true>(a, num_cu, s) : launch_wstat_o<K0, K1, NW, RES, false>(a, num_cu, s);
}

hipError_t launch_conv1x1_wstat(const ConvArgs& a, int num_cu, hipStream_t s) {
    const bool res = a.res != nullptr;
    const int k1 = a.x2 ? a.ktot2 : 0;
    if (num_cu <= 0) num_cu = 256;
    if (k1 == 0 && !res && a.CK == 512) return launch_wstat_f<512, 0, 8, false>(a, num_cu, s);
    if (k1 == 0 && !res && a.CK == 1024) return launch_wstat_f<1024, 0, 8, false>(a, num_cu, s);
    if (k1 == 0 && res && a.CK == 512) return launch_wstat_f<512, 0, 8, true>(a, num_cu, s);
    if (a.CK == 512 && k1 == 256) return launch_wstat_f<512, 256, 8, false>(a, num_cu, s);
    if (a.CK == 1024 && k1 == 512) return launch_wstat_f<1024, 512, 4, false>(a, num_cu, s);
    return hipErrorInvalidValue;
}

}  // namespace f8
