// f8_bchain.hip — consecutive ResNet BasicBlocks of one stage in one launch; the int32 residual stream never leaves the chip.
//
// IntBlock.forward of /root/reference/models/fix_resnet.py:26-54,77 for BasicBlock identity blocks (ResNet-18 / 34:
// fix_resnet.py:125-221):   x --3x3, ReLU--> mid --3x3--> + x (aligned, wrapping add, clamp) --ReLU--> y,   every
// int_op_only_fix_quant (fix_quant_ops.py:90-114) in place, applied to every identity block of the stage in turn.
//
// Same idea as f8_chain.hip (the bottleneck stages of ResNet-50): a workgroup owns a tile of R rows x full width of one image and
// keeps the tile's int32 stream in REGISTERS from block to block; HBM sees the stage input once (int32) and the stage output once.
// A BasicBlock has TWO 3x3 convolutions, so a tile swaps halo rows with its vertical neighbours twice per block (the int8 input of
// the first conv, then `mid`), by the same placement-independent protocol (write-through stores, drain, barrier, flag; one lane
// polls; sc0 sc1 loads; tickets; bounded spins).  Unlike the bottleneck chains the stream is small here (64 of 256 registers), which
// leaves room for the overlap those kernels could not afford: the 3x3's K loop starts with its CENTRE-ROW taps, which read no halo
// row, and the neighbours' rows are only waited for in front of the first tap that needs them — the exchange runs under a third of
// the K loop.
//
// 512 threads = 8 waves; wave (ct, pg) computes output channel tile ct for pixel tiles pg, pg + PG, ... (4 of them in every
// instance) in BOTH convs, and owns the same tiles of the stream.  Weights travel L2 -> registers in MFMA-fragment order
// (pack_frag_weights), B operands come from two LDS patches (rows padded by 16 bytes: one base register + immediates), no barrier
// inside a K loop apart from the halo hand-over.
#include "f8_device.h"
#include <cstdio>
#include <cstdlib>

// Waves per workgroup.  8 = two per SIMD, four pixel tiles per wave (the default).  16 (tuning builds) = four per SIMD, two tiles per wave, 128 VGPRs:
// tried in round 4 because the limiter is latency (rocprof_r04_r18_valu.md) — bit-exact and SLOWER on every instance (123 vs 119, 84 vs 78, 73 vs 66 us
// per 128 images, same box): a weight fragment then feeds two MFMAs instead of four, twice the waves meet at every barrier, 8-28 bytes of scratch.
#ifndef F8_BCH_NW
#define F8_BCH_NW 8
#endif

namespace f8 {

template <int C, int W, int H, int R, bool DS = false>
struct BChainCfg {
    static constexpr int T = (H + R - 1) / R;
    static constexpr int PX = R * W, NPT = (PX + 31) / 32;
    static constexpr int PW = W + 2, PR = R + 2, CS = C + 16;          // patch entry stride: padded, see f8_chain.hip
    static constexpr int PATCH_BYTES = (PR * PW * CS + 255) / 256 * 256;
    // stage-opening block (DS): its int8 input at twice the resolution, half the channels: rows 2 p0 - 1 .. 2 (p0 + R) - 1, columns -1 .. 2 W - 1
    static constexpr int CIN = C / 2, PWI = 2 * W + 1, PRI = 2 * R + 1, IS = CIN + 16;
    static constexpr int PATCHI_BYTES = DS ? (PRI * PWI * IS + 255) / 256 * 256 : 0;
    static constexpr int BIAS_BYTES = (kBChainMaxBlocks * 2 + 1) * C * 4;      // per block ba | bb, then the opening block's shortcut bias
    static constexpr int LDS_BYTES = 2 * PATCH_BYTES + PATCHI_BYTES + BIAS_BYTES + 256;
    static constexpr int ROWB = W * C;
    static_assert(LDS_BYTES <= 160 * 1024 && PATCH_BYTES < 65536, "LDS / immediate offsets");
    static_assert(!DS || PX * IS <= PATCH_BYTES, "the shortcut's operand borrows patchX");
    static_assert(ROWB / 16 <= 256 && (size_t)256 * 4 * ROWB <= kChainXchgBytes, "one 16-byte piece of a halo row per thread of a half workgroup");
};

// FAST: 0 generic, 1 float-converter requantisation, 2 integer requantisation (f8_chain.hip: quant_tile16)
template <int FAST, bool ACC = false, class Y>
__device__ __forceinline__ v4i bquant_tile16(const Y& y, int n, int lo, int hi, unsigned x_or) {
    unsigned d[4];
    // FAST == 1: every requantised value is bounded by the planner (BChainArgs::acc_ok, ::stream_ok) -> the 3-operation float form; 2: integer form
    const float sc = FAST == 1 ? requant_u8_scale(n) : 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if constexpr (FAST == 1) d[g] = requant_u8x4(y[4 * g], y[4 * g + 1], y[4 * g + 2], y[4 * g + 3], sc) ^ x_or;
        else if constexpr (FAST == 2) d[g] = requant_u8x4_int(y[4 * g], y[4 * g + 1], y[4 * g + 2], y[4 * g + 3], n) ^ x_or;
        else d[g] = pack4(requant1(y[4 * g], n, lo, hi), requant1(y[4 * g + 1], n, lo, hi), requant1(y[4 * g + 2], n, lo, hi), requant1(y[4 * g + 3], n, lo, hi)) ^ x_or;
    }
    auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
    auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
    const v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
    return o;
}
__device__ __forceinline__ int bopaque(int v) { asm volatile("" : "+s"(v)); return v; }

// FAST: ReLU after the first conv and after the join, every int8 format of the chain unsigned with a right shift, the stream never shifted
// DS: the chain starts with the stage-opening block (3x3 / 2 -> 3x3, 1x1 / 2 shortcut; C / 2 input channels at twice the resolution)
template <int C, int W, int H, int R, int NB, int NBUF, int FAST, bool DS, int NW = 8>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4)))
bchain_kernel(const BChainArgs a) {
    constexpr int NT = NW * 64;                                 // threads: 8 waves (two per SIMD) or 16 (four per SIMD: half the tiles per wave)
    using Cfg = BChainCfg<C, W, H, R, DS>;
    constexpr int T = Cfg::T, NPT = Cfg::NPT, PW = Cfg::PW, CS = Cfg::CS, ROWB = Cfg::ROWB;
    constexpr int CT = C / 32, PG = NW / CT, NPW = (NPT + PG - 1) / PG;
    constexpr int CTI = Cfg::CIN / 32, PWI = Cfg::PWI, IS = Cfg::IS;
    static_assert(CT == 2 || CT == 4 || CT == 8, "8 waves = CT channel tiles x PG pixel-tile groups");
    static_assert(NB <= CT && CT % NB == 0 && (!DS || (NB <= CTI && CTI % NB == 0)), "a batch of K steps stays inside one tap");
    static_assert(NPW <= 4, "stream + accumulators in registers");
    using ic_ct = std::integral_constant<int, CT>;
    using ic_cti = std::integral_constant<int, CTI>;

    if constexpr (FAST == 1) set_fp_round_nearest_even();
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const patchX = lds;                                   // [(R+2)][(W+2)][CS]: int8 copy of the stream in the first conv's input format, with halo
    char* const patchM = lds + Cfg::PATCH_BYTES;                // the same for `mid`
    char* const patchI = lds + 2 * Cfg::PATCH_BYTES;            // DS: [(2R+1)][(2W+1)][IS]: the opening block's int8 input
    int* const bias_lds = (int*)(lds + 2 * Cfg::PATCH_BYTES + Cfg::PATCHI_BYTES);   // per block: ba | bb; then bsc
    int* const misc = bias_lds + Cfg::BIAS_BYTES / 4;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & (NW - 1);
    const int ct = wave & (CT - 1), pg = wave / CT;

    if (tid == 0) { misc[0] = (int)__hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); misc[1] = 0; }
    for (int b = 0; b < a.nblk; ++b) {
        const BChainBlk& B = a.blk[b];
        for (int i = tid; i < 2 * C; i += NT) bias_lds[b * 2 * C + i] = i < C ? B.ba[i] : B.bb[i - C];
    }
    if constexpr (DS) for (int i = tid; i < C; i += NT) bias_lds[kBChainMaxBlocks * 2 * C + i] = a.bsc[i];
    __syncthreads();
    const int L = __builtin_amdgcn_readfirstlane(misc[0]);
    const int grp = L / T, ti = L - grp * T;
    const int p0 = ti * R;
    const int rows = (H - p0) < R ? (H - p0) : R;
    const int npx = rows * W;
    const bool has_up = ti > 0, has_dn = ti < T - 1;
    unsigned* const flags = a.sync + 16;
    const unsigned long long t_limit = (unsigned long long)a.timeout_ticks;
#ifdef F8_TRACE
    unsigned long long tt[8] = {}; unsigned long long t_prev = __builtin_readcyclecounter();
#define F8_BT(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); tt[i] += now_ - t_prev; t_prev = now_; } while (0)
#else
#define F8_BT(i)
#endif

    // per-lane constants, re-derived where they are used (see f8_chain.hip)
#define F8_BLANES                                                                                                   \
    int tq_ = tid; asm volatile("" : "+v"(tq_));                                                                    \
    const int lane = tq_ & 63, l31 = lane & 31, lh = lane >> 5;                                                     \
    int bpix[NPW];                                                                                                  \
    _Pragma("unroll") for (int j = 0; j < NPW; ++j) { const int pt = (pg + PG * j) < NPT ? (pg + PG * j) : NPT - 1; bpix[j] = pt * 32 + l31; } \
    (void)lane; (void)lh; (void)bpix

    v16i res[NPW];                                              // the stream: this wave's channel tile x its pixel tiles (a missing tile repeats the last one)
    v4i wbuf[NBUF][NB];
    unsigned seq = 0;

#ifndef F8_BCH_ABL_NOW
    // buffer loads: scalar step offset in soffset, ONE per-lane offset register (f8_chain.hip: as flat loads the optimiser hoisted 64-bit
    // address pairs per stream and added two 64-bit vector adds per load)
    auto ldw = [](const int8_t* base, int soff, unsigned voff) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7ffffff0, 0x00020000);
        return __builtin_amdgcn_raw_buffer_load_b128(r, voff, bopaque(soff), 0);
    };
#else
    auto ldw = [](const int8_t* base, int soff, unsigned voff) { const int q = (int)(size_t)base + soff + (int)voff; const v4i r = {q, q, q, q}; return r; };
#endif
    auto tap_at = [](int t) constexpr { return t < 3 ? t + 3 : (t < 6 ? t - 3 : t); };       // centre row first: it reads no halo row
    // ctk: K steps per tap (input channels / 32) of the conv the weights belong to
    auto w_load = [&](auto ctk, const int8_t* w, v4i (&dst)[NB], int bi, unsigned wl) {
        constexpr int CTK = decltype(ctk)::value, BPTAP = CTK / NB;
        const int k0 = tap_at(bi / BPTAP) * CTK + (bi % BPTAP) * NB;
#pragma unroll
        for (int s = 0; s < NB; ++s) dst[s] = ldw(w, (k0 + s) * 1024, (unsigned)(ct * 9 * CTK * 1024) + wl);
    };
    auto w_prime = [&](auto ctk, const int8_t* w, unsigned wl) {
        static_for<NBUF - 1>([&](auto bc) { constexpr int Bi = decltype(bc)::value; w_load(ctk, w, wbuf[Bi], Bi, wl); });
    };

    // ---- halo rows of `patch`: publish this tile's first / last row
    auto publish = [&](char* patch) {
        if constexpr (T > 1) {
#ifdef F8_TRACE
            const unsigned long long tp0 = __builtin_readcyclecounter();
#endif
            ++seq;
            constexpr int RCH = ROWB / 16, CPE = C / 16;
            const __amdgpu_buffer_rsrc_t rxc = __builtin_amdgcn_make_buffer_rsrc((void*)a.xchg, 0, (unsigned)kChainXchgBytes, 0x00020000);
            const int side = tid / (NT / 2), idx = tid % (NT / 2);
            const bool mine = idx < RCH && (side == 0 ? has_up : has_dn);
            if (mine) {
                const int col = idx / CPE, c16 = idx % CPE;
                const int ent = (side == 0 ? 1 : rows) * PW + col + 1;
                const v4i v = *(const v4i*)(patch + ent * CS + c16 * 16);
                __builtin_amdgcn_raw_buffer_store_b128(v, rxc, (unsigned)(((L * 2 + (int)(seq & 1u)) * 2 + side) * ROWB + idx * 16), 0, 17);   // sc0 sc1
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flags + L, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef F8_TRACE
            tt[4] += __builtin_readcyclecounter() - tp0;
#endif
        }
    };
    // ... and fetch the neighbours' rows into the patch's halo rows (called inside the K loop, after the centre-row taps)
    auto consume = [&](char* patch) {
        if constexpr (T > 1) {
#ifdef F8_TRACE
            const unsigned long long tc0 = __builtin_readcyclecounter();
#endif
            constexpr int RCH = ROWB / 16, CPE = C / 16;
            int t2 = tid; asm volatile("" : "+v"(t2));
            if ((t2 == 0 && has_up) || (t2 == NT / 2 && has_dn)) {
                unsigned* const f = flags + (t2 == 0 ? L - 1 : L + 1);
                const unsigned long long t0 = wall_clock64();
                bool ok = true;
                // a neighbour that never arrives: the sticky error word is set and the launch runs on WITHOUT waiting any more, here and in
                // every other workgroup (they see the word in their own polls) — no second exit from the block loop (f8_chain.hip)
                while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) < 0) {
                    __builtin_amdgcn_s_sleep(1);
                    if (wall_clock64() - t0 > t_limit) { ok = false; break; }
                    if ((__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 8) == a.epoch) break;   // another tile of THIS run gave up
                }
                if (!ok) {
                    __hip_atomic_store(a.err, (a.epoch << 8) | 0x80u | ((unsigned)seq & 0x3fu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (a.err_host) __hip_atomic_store(a.err_host, (a.epoch << 8) | 0x80u | ((unsigned)seq & 0x3fu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            __syncthreads();
            const int side = t2 / (NT / 2), idx = t2 % (NT / 2);
            if (idx < RCH && (side == 0 ? has_up : has_dn)) {
                const __amdgpu_buffer_rsrc_t rxc = __builtin_amdgcn_make_buffer_rsrc((void*)a.xchg, 0, (unsigned)kChainXchgBytes, 0x00020000);
                const int nb_wg = side == 0 ? L - 1 : L + 1;
                const v4i v = __builtin_amdgcn_raw_buffer_load_b128(rxc, (unsigned)(((nb_wg * 2 + (int)(seq & 1u)) * 2 + (1 - side)) * ROWB + idx * 16), 0, 17);
                const int col = idx / CPE, c16 = idx % CPE;
                *(v4i*)(patch + ((side == 0 ? 0 : rows + 1) * PW + col + 1) * CS + c16 * 16) = v;
            }
#ifdef F8_TRACE
            tt[6] += __builtin_readcyclecounter() - tc0;
#endif
            __syncthreads();
#ifdef F8_TRACE
            tt[5] += __builtin_readcyclecounter() - tc0;
#endif
        }
    };

    // ---- one 3x3: acc[j] (started at the bias) += W . patch taps; the halo rows of `patch` are fetched in front of the first tap that reads one
    //      (opening block's first conv: stride 2 over patchI, no halo rows: the tile read all its input rows itself)
    auto conv3x3 = [&](auto ctk, auto first_ds, char* patch, const int8_t* w, const int* bias, v16i (&acc)[NPW]) {
        constexpr int CTK = decltype(ctk)::value, NK = 9 * CTK, NBAT = NK / NB;
        constexpr bool S2 = decltype(first_ds)::value;
        constexpr int PWp = S2 ? PWI : PW, CSp = S2 ? IS : CS, STR = S2 ? 2 : 1;
        F8_BLANES;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const v4i bv = *(const v4i*)(bias + ct * 32 + 8 * g + 4 * lh);
#pragma unroll
            for (int j = 0; j < NPW; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][4 * g + e] = bv[e];
        }
        unsigned bpb[NPW];
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int oc = bpix[j] < npx ? bpix[j] : npx - 1;                    // padding lanes read a valid pixel, result unused
            const int orow = oc / W, ocol = oc - orow * W;
            bpb[j] = (unsigned)((orow * STR * PWp + ocol * STR) * CSp + lh * 16);
        }
        auto rd = [&](v4i (&xf)[NPW], auto gc) {
            constexpr int G = decltype(gc)::value, TAP = tap_at(G / CTK), CI = G % CTK;
#pragma unroll
            for (int j = 0; j < NPW; ++j) xf[j] = *(const v4i*)(patch + bpb[j] + ((TAP / 3) * PWp + TAP % 3) * CSp + CI * 32);
        };
        const unsigned wl = (unsigned)(lane * 16);
        constexpr int GH = (T > 1 && !S2) ? 3 * CTK : NK + 1;                    // first step of a tap that reads a halo row
        v4i xfa[NPW], xfb[NPW];
        rd(xfa, std::integral_constant<int, 0>{});
        static_for<NK>([&](auto gc) {
            constexpr int G = decltype(gc)::value, Bi = G / NB, S = G % NB;
            if constexpr (S == 0 && Bi + NBUF - 1 < NBAT) w_load(ctk, w, wbuf[(Bi + NBUF - 1) % NBUF], Bi + NBUF - 1, wl);
            v4i (&cur)[NPW] = (G & 1) ? xfb : xfa;
            v4i (&nxt)[NPW] = (G & 1) ? xfa : xfb;
#ifndef F8_BCH_ABL_NOB
            if constexpr (G == GH) { consume(patch); rd(cur, gc); }
            if constexpr (G + 1 < NK && G + 1 != GH) rd(nxt, std::integral_constant<int, G + 1>{});
#else
            if constexpr (G == GH) { consume(patch); }
#endif
#pragma unroll
            for (int j = 0; j < NPW; ++j) asm volatile("" : "+v"(cur[j]));
#pragma unroll
            for (int j = 0; j < NPW; ++j) {
#ifndef F8_BCH_ABL_NOMMA
                acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wbuf[Bi % NBUF][S], cur[j], acc[j], 0, 0, 0);
#else
                asm volatile("" :: "v"(wbuf[Bi % NBUF][S]), "v"(cur[j]));
#endif
            }
        });
    };

    for (int n = grp; n < a.N; n += a.NG) {
        const int m_tile = (n * H + p0) * W;
        const BChainBlk& B0 = a.blk[0];
        auto fill_patchX = [&] {   // patchX <- biased zero (border; the interior is written before it is read)
            const unsigned zq = FAST ? 0x80808080u : a.blk[DS && a.nblk > 1 ? 1 : 0].xorq;
            const v4i zx = {(int)zq, (int)zq, (int)zq, (int)zq};
            for (int o = tid * 16; o < Cfg::PATCH_BYTES; o += NT * 16) *(v4i*)(patchX + o) = zx;
        };
        if constexpr (!DS) fill_patchX();
        if constexpr (!DS) {   // ---- stage input (int32 stream) -> registers; its int8 copy -> patchX interior
            F8_BLANES;
            const __amdgpu_buffer_rsrc_t rxr = __builtin_amdgcn_make_buffer_rsrc((void*)a.xr, 0, (unsigned)(((a.N * H * W + 31) & ~31) * C * 4), 0x00020000);
#pragma unroll
            for (int j = 0; j < NPW; ++j) {
                const int mc = m_tile + (bpix[j] < npx ? bpix[j] : 0);
                const unsigned vo = (unsigned)((mc >> 5) * (C * 128) + lh * 512 + (mc & 31) * 16);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const v4i v = __builtin_amdgcn_raw_buffer_load_b128(rxr, vo + g * 1024, ct * 4096, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) res[j][4 * g + e] = v[e];
                }
            }
            w_prime(ic_ct{}, B0.wa, (unsigned)(lane * 16));
            __syncthreads();                                    // the zero fill is complete
#pragma unroll
            for (int j = 0; j < NPW; ++j) {
                const int pix = bpix[j], pr = pix / W, pc = pix - pr * W;
                const v4i o = bquant_tile16<FAST>(res[j], B0.nq, FAST ? 0 : B0.loq, FAST ? 255 : B0.hiq, FAST ? 0x80808080u : B0.xorq);
                if (pix < npx) *(v4i*)(patchX + ((pr + 1) * PW + pc + 1) * CS + ct * 32 + lh * 16) = o;
            }
            __syncthreads();
            publish(patchX);
        } else {               // ---- opening block: its int8 input rows (NHWC, twice the resolution) -> patchI; left column / rows outside the image <- biased zero
            constexpr int CPI = Cfg::CIN / 16, RCHI = 2 * W * CPI, NPC = Cfg::PRI * RCHI, PER = (NPC + NT - 1) / NT;
            const int8_t* const xin = a.x8in + (size_t)n * (2 * H) * (2 * W) * Cfg::CIN;
            const int r0 = 2 * p0 - 1, nr = 2 * rows + 1;
            const v4i zi = {(int)B0.xorq, (int)B0.xorq, (int)B0.xorq, (int)B0.xorq};
            int t3 = tid; asm volatile("" : "+v"(t3));
            v4i v[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int idx = t3 + k * NT, pr = idx / RCHI, pc = idx - pr * RCHI;
                const int ir = r0 + pr;
                v[k] = zi;
                if (idx < NPC && pr < nr && ir >= 0 && ir < 2 * H) v[k] = *(const v4i*)(xin + (size_t)ir * (2 * W * Cfg::CIN) + pc * 16);
            }
            w_prime(ic_cti{}, B0.wa, (unsigned)((t3 & 63) * 16));
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int idx = t3 + k * NT, pr = idx / RCHI, pc = idx - pr * RCHI;
                if (idx < NPC) *(v4i*)(patchI + (pr * PWI + pc / CPI + 1) * IS + (pc % CPI) * 16) = v[k];
            }
            for (int idx = t3; idx < Cfg::PRI * CPI; idx += NT) *(v4i*)(patchI + ((idx / CPI) * PWI) * IS + (idx % CPI) * 16) = zi;
            // the shortcut's operand: the pixels (2r, 2c) of the block input in the SHORTCUT's int8 format -> [R][W][IS] where patchX will be
            const int8_t* const xsc = a.x8sc + (size_t)n * (2 * H) * (2 * W) * Cfg::CIN;
            for (int idx = t3; idx < npx * CPI; idx += NT) {
                const int px = idx / CPI, c16 = idx - px * CPI, pr = px / W, pc = px - pr * W;
                *(v4i*)(patchX + px * IS + c16 * 16) = *(const v4i*)(xsc + ((size_t)(2 * (p0 + pr)) * (2 * W) + 2 * pc) * Cfg::CIN + c16 * 16);
            }
            __syncthreads();
        }
        F8_BT(0);

        for (int b = 0; b < a.nblk; ++b) {
            const BChainBlk& B = a.blk[b];
            const bool last = b + 1 == a.nblk;
            const BChainBlk& BN = a.blk[last ? b : b + 1];
            const int n1 = B.n1, acc_shl = B.acc_shl, res_shl = B.res_shl;
            const int lo1 = FAST ? 0 : B.lo1, hi1 = FAST ? 255 : B.hi1;
            const unsigned xor1 = FAST ? 0x80808080u : B.xor1;
            const int relu_a = FAST ? 1 : B.relu_a, relu1 = FAST ? 1 : B.relu1;
            const int nq = last ? a.q[0].n : BN.nq;
            const int loq = FAST ? 0 : (last ? a.q[0].lo : BN.loq), hiq = FAST ? 255 : (last ? a.q[0].hi : BN.hiq);
            const unsigned xorq = FAST ? 0x80808080u : (last ? a.q[0].bias_xor : BN.xorq);
            const int* const bl = bias_lds + b * 2 * C;

            {   // ============ first conv: mid = requant(relu(conv3x3(x8) + ba)) -> patchM interior
                {   // patchM <- biased zero (border, halo rows outside the image); nobody reads it before the barriers of this conv
                    const v4i zm = {(int)xor1, (int)xor1, (int)xor1, (int)xor1};
                    for (int o = tid * 16; o < Cfg::PATCH_BYTES; o += NT * 16) *(v4i*)(patchM + o) = zm;
                }
                v16i acc[NPW];
                bool opened = false;
                if constexpr (DS) {
                    if (b == 0) {
                        // the opening block: 3x3 / 2 over patchI, then the 1x1 / 2 shortcut INTO THE STREAM
                        opened = true;
                        v4i wsc[CTI];
                        { F8_BLANES;
#pragma unroll
                          for (int k = 0; k < CTI; ++k) wsc[k] = ldw(a.wsc, k * 1024, (unsigned)(ct * CTI * 1024) + (unsigned)(lane * 16)); }
                        conv3x3(ic_cti{}, std::true_type{}, patchI, B.wa, bl, acc);
                        F8_BLANES;
                        w_prime(ic_ct{}, B.wb, (unsigned)(lane * 16));
                        const int* const bsc = bias_lds + kBChainMaxBlocks * 2 * C;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const v4i bv = *(const v4i*)(bsc + ct * 32 + 8 * g + 4 * lh);
#pragma unroll
                            for (int j = 0; j < NPW; ++j)
#pragma unroll
                                for (int e = 0; e < 4; ++e) res[j][4 * g + e] = bv[e];
                        }
#pragma unroll
                        for (int k = 0; k < CTI; ++k)
#pragma unroll
                            for (int j = 0; j < NPW; ++j) {
                                const int oc = bpix[j] < npx ? bpix[j] : npx - 1;
                                const v4i xf = *(const v4i*)(patchX + oc * IS + lh * 16 + k * 32);
                                res[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wsc[k], xf, res[j], 0, 0, 0);
                            }
                    }
                }
                if (!opened) {
                    conv3x3(ic_ct{}, std::false_type{}, patchX, B.wa, bl, acc);
                    F8_BLANES; w_prime(ic_ct{}, B.wb, (unsigned)(lane * 16));
                }
                { F8_BLANES;
                __syncthreads();                                // the zero fill is complete (T == 1: there was no hand-over barrier)
                if constexpr (DS) { if (b == 0) fill_patchX(); } // the shortcut's operand has been read by every wave; the barrier in front of publish() follows
                const int floor0 = relu_a ? 0 : INT32_MIN;
#pragma unroll
                for (int j = 0; j < NPW; ++j) {
                    const int pix = bpix[j], pr = pix / W, pc = pix - pr * W;
                    if constexpr (!FAST) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[j][r] = max(acc[j][r], floor0);
                    }
                    const v4i o = bquant_tile16<FAST, true>(acc[j], n1, lo1, hi1, xor1);
                    if (pix < npx) *(v4i*)(patchM + ((pr + 1) * PW + pc + 1) * CS + ct * 32 + lh * 16) = o;
                } }
            }
            __syncthreads();
            publish(patchM);
            F8_BT(1);

            {   // ============ second conv + join: stream' = clamp((conv3x3(mid) + bb) << sa + (stream << sr)) [ReLU]; x8' = requant(stream') -> patchX
                v16i acc[NPW];
                conv3x3(ic_ct{}, std::false_type{}, patchM, B.wb, bl + C, acc);
                F8_BLANES;
                if (!last) w_prime(ic_ct{}, BN.wa, (unsigned)(lane * 16));
                const int floor1 = relu1 ? 0 : -2147483647;
#pragma unroll
                for (int j = 0; j < NPW; ++j) {
                    const int pix = bpix[j], pr = pix / W, pc = pix - pr * W;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if constexpr (FAST && !DS) res[j][r] = max((int)(((unsigned)acc[j][r] << acc_shl) + (unsigned)res[j][r]), 0);
                        else res[j][r] = max((int)(((unsigned)acc[j][r] << acc_shl) + ((unsigned)res[j][r] << res_shl)), FAST ? 0 : floor1);
                    }
                    if (!last || a.q[0].ptr) {
                        const v4i o = bquant_tile16<FAST>(res[j], nq, loq, hiq, xorq);
                        if (pix < npx) *(v4i*)(patchX + ((pr + 1) * PW + pc + 1) * CS + ct * 32 + lh * 16) = o;
                    }
                    if (last && pix < npx) {
                        const int m = bopaque(m_tile) + pix;
                        const unsigned tot = (unsigned)(((a.N * H * W + 31) & ~31) * C);
                        if (a.out32) {
                            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)a.out32, 0, tot * 4u, 0x00020000);
                            const unsigned vo = (unsigned)((m >> 5) * (C * 128) + lh * 512 + (m & 31) * 16);
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const v4i o = {res[j][4 * g], res[j][4 * g + 1], res[j][4 * g + 2], res[j][4 * g + 3]};
                                __builtin_amdgcn_raw_buffer_store_b128(o, ro, vo + g * 1024, ct * 4096, 0);
                            }
                        }
                        if (a.q[1].ptr) {
                            const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)a.q[1].ptr, 0, tot, 0x00020000);
                            __builtin_amdgcn_raw_buffer_store_b128(bquant_tile16<false>(res[j], a.q[1].n, a.q[1].lo, a.q[1].hi, a.q[1].bias_xor), rq,
                                                                   (unsigned)(m * C + 16 * lh), ct * 32, 0);
                        }
                    }
                }
            }
            __syncthreads();                                    // patchX interior is complete; patchM may be zeroed again
            if (!last) publish(patchX);
            F8_BT(2);
        }

        // ---- the first int8 form of the stage output: patchX interior -> whole NHWC rows in HBM
        if (a.q[0].ptr) {
            constexpr int CH = C / 16;
            for (int idx = tid; idx < npx * CH; idx += NT) {
                const int px = idx / CH, c16 = idx % CH;
                const int pr = px / W, pc = px - pr * W;
                const v4i v = *(const v4i*)(patchX + ((pr + 1) * PW + pc + 1) * CS + c16 * 16);
                *(v4i*)(a.q[0].ptr + (size_t)(m_tile + px) * C + c16 * 16) = v;
            }
        }
        __syncthreads();
        F8_BT(3);
    }
    // ---- re-arm the ticket and the flags for the NEXT launch on this scratch (f8_chain.hip).  A workgroup counts itself out once ITS flag stores have been performed (lane 0 issued
    //      them: its vmcnt(0)) and its last poll has returned; the last one out sees every other workgroup past its last access of the words
    //      and zeroes them; the kernel boundary orders the zeroes before the next launch.  Every workgroup gets here — a timed-out wait sets the
    //      error word and runs on — and the words are zeroed once at allocation (f8_net.cpp), so the first launch starts clean.
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        misc[2] = (__hip_atomic_fetch_add(a.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) ? 1 : 0;
    }
    __syncthreads();
    if (misc[2]) {
        if (tid < (int)gridDim.x) __hip_atomic_store(flags + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) { __hip_atomic_store(a.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(a.sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
#ifdef F8_TRACE
    if (a.trace && tid == 0) {
        unsigned long long* tp = (unsigned long long*)a.trace + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 8; ++i) tp[i] = tt[i];
    }
#endif
}

// instances: ResNet-18 / 34 stages 0-2 (the 7x7 stage streams 4.7 MB of weights per block and image: no instance)
static int bchain_rows(int C, int H, int W) {
    if (C == 64 && H == 56 && W == 56) return 8;
    if (C == 128 && H == 28 && W == 28) return 7;
    if (C == 256 && H == 14 && W == 14) return 7;
    return 0;
}
bool bchain_supported(int C, int H, int W) { return bchain_rows(C, H, W) > 0; }
// ... starting with the stage-opening block (input 2H x 2W x C/2): the 28x28 and 14x14 stages
bool bchain_ds_supported(int C, int H, int W) { return bchain_rows(C, H, W) > 0 && C >= 128; }
int bchain_tiles_per_img(int C, int H, int W) { const int r = bchain_rows(C, H, W); return r ? (H + r - 1) / r : 0; }

// 0 = generic instance, 1 = constant formats + float-converter requantisation, 2 = constant formats + integer requantisation (f8_chain.hip: chain_fast)
int bchain_fast(const BChainArgs& a) {
    bool f16 = true;
    for (int k = 0; k < a.nblk; ++k) {
        const BChainBlk& B = a.blk[k];
        if (!(B.relu_a && B.relu1 && B.n1 > 0 && B.n1 <= 30 && B.lo1 == 0)) return 0;
        f16 = f16 && B.n1 <= kRequantU8MaxShift;
        if (k == 0 && a.x8in) continue;                      // opening block: its input arrives as int8, its join shifts either operand
        if (!(B.nq > 0 && B.nq <= 30 && B.loq == 0 && B.res_shl == 0)) return 0;
        f16 = f16 && B.nq <= kRequantU8MaxShift;
    }
    if (a.q[0].ptr) {
        if (!(a.q[0].n > 0 && a.q[0].n <= 30 && a.q[0].lo == 0)) return 0;
        f16 = f16 && a.q[0].n <= kRequantU8MaxShift;
    }
    return (a.rq_int || !a.acc_ok || !a.stream_ok || !f16) ? 2 : 1;
}

template <int C, int W, int H, int R, int NB, int NBUF, int FAST, bool DS>
static hipError_t launch_bchain_t(const BChainArgs& a, hipStream_t s) {
    using Cfg = BChainCfg<C, W, H, R, DS>;
    static unsigned long long attr_done = 0; int attr_dev = -1;
    if (!dyn_lds_opted_in(&attr_done, &attr_dev)) {
        hipError_t e = hipFuncSetAttribute((const void*)bchain_kernel<C, W, H, R, NB, NBUF, FAST, DS, F8_BCH_NW>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        if (attr_dev >= 0) attr_done |= 1ull << attr_dev;
    }
    const int grid = a.NG * Cfg::T;
    if (grid < 1 || grid > 256) return hipErrorInvalidValue;
#ifdef F8_TRACE
    static unsigned long long* tbuf = nullptr; static int count = 0;
    static const int want = [] { const char* e = getenv("F8_TRACE_BCHAIN"); return e ? atoi(e) : -1; }();
    BChainArgs b = a;
    const bool tracing = (count++ == want);
    if (tracing) { if (!tbuf) (void)hipMalloc((void**)&tbuf, (size_t)1 << 16); (void)hipMemsetAsync(tbuf, 0, (size_t)grid * 64, s); b.trace = tbuf; }
    hipLaunchKernelGGL((bchain_kernel<C, W, H, R, NB, NBUF, FAST, DS, F8_BCH_NW>), dim3(grid), dim3(F8_BCH_NW * 64), Cfg::LDS_BYTES, s, b);
    if (tracing) {
        (void)hipStreamSynchronize(s);
        static unsigned long long hb[256 * 8];
        (void)hipMemcpy(hb, tbuf, (size_t)grid * 64, hipMemcpyDeviceToHost);
        double ph[8] = {}; for (int i = 0; i < grid; ++i) for (int k = 0; k < 8; ++k) ph[k] += (double)hb[(size_t)i * 8 + k];
        fprintf(stderr, "[trace bchain<%d,%d>] grid %d, %d blocks, N %d: avg cycles per WG (whole launch): load %.0f | conv A %.0f | conv B + join %.0f | out %.0f || inside: publish %.0f, consume %.0f (of which to the data in LDS %.0f)\n", C, W, grid,
                a.nblk, a.N, ph[0] / grid, ph[1] / grid, ph[2] / grid, ph[3] / grid, ph[4] / grid, ph[5] / grid, ph[6] / grid);
    }
    return hipGetLastError();
#else
    hipLaunchKernelGGL((bchain_kernel<C, W, H, R, NB, NBUF, FAST, DS, F8_BCH_NW>), dim3(grid), dim3(F8_BCH_NW * 64), Cfg::LDS_BYTES, s, a);
    return hipGetLastError();
#endif
}

// K steps per register batch, batches in rotation (NB, NBUF) per instance (tuning builds override)
#ifndef F8_BCH_S0
#define F8_BCH_S0 2, (F8_BCH_NW == 16 ? 2 : 3)
#endif
#ifndef F8_BCH_S1
#define F8_BCH_S1 2, (F8_BCH_NW == 16 ? 2 : 3)
#endif
#ifndef F8_BCH_S2
#define F8_BCH_S2 2, (F8_BCH_NW == 16 ? 2 : 3)
#endif
// rows per tile of the three instances; the (NB, NBUF) pairs as VALUES (the macros may hold expressions)
constexpr int bchain_rows(int C) { return C == 64 ? 8 : 7; }
template <int NB, int NBUF> struct BChPair { static constexpr int nb = NB, nbuf = NBUF; };
// The device symbol launch_bchain starts, as rocprofv3 prints it: built from the same tables the launcher below instantiates with (see chain_kernel_name)
int bchain_kernel_name(char* buf, size_t cap, int C, int H, int W, bool ds, int fast) {
    const int nb = C == 64 ? BChPair<F8_BCH_S0>::nb : (C == 128 ? BChPair<F8_BCH_S1>::nb : BChPair<F8_BCH_S2>::nb);
    const int nbuf = C == 64 ? BChPair<F8_BCH_S0>::nbuf : (C == 128 ? BChPair<F8_BCH_S1>::nbuf : BChPair<F8_BCH_S2>::nbuf);
    return snprintf(buf, cap, "f8::bchain_kernel<%d, %d, %d, %d, %d, %d, %d, %s, %d>", C, W, H, bchain_rows(C), nb, nbuf, fast, ds ? "true" : "false", F8_BCH_NW);
}
hipError_t launch_bchain(const BChainArgs& a, int C, int H, int W, hipStream_t s, char* launched, size_t cap) {
    if (a.nblk < 1 || a.nblk > kBChainMaxBlocks) return hipErrorInvalidValue;
    const int fast = bchain_fast(a);
    const bool ds = a.x8in != nullptr;
    if (launched) bchain_kernel_name(launched, cap, C, H, W, ds, fast);      // the instance that runs (see launch_chain)
    if (ds ? !(a.x8sc && a.wsc && a.bsc && bchain_ds_supported(C, H, W)) : !a.xr) return hipErrorInvalidValue;
#define F8_BCH(...) (fast == 1 ? launch_bchain_t<__VA_ARGS__, 1, false>(a, s) : fast == 2 ? launch_bchain_t<__VA_ARGS__, 2, false>(a, s) : launch_bchain_t<__VA_ARGS__, 0, false>(a, s))
#define F8_BCHD(...) (fast == 1 ? launch_bchain_t<__VA_ARGS__, 1, true>(a, s) : fast == 2 ? launch_bchain_t<__VA_ARGS__, 2, true>(a, s) : launch_bchain_t<__VA_ARGS__, 0, true>(a, s))
    if (C == 64 && H == 56 && W == 56) return F8_BCH(64, 56, 56, bchain_rows(64), F8_BCH_S0);
    if (C == 128 && H == 28 && W == 28) return ds ? F8_BCHD(128, 28, 28, bchain_rows(128), F8_BCH_S1) : F8_BCH(128, 28, 28, bchain_rows(128), F8_BCH_S1);
    if (C == 256 && H == 14 && W == 14) return ds ? F8_BCHD(256, 14, 14, bchain_rows(256), F8_BCH_S2) : F8_BCH(256, 14, 14, bchain_rows(256), F8_BCH_S2);
#undef F8_BCH
#undef F8_BCHD
    return hipErrorInvalidValue;
}

}  // namespace f8
