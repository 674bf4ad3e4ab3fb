// f8_chain.hip — ALL consecutive bottleneck blocks of a ResNet stage in one launch; the int32 residual stream never leaves the chip.
//
// IntBlock.forward of /root/reference/models/fix_resnet.py:26-77, applied NBLK times in a row as IntModel.forward does for the
// blocks of one stage (fix_resnet.py:361-366), every int_op_only_fix_quant (fix_quant_ops.py:90-114) in place.
//
// Why: with one launch per block (f8_fused.hip) a block is bound by its int32 residual stream — 4 bytes in and 4 bytes out per
// element against 1 + 1 for the int8 activations: 80 % of the bytes of the 56x56 / 28x28 / 14x14 blocks, which together were 60 %
// of ResNet-50's kernel time at 0.44-0.56 of the HBM peak.  The stream only exists BETWEEN blocks.  Here a workgroup owns a tile of
// R rows x full width of one image and keeps the tile's int32 stream in REGISTERS (NPT x C/8 accumulator registers per lane: 112-128
// of the 256 a wave has) across all the blocks of the stage; per block it reads nothing but weights (L2-resident, streamed straight
// into registers in MFMA-fragment order) and writes nothing.  HBM sees the stage input once and the stage output once.
//
//   per block, per tile:   x8 (LDS, int8) --1x1, ReLU--> mid1 (LDS patch) --3x3, ReLU--> mid2 (LDS) --1x1--> + stream (registers)
//                          -> clamp, ReLU -> stream' (registers) -> requant -> x8' (LDS)
//
// The 3x3 needs one row of mid1 above and below the tile: vertically adjacent tiles of an image run on different CUs at the same
// time and swap those rows (3.5 KB each) through global memory once per block:
//   producer: write-through (sc0 sc1) stores -> every storing wave drains -> barrier -> one lane stores the flag (relaxed, agent);
//   consumer: one lane polls the neighbour's flag (relaxed) -> barrier -> sc0 sc1 loads
// (cdna_hip_programming.md Guideline 16, form {sc0 sc1 stores and loads on both sides}); nothing depends on placement or dispatch
// order: a workgroup's place in the grid is a TICKET it draws when it starts, so the set of started workgroups is always a prefix of
// the logical grid, the tiles of one image are consecutive tickets, and a group whose last member has not started yet is the only
// one that waits — on workgroups that start as soon as any other one finishes.  Every spin is bounded (error word, kernel exits).
//
// 512 threads = 8 waves.  P1 / P2: wave (mt, pg) computes mid channel tile mt for pixel tiles pg, pg + PG, ...; P3: wave w owns
// channel tiles [w CT/8, (w+1) CT/8) of the stream for all pixel tiles.  No barrier inside a K loop: B operands are read-only LDS,
// A operands rotate through NBUF register batches of NB K-steps; every (workgroup, wave) walks K in a rotated order (integer sums
// are exact in any order) so that the workgroups do not all ask the L2 for the same kilobyte at the same moment (f8_p12.hip).
#include "f8_device.h"
#include <cstdio>
#include <cstdlib>

namespace f8 {

template <int C, int MID, int W, int H, int R, int CIN0>
struct ChainCfg {
    static constexpr int T = (H + R - 1) / R;                  // tiles (workgroups) per image
    static constexpr int PX = R * W, NPT = (PX + 31) / 32, ROWS = NPT * 32;
    static constexpr int PW = W + 2, PR = R + 2;
    static constexpr int X8_BYTES = ROWS * C;
    static constexpr int PATCH_BYTES = (PR * PW * MID + 255) / 256 * 256;
    static constexpr int MID2_BYTES = ROWS * MID;
    static constexpr int XIN_BYTES = CIN0 != C ? ROWS * CIN0 : 0;
    static constexpr int MISC_BYTES = 256;
    static constexpr int LDS_BYTES = X8_BYTES + PATCH_BYTES + MID2_BYTES + XIN_BYTES + MISC_BYTES;
    static constexpr int ROWB = W * MID;                       // one exchanged row of mid1
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(ROWB / 16 <= 256 && (size_t)256 * 4 * ROWB <= kChainXchgBytes, "one 16-byte piece of a halo row per thread of a half workgroup");
};

// 16 accumulator values of one 32x32 tile (this lane: one pixel, channels 8g + 4 lh + e) -> this lane's 16 bytes of the int8 row:
// channels [16 lh, 16 lh + 16) of the tile (two v_permlane32_swap put a lane's four dwords side by side)
__device__ __forceinline__ v4i quant_tile16(const v16i& y, int n, int lo, int hi, unsigned x_or) {
    unsigned d[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
        d[g] = pack4(requant1(y[4 * g + 0], n, lo, hi), requant1(y[4 * g + 1], n, lo, hi), requant1(y[4 * g + 2], n, lo, hi), requant1(y[4 * g + 3], n, lo, hi)) ^ x_or;
    auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
    auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
    const v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
    return o;
}

template <int C, int MID, int W, int H, int R, int CIN0, int NB, int NBUF>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
chain_kernel(const ChainArgs a) {
    using Cfg = ChainCfg<C, MID, W, H, R, CIN0>;
    constexpr bool DS0 = CIN0 != C;
    constexpr int T = Cfg::T, PX = Cfg::PX, NPT = Cfg::NPT, PW = Cfg::PW, ROWB = Cfg::ROWB;
    constexpr int CT = C / 32, CM = MID / 32, CTW = CT / 8;
    static_assert(CT % 8 == 0 && (CM == 2 || CM == 4 || CM == 8), "8 waves: whole channel tiles per wave in P3, whole pixel-tile groups in P1 / P2");
    constexpr int PG = 8 / CM;                                  // pixel-tile groups in P1 / P2
    constexpr int NPW = (NPT + PG - 1) / PG;                    // pixel tiles per wave there
    constexpr int NK1 = C / 32, NK2 = 9 * CM, KK = CM;          // K32 steps of body.0 / body.2 / body.4
    constexpr int NPAIR = (NPT + 1) / 2;
    static_assert(NB <= CM && CM % NB == 0 && NK1 % NB == 0, "a batch of K steps stays inside one 3x3 tap / one weight tile");
    static_assert(NPT <= 2 || KK == NB, "several pixel pairs per channel tile: the tile's whole K range is one register batch");
    static_assert((NK1 & (NK1 - 1)) == 0 && (KK & (KK - 1)) == 0, "rotation by masking");
    static_assert(!DS0 || (CIN0 / 32 == NB), "stage-opening block: its K range is one batch");
    static_assert(PX <= NPT * 32 && W <= 62, "tile");

    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const x8 = lds;                                       // [NPT*32 px][C] int8, body.0's input format of the NEXT P1
    char* const patch = x8 + Cfg::X8_BYTES;                     // [(R+2)][(W+2)][MID] mid1, border = biased zero
    char* const mid2 = patch + Cfg::PATCH_BYTES;                // [NPT*32 px][MID]
    char* const xin = mid2 + Cfg::MID2_BYTES;                   // DS0: [NPT*32 px][CIN0], the stage input tile
    int* const misc = (int*)(xin + Cfg::XIN_BYTES);
    using SX = Swz<C>;
    using SM = Swz<MID>;
    using SI = Swz<CIN0>;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- place in the logical grid: a ticket
    if (tid == 0) { misc[0] = (int)__hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); misc[1] = 0; }
    __syncthreads();
    const int L = __builtin_amdgcn_readfirstlane(misc[0]);
    const int grp = L / T, ti = L - grp * T;
    const int p0 = ti * R;
    const int rows = (H - p0) < R ? (H - p0) : R;
    const int npx = rows * W;
    const bool has_up = ti > 0, has_dn = ti < T - 1;
    const int rot = __builtin_amdgcn_readfirstlane(L * 5 + wave * 3);

    unsigned* const flags = a.sync + 16;
    const unsigned long long t_limit = (unsigned long long)a.timeout_ticks;
#ifdef F8_TRACE
    unsigned long long tt[8] = {}; unsigned long long t_prev = __builtin_readcyclecounter();
#define F8_CT(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); tt[i] += now_ - t_prev; t_prev = now_; } while (0)
#else
#define F8_CT(i)
#endif

    // P1 / P2 roles
    const int mt = wave & (CM - 1), pg = wave / CM;
    int p12_pt[NPW];
#pragma unroll
    for (int j = 0; j < NPW; ++j) { const int pt = pg + PG * j; p12_pt[j] = pt < NPT ? pt : NPT - 1; }   // a missing tile repeats the last one (same bytes written twice)

    v16i res[NPT][CTW];                                         // the tile's int32 stream: pixel tile x this wave's channel tiles
    unsigned seq = 0;

    for (int n = grp; n < a.N; n += a.NG) {
        const int m_tile = (n * H + p0) * W;                    // global pixel index of the tile's first pixel

        // =====================================================================================
        // stage input -> registers (identity first block) / LDS (stage-opening first block)
        // =====================================================================================
        if constexpr (!DS0) {
            const ChainBlk& B0 = a.blk[0];
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {
                const int pix = pt * 32 + l31;
                const int mc = m_tile + (pix < npx ? pix : 0);
#pragma unroll
                for (int i = 0; i < CTW; ++i) {
                    const int ct = wave * CTW + i;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const v4i v = *(const v4i*)(a.xr + i32t_index(mc, ct * 32 + 8 * g + 4 * lh, C));
#pragma unroll
                        for (int e = 0; e < 4; ++e) res[pt][i][4 * g + e] = v[e];
                    }
                }
            }
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
                for (int i = 0; i < CTW; ++i)
                    *(v4i*)(x8 + SX::off(pt * 32 + l31, (wave * CTW + i) * 2 + lh)) = quant_tile16(res[pt][i], B0.nq, B0.loq, B0.hiq, B0.xorq);
        } else {
            constexpr int CH = CIN0 / 16;                       // 16-byte chunks per pixel
            for (int idx = tid; idx < NPT * 32 * CH; idx += 512) {
                const int row = idx / CH, c16 = idx % CH;
                v4i v = {0, 0, 0, 0};
                if (row < npx) v = *(const v4i*)(a.x8in + (size_t)(m_tile + row) * CIN0 + c16 * 16);
                *(v4i*)(xin + SI::off(row, c16)) = v;
            }
        }
        __syncthreads();
        F8_CT(0);

        for (int b = 0; b < a.nblk; ++b) {
            ++seq;
            const ChainBlk& B = a.blk[b];
            auto block = [&](auto dsc) {
                constexpr bool DSB = decltype(dsc)::value;      // this block is the stage-opening block (first block of a DS0 chain)
                constexpr int NK1B = DSB ? CIN0 / 32 : NK1;
                constexpr int NBAT1 = NK1B / NB;
                const char* const xsrc = DSB ? xin : x8;
                using SXB = std::conditional_t<DSB, SI, SX>;
                constexpr int XROWB = DSB ? CIN0 : C;

                // ============================ P1: mid1 = requant(relu(W0 . x8 + b0)) -> patch interior
                {
                    {   // the whole patch <- biased zero: border columns, rows outside the image; everything else is overwritten below
                        const v4i zv = {(int)B.xor1, (int)B.xor1, (int)B.xor1, (int)B.xor1};
                        for (int o = tid * 16; o < Cfg::PATCH_BYTES; o += 512 * 16) *(v4i*)(patch + o) = zv;
                    }
                    v16i acc[NPW];
                    {
                        v4i bv[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g) bv[g] = *(const v4i*)(B.b0 + mt * 32 + 8 * g + 4 * lh);
#pragma unroll
                        for (int j = 0; j < NPW; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[j][r] = bv[r >> 2][r & 3];
                    }
                    const v4i* const wp = (const v4i*)B.w0 + (size_t)mt * NK1B * 64 + lane;
                    unsigned xrow[NPW], xsw[NPW];
#pragma unroll
                    for (int j = 0; j < NPW; ++j) {
                        const int row = p12_pt[j] * 32 + l31;
                        xrow[j] = (unsigned)(row * XROWB);
                        xsw[j] = (unsigned)((lh ^ SXB::f(row)) << 4);
                    }
                    const int rotk = rot & (NK1B - 1);
                    v4i wbuf[NBUF][NB];
                    auto load_batch = [&](v4i (&dst)[NB], int bi) {
#pragma unroll
                        for (int s = 0; s < NB; ++s) dst[s] = wp[(size_t)((bi * NB + s + rotk) & (NK1B - 1)) * 64];
                    };
                    static_for<(NBUF - 1 < NBAT1 ? NBUF - 1 : NBAT1)>([&](auto bc) { constexpr int Bi = decltype(bc)::value; load_batch(wbuf[Bi], Bi); });
                    static_for<NBAT1>([&](auto bc) {
                        constexpr int Bi = decltype(bc)::value;
                        if constexpr (Bi + NBUF - 1 < NBAT1) load_batch(wbuf[(Bi + NBUF - 1) % NBUF], Bi + NBUF - 1);
                        int kb = Bi * NB + rotk;
                        asm volatile("" : "+s"(kb));                // address arithmetic just in time (hoisted out of the block loop it costs hundreds of registers)
#pragma unroll
                        for (int s = 0; s < NB; ++s) {
                            const unsigned k32 = (unsigned)(((kb + s) & (NK1B - 1)) << 5);   // byte offset of the K step = chunk 2k << 4
#pragma unroll
                            for (int j = 0; j < NPW; ++j) {
                                const v4i xf = *(const v4i*)(xsrc + xrow[j] + (k32 ^ xsw[j]));
                                acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wbuf[Bi % NBUF][s], xf, acc[j], 0, 0, 0);
                            }
                        }
                    });
                    __syncthreads();                            // the zero fill is complete
                    const int floor0 = B.relu_a ? 0 : INT32_MIN;
#pragma unroll
                    for (int j = 0; j < NPW; ++j) {
                        const int pix = p12_pt[j] * 32 + l31;
                        const int pr = pix / W, pc = pix - pr * W;
                        const int ent = (pr + 1) * PW + pc + 1;
                        v16i y;
#pragma unroll
                        for (int r = 0; r < 16; ++r) y[r] = max(acc[j][r], floor0);
                        const v4i o = quant_tile16(y, B.n1, B.lo1, B.hi1, B.xor1);
                        if (pix < npx) *(v4i*)(patch + SM::off(ent, mt * 2 + lh)) = o;
                    }
                }
                F8_CT(1);
                __syncthreads();                                // the patch interior is complete

                // ============================ halo rows: publish mine, fetch the neighbours'
                if constexpr (T > 1) {
                    constexpr int RCH = ROWB / 16, CPE = MID / 16;              // 16-byte pieces per row / per patch entry
                    const __amdgpu_buffer_rsrc_t rxc = __builtin_amdgcn_make_buffer_rsrc((void*)a.xchg, 0, (unsigned)kChainXchgBytes, 0x00020000);
                    const int side = tid >> 8, idx = tid & 255;                 // threads 0..255: top row / upper neighbour; 256..511: bottom / lower
                    const bool mine = idx < RCH && (side == 0 ? has_up : has_dn);
                    const int col = idx / CPE, c16 = idx % CPE;
                    const unsigned par = seq & 1u;
                    if (mine) {
                        const int ent = (side == 0 ? 1 : rows) * PW + col + 1;
                        const v4i v = *(const v4i*)(patch + SM::off(ent, c16));
                        __builtin_amdgcn_raw_buffer_store_b128(v, rxc, (unsigned)(((L * 2 + (int)par) * 2 + side) * ROWB + idx * 16), 0, 17);   // sc0 sc1: write-through
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains
                    __syncthreads();
                    if (tid == 0) __hip_atomic_store(flags + L, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // one lane per neighbour polls its flag
                    if ((tid == 0 && has_up) || (tid == 256 && has_dn)) {
                        unsigned* const f = flags + (tid == 0 ? L - 1 : L + 1);
                        const unsigned long long t0 = wall_clock64();
                        bool ok = true;
                        while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) < 0) {
                            __builtin_amdgcn_s_sleep(4);
                            if (wall_clock64() - t0 > t_limit) { ok = false; break; }
                        }
                        if (!ok) { misc[1] = 1; __hip_atomic_store(a.err, 0x100u + (unsigned)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                    }
                    __syncthreads();
                    if (misc[1]) return false;                                  // a neighbour never arrived: give up (uniform)
                    if (mine) {
                        const int nb_wg = side == 0 ? L - 1 : L + 1;            // upper neighbour's BOTTOM row / lower neighbour's TOP row
                        const v4i v = __builtin_amdgcn_raw_buffer_load_b128(rxc, (unsigned)(((nb_wg * 2 + (int)par) * 2 + (1 - side)) * ROWB + idx * 16), 0, 17);
                        const int ent = (side == 0 ? 0 : rows + 1) * PW + col + 1;
                        *(v4i*)(patch + SM::off(ent, c16)) = v;
                    }
                    __syncthreads();
                }
                F8_CT(2);

                // ============================ P2: mid2 = requant(relu(conv3x3(mid1) + b2)) -> mid2
                {
                    constexpr int NBAT2 = NK2 / NB;
                    v16i acc[NPW];
                    {
                        v4i bv[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g) bv[g] = *(const v4i*)(B.b2 + mt * 32 + 8 * g + 4 * lh);
#pragma unroll
                        for (int j = 0; j < NPW; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[j][r] = bv[r >> 2][r & 3];
                    }
                    int bpx[NPW];                               // patch entry of tap (0, 0) of this lane's output pixel
#pragma unroll
                    for (int j = 0; j < NPW; ++j) {
                        const int pix = p12_pt[j] * 32 + l31, oc = pix < npx ? pix : npx - 1;   // padding lanes read a valid pixel, result unused
                        const int orow = oc / W, ocol = oc - orow * W;
                        bpx[j] = orow * PW + ocol;
                    }
                    const v4i* const wp = (const v4i*)B.w2 + (size_t)mt * NK2 * 64 + lane;
                    const int rotb = (int)((unsigned)rot % (unsigned)NBAT2);
                    auto kof = [&](int bi) { int q = bi + rotb; if (q >= NBAT2) q -= NBAT2; return q * NB; };   // first K step of batch bi
                    v4i wbuf[NBUF][NB];
                    auto load_batch = [&](v4i (&dst)[NB], int bi) {
                        const int k0 = kof(bi);
#pragma unroll
                        for (int s = 0; s < NB; ++s) dst[s] = wp[(size_t)(k0 + s) * 64];
                    };
                    static_for<NBUF - 1>([&](auto bc) { constexpr int Bi = decltype(bc)::value; load_batch(wbuf[Bi], Bi); });
                    static_for<NBAT2>([&](auto bc) {
                        constexpr int Bi = decltype(bc)::value;
                        if constexpr (Bi + NBUF - 1 < NBAT2) load_batch(wbuf[(Bi + NBUF - 1) % NBUF], Bi + NBUF - 1);
                        int k0 = kof(Bi);
                        asm volatile("" : "+s"(k0));
                        const int tap = k0 / CM, c0 = k0 - tap * CM;
                        const int tr = tap / 3, ts = tap - tr * 3;
                        const int eoff = tr * PW + ts;
#pragma unroll
                        for (int j = 0; j < NPW; ++j) {
                            const int ent = bpx[j] + eoff;
                            const unsigned ebase = (unsigned)(ent * MID);
                            const unsigned esw = (unsigned)((lh ^ SM::f(ent)) << 4);
#pragma unroll
                            for (int s = 0; s < NB; ++s) {
                                const v4i xf = *(const v4i*)(patch + ebase + ((unsigned)((c0 + s) << 5) ^ esw));
                                acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wbuf[Bi % NBUF][s], xf, acc[j], 0, 0, 0);
                            }
                        }
                    });
                    const int floor0 = B.relu_b ? 0 : INT32_MIN;
#pragma unroll
                    for (int j = 0; j < NPW; ++j) {
                        v16i y;
#pragma unroll
                        for (int r = 0; r < 16; ++r) y[r] = max(acc[j][r], floor0);
                        *(v4i*)(mid2 + SM::off(p12_pt[j] * 32 + l31, mt * 2 + lh)) = quant_tile16(y, B.n2, B.lo2, B.hi2, B.xor2);
                    }
                }
                F8_CT(3);
                __syncthreads();                                // mid2 is complete; nobody reads the patch any more

                // ============================ P3: stream' = clamp((W4 . mid2 + b4) << sa + (stream << sr)) [ReLU]; x8' = requant(stream')
                {
                    const bool last = b + 1 == a.nblk;
                    // format of the int8 copy in LDS: the next block's body.0 input, or the first int8 form of the stage output
                    const int nq = last ? a.q[0].n : a.blk[last ? b : b + 1].nq, loq = last ? a.q[0].lo : a.blk[last ? b : b + 1].loq;
                    const int hiq = last ? a.q[0].hi : a.blk[last ? b : b + 1].hiq;
                    const unsigned xorq = last ? a.q[0].bias_xor : a.blk[last ? b : b + 1].xorq;
                    const int floor1 = B.relu1 ? 0 : -2147483647;   // the join's clamp_(min=-(2^31-1)) and the ReLU floor are one max
                    constexpr int NBAT4 = KK / NB;
                    constexpr int NQ = CTW * NBAT4;             // batches of this wave's weight stream (its channel tiles are consecutive)
                    const v4i* const wp = (const v4i*)B.w4 + (size_t)(wave * CTW) * KK * 64 + lane;
                    const int rotk = rot & (KK - 1);
                    v4i wbuf[NBUF][NB];
                    auto load_batch = [&](v4i (&dst)[NB], int qi) {            // qi = tile i * NBAT4 + batch
                        const int i = qi / NBAT4, bi = qi - i * NBAT4;
#pragma unroll
                        for (int s = 0; s < NB; ++s) dst[s] = wp[(size_t)(i * KK + ((bi * NB + s + rotk) & (KK - 1))) * 64];
                    };
                    static_for<(NBUF - 1 < NQ ? NBUF - 1 : NQ)>([&](auto bc) { constexpr int Qi = decltype(bc)::value; load_batch(wbuf[Qi], Qi); });
                    v16i acc[2];
                    // one pixel pair of channel tile I is complete: join, clamp, new stream, its int8 copy
                    auto finish = [&](auto ic, auto ppc, const v16i (&acs)[2]) {
                        constexpr int I = decltype(ic)::value, pp = decltype(ppc)::value;
                        const int ct = wave * CTW + I;
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int pt = pp * 2 + jj;
                            if (pt >= NPT) continue;
                            const int pix = pt * 32 + l31;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                unsigned v = (unsigned)acc[jj][r], o;
                                if constexpr (DSB) { o = v; v = (unsigned)acs[jj][r]; }   // the shortcut conv hosts the join (planner's convention)
                                else o = (unsigned)res[pt][I][r];
                                res[pt][I][r] = max((int)((v << B.acc_shl) + (o << B.res_shl)), floor1);
                            }
                            if (!last || a.q[0].ptr) *(v4i*)(x8 + SX::off(pix, ct * 2 + lh)) = quant_tile16(res[pt][I], nq, loq, hiq, xorq);
                            if (last && pix < npx) {
                                const int m = m_tile + pix;
                                if (a.out32) {
#pragma unroll
                                    for (int g = 0; g < 4; ++g) {
                                        const v4i o = {res[pt][I][4 * g], res[pt][I][4 * g + 1], res[pt][I][4 * g + 2], res[pt][I][4 * g + 3]};
                                        *(v4i*)(a.out32 + i32t_index(m, ct * 32 + 8 * g + 4 * lh, C)) = o;
                                    }
                                }
                                if (a.q[1].ptr)
                                    *(v4i*)(a.q[1].ptr + (size_t)m * C + ct * 32 + 16 * lh) = quant_tile16(res[pt][I], a.q[1].n, a.q[1].lo, a.q[1].hi, a.q[1].bias_xor);
                            }
                        }
                    };
                    static_for<NQ>([&](auto qc) {
                        constexpr int Qi = decltype(qc)::value;
                        constexpr int I = Qi / NBAT4, Bi = Qi % NBAT4;
                        const int ct = wave * CTW + I;
                        if constexpr (Qi + NBUF - 1 < NQ) load_batch(wbuf[(Qi + NBUF - 1) % NBUF], Qi + NBUF - 1);
                        v4i bv[4];
                        if constexpr (Bi == 0) {
#pragma unroll
                            for (int g = 0; g < 4; ++g) bv[g] = *(const v4i*)(B.b4 + ct * 32 + 8 * g + 4 * lh);
                        }
                        if constexpr (NBAT4 > 1) {
                            // NPT <= 2: one pixel pair, its accumulators live across the tile's batches
                            if constexpr (Bi == 0) {
#pragma unroll
                                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                                    for (int r = 0; r < 16; ++r) acc[jj][r] = bv[r >> 2][r & 3];
                            }
                            int kb = Bi * NB + rotk;
                            asm volatile("" : "+s"(kb));
#pragma unroll
                            for (int s = 0; s < NB; ++s) {
                                const int k = (kb + s) & (KK - 1);
#pragma unroll
                                for (int jj = 0; jj < 2; ++jj) {
                                    if (jj >= NPT) continue;
                                    const v4i xf = *(const v4i*)(mid2 + SM::off(jj * 32 + l31, k * 2 + lh));
                                    acc[jj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wbuf[Qi % NBUF][s], xf, acc[jj], 0, 0, 0);
                                }
                            }
                            if constexpr (Bi == NBAT4 - 1) { const v16i none[2] = {}; finish(std::integral_constant<int, I>{}, std::integral_constant<int, 0>{}, none); }
                        } else {
                            // the tile's whole K range is this batch: every pixel pair in turn
                            v4i wsf[DSB ? NB : 1]; v4i bs[4];
                            if constexpr (DSB) {
                                const v4i* const wps = (const v4i*)B.wsc + (size_t)ct * (CIN0 / 32) * 64 + lane;
#pragma unroll
                                for (int s = 0; s < NB; ++s) wsf[s] = wps[(size_t)s * 64];
#pragma unroll
                                for (int g = 0; g < 4; ++g) bs[g] = *(const v4i*)(B.bsc + ct * 32 + 8 * g + 4 * lh);
                            }
                            static_for<NPAIR>([&](auto ppc) {
                                constexpr int pp = decltype(ppc)::value;
                                v16i acs[2];
#pragma unroll
                                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                                    for (int r = 0; r < 16; ++r) { acc[jj][r] = bv[r >> 2][r & 3]; if constexpr (DSB) acs[jj][r] = bs[r >> 2][r & 3]; else acs[jj][r] = 0; }
#pragma unroll
                                for (int s = 0; s < NB; ++s) {
                                    const int k = (s + rotk) & (KK - 1);
#pragma unroll
                                    for (int jj = 0; jj < 2; ++jj) {
                                        const int pt = pp * 2 + jj;
                                        if (pt >= NPT) continue;
                                        const v4i xf = *(const v4i*)(mid2 + SM::off(pt * 32 + l31, k * 2 + lh));
                                        acc[jj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wbuf[Qi % NBUF][s], xf, acc[jj], 0, 0, 0);
                                        if constexpr (DSB) {
                                            const v4i xs = *(const v4i*)(xin + SI::off(pt * 32 + l31, s * 2 + lh));
                                            acs[jj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wsf[s], xs, acs[jj], 0, 0, 0);
                                        }
                                    }
                                }
                                finish(std::integral_constant<int, I>{}, ppc, acs);
                            });
                        }
                    });
                }
                F8_CT(4);
                __syncthreads();                                // x8 is complete (the next block's P1 reads it); mid2 may be rewritten
                return true;
            };
            bool ok;
            if constexpr (DS0) { if (b == 0) ok = block(std::true_type{}); else ok = block(std::false_type{}); }
            else ok = block(std::false_type{});
            if (!ok) return;
        }

        // ---- the int8 copy of the stage output: LDS rows -> whole NHWC rows in HBM
        if (a.q[0].ptr) {
            constexpr int CH = C / 16;
            for (int idx = tid; idx < npx * CH; idx += 512) {
                const int row = idx / CH, c16 = idx % CH;
                const v4i v = *(const v4i*)(x8 + SX::off(row, c16));
                *(v4i*)(a.q[0].ptr + (size_t)(m_tile + row) * C + c16 * 16) = v;
            }
        }
        __syncthreads();                                        // before the next image's tile overwrites x8 / xin
        F8_CT(5);
    }
#ifdef F8_TRACE
    if (a.trace && tid == 0) {
        unsigned long long* tp = (unsigned long long*)a.trace + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 6; ++i) tp[i] = tt[i];
    }
#endif
}

// instances: ResNet-50 stage 0 (opening block + identity blocks, 56x56), stage 1 (28x28), stage 2 (14x14) identity chains
bool chain_supported(int C, int MID, int H, int W, int cin0) {
    if (C == 256 && MID == 64 && H == 56 && W == 56 && (cin0 == 64 || cin0 == 256)) return true;
    if (C == 512 && MID == 128 && H == 28 && W == 28 && cin0 == 512) return true;
    if (C == 1024 && MID == 256 && H == 14 && W == 14 && cin0 == 1024) return true;
    return false;
}
int chain_tiles_per_img(int H, int W) { (void)W; return (H + 3) / 4; }

template <int C, int MID, int W, int H, int R, int CIN0, int NB, int NBUF>
static hipError_t launch_chain_t(const ChainArgs& a, hipStream_t s) {
    using Cfg = ChainCfg<C, MID, W, H, R, CIN0>;
    static unsigned long long attr_done = 0; int attr_dev = -1;
    if (!dyn_lds_opted_in(&attr_done, &attr_dev)) {
        hipError_t e = hipFuncSetAttribute((const void*)chain_kernel<C, MID, W, H, R, CIN0, NB, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        if (attr_dev >= 0) attr_done |= 1ull << attr_dev;
    }
    const int grid = a.NG * Cfg::T;
    if (grid < 1 || grid > 256) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(a.sync, 0, (size_t)kChainSyncWords * 4, s);     // ticket and flags: every launch (also under graph replay)
    if (e != hipSuccess) return e;
#ifdef F8_TRACE
    static unsigned long long* tbuf = nullptr; static int count = 0;
    static const int want = [] { const char* e = getenv("F8_TRACE_CHAIN"); return e ? atoi(e) : -1; }();
    ChainArgs b = a;
    const bool tracing = (count++ == want);
    if (tracing) { if (!tbuf) (void)hipMalloc((void**)&tbuf, (size_t)1 << 16); (void)hipMemsetAsync(tbuf, 0, (size_t)grid * 64, s); b.trace = tbuf; }
    hipLaunchKernelGGL((chain_kernel<C, MID, W, H, R, CIN0, NB, NBUF>), dim3(grid), dim3(512), Cfg::LDS_BYTES, s, b);
    if (tracing) {
        (void)hipStreamSynchronize(s);
        unsigned long long hb[256 * 8];
        (void)hipMemcpy(hb, tbuf, (size_t)grid * 64, hipMemcpyDeviceToHost);
        double ph[6] = {}; int n = 0;
        for (int i = 0; i < grid; ++i) { unsigned long long* p = hb + (size_t)i * 8; ++n; for (int k = 0; k < 6; ++k) ph[k] += (double)p[k]; }
        fprintf(stderr, "[trace chain<%d,%d,%d>] grid %d, %d blocks, N %d: avg cycles per WG (whole launch): load %.0f | P1 %.0f | halo %.0f | P2 %.0f | P3 %.0f | out %.0f\n",
                C, MID, W, grid, a.nblk, a.N, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n, ph[5] / n);
    }
    return hipGetLastError();
#else
    hipLaunchKernelGGL((chain_kernel<C, MID, W, H, R, CIN0, NB, NBUF>), dim3(grid), dim3(512), Cfg::LDS_BYTES, s, a);
    return hipGetLastError();
#endif
}

hipError_t launch_chain(const ChainArgs& a, int C, int MID, int H, int W, int cin0, hipStream_t s) {
    if (a.nblk < 1 || a.nblk > kChainMaxBlocks) return hipErrorInvalidValue;
    if (C == 256 && MID == 64 && H == 56 && W == 56 && cin0 == 64) return launch_chain_t<256, 64, 56, 56, 4, 64, 2, 4>(a, s);
    if (C == 256 && MID == 64 && H == 56 && W == 56 && cin0 == 256) return launch_chain_t<256, 64, 56, 56, 4, 256, 2, 4>(a, s);
    if (C == 512 && MID == 128 && H == 28 && W == 28 && cin0 == 512) return launch_chain_t<512, 128, 28, 28, 4, 512, 4, 3>(a, s);
    if (C == 1024 && MID == 256 && H == 14 && W == 14 && cin0 == 1024) return launch_chain_t<1024, 256, 14, 14, 4, 1024, 4, 3>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace f8
