// f8_block.h — device helpers of the 8-wave "2x2 block + K split" conv kernel (f8_conv3x3.hip).
//
// Workgroup tile = 4 pixel tiles x NCO cout tiles of 32x32 (NCO = 2 or 4).  NB = 2 * (NCO / 2) wave BLOCKS of 2x2 tiles;
// the KS = 8 / NB waves of a block split the 32-byte K slices of every stage between them.  acc[i][j]: i = cout tile of
// the pair, j = pixel tile of the pair.  After the K loop each wave FINISHES NF = 4 / KS tiles of its block:
//   KS == 2: pixel tile ks, cout tiles 0 and 1        KS == 4: pixel tile ks & 1, cout tile ks >> 1
#pragma once
#include "f8_device.h"

namespace f8 {

// Partial sums meet in LDS: a wave keeps its partials of the tiles it finishes in registers, parks the other
// 4 - NF tiles at park[wave][slot], and adds its partners' parked partials of its own tiles (wrapping int32 adds:
// order-free, exact).  The caller guarantees that nobody reads the LDS bytes at `park` any more (the helper
// starts with a barrier) and that 8 * (4 - NF) * 4 KB are available there.
template <int KS, int NB, int NF>
__device__ __forceinline__ void block_exchange(v16i (&acc)[2][2], v4i* park, int wave, int blk, int ks, int lane, v4i (&fin)[NF][4]) {
    constexpr int NPARK = 4 - NF;
    auto park_at = [&](int w, int slot, int g) { return park + ((w * NPARK + slot) * 4 + g) * 64 + lane; };
    auto owner = [](int i, int j) { return KS == 2 ? j : (2 * i + j); };
    // parking slot of tile (i, j) in non-owner k's area: rank of the tile among those wave k does not own
    auto slot_of = [&](int k, int i, int j) {
        int sl = 0;
        for (int ii = 0; ii < 2; ++ii)
            for (int jj = 0; jj < 2; ++jj) {
                if (ii == i && jj == j) return sl;
                if (owner(ii, jj) != k) ++sl;
            }
        return sl;
    };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's earlier LDS reads (fragments, a previous exchange) are done
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                if (k == ks && owner(i, j) != k) {              // wave-uniform
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        v4i v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        *park_at(wave, slot_of(k, i, j), g) = v;
                    }
                }
            }
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // Own tiles: explicit wave-uniform switch with CONSTANT accumulator indices in every arm (a loop with a run-time
    // "is this tile mine" test gets folded by the compiler into acc[ks >> 1][ks & 1], i.e. a dynamically indexed private
    // array = the accumulators in scratch memory)
    auto finish_tile = [&](v16i& t, int i, int j, int f) {      // i, j, f are literals at every call site
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            v4i v = {t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]};
#pragma unroll
            for (int o = 0; o < KS; ++o)
                if (o != ks) v += *park_at(blk + o * NB, slot_of(o, i, j), g);
            fin[f][g] = v;
        }
    };
    if constexpr (KS == 2) {
        if (ks == 0) { finish_tile(acc[0][0], 0, 0, 0); finish_tile(acc[1][0], 1, 0, 1); }
        else { finish_tile(acc[0][1], 0, 1, 0); finish_tile(acc[1][1], 1, 1, 1); }
    } else {
        switch (ks) {
            case 0: finish_tile(acc[0][0], 0, 0, 0); break;
            case 1: finish_tile(acc[0][1], 0, 1, 0); break;
            case 2: finish_tile(acc[1][0], 1, 0, 0); break;
            default: finish_tile(acc[1][1], 1, 1, 0); break;
        }
    }
}

// Fused epilogue of NF finished 32x32 tiles (cout tiles cot0, cot0 + 32, ...; this lane's output pixel m):
// bias -> ReLU floor -> [align + residual + clamp -> ReLU floor] -> int32 (I32T) and / or up to two requantised int8
// (NHWC) outputs.  rv[i][g] = residual operand (or the second GEMM's result), bq[i][g] = bias, fragment order.
template <int NF, bool HAS_RES>
__device__ __forceinline__ void block_finish(const ConvArgs& a, v4i (&fin)[NF][4], v4i (&bq)[NF][4], v4i (&rv)[HAS_RES ? NF : 1][4],
                                             int cot0, int m, bool pix_ok, int lh) {
    const int floor0 = a.relu0 ? 0 : INT32_MIN, floor1 = a.relu1 ? 0 : -2147483647 /* the join's clamp_(min=-(2^31-1)) and the ReLU floor are one max */;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int cot = cot0 + i * 32;
        if (cot >= a.coutP) continue;                           // wave-uniform
        int y[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int v = max((int)((unsigned)fin[i][g][e] + (unsigned)bq[i][g][e]), floor0);
                if (HAS_RES) {
                    const unsigned s = ((unsigned)v << a.acc_shl) + ((unsigned)rv[i][g][e] << a.res_shl);
                    v = max((int)s, floor1);
                }
                y[g][e] = v;
            }
        if (a.out32 && pix_ok) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v4i o = {y[g][0], y[g][1], y[g][2], y[g][3]};
                *(v4i*)(a.out32 + i32t_index(m, cot + 8 * g + 4 * lh, a.coutP)) = o;
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!a.q[k].ptr) continue;                          // wave-uniform
            unsigned d[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                d[g] = pack4(requant1(y[g][0], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(y[g][1], a.q[k].n, a.q[k].lo, a.q[k].hi),
                             requant1(y[g][2], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(y[g][3], a.q[k].n, a.q[k].lo, a.q[k].hi)) ^ a.q[k].bias_xor;
            auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
            if (pix_ok) {
                v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
                *(v4i*)(a.q[k].ptr + (size_t)m * a.coutP + cot + 16 * lh) = o;
            }
        }
    }
}

}  // namespace f8
