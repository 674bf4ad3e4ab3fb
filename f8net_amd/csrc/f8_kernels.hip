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
// The XOR is applied in registers between the global load and the LDS write; out-of-image taps are
// fetched through the buffer range check (returns 0 -> XOR -> -128 == real 0), and 128*sum(w) is
// folded into the packed bias on the host (f8_net.cpp: pack_conv_weights).  Everything is mod 2^32,
// so the identity is exact under wrap-around.
#include "f8_internal.h"
#include <cstdlib>

namespace f8 {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

static constexpr unsigned kOOB = 0x80000000u;   // voffset sentinel: beyond any buffer (< 2 GiB each)

// int_op_only_fix_quant on one value; n, lo, hi are wave-uniform.
// n > 0: q = (v + 2^(n-1)) >> n, with the LSB cleared on an exact tie (== ((r >> (n+1)) << 1)).
__device__ __forceinline__ int requant1(int v, int n, int lo, int hi) {
    int q;
    if (n > 0) {
        const unsigned half = 1u << (n - 1);
        const unsigned mask = (half << 1) - 1u;
        const int r = (int)((unsigned)v + half);
        q = r >> n;
        if (((unsigned)v & mask) == half) q &= ~1;
    } else {
        q = (int)((unsigned)v << (-n));
    }
    return min(max(q, lo), hi);
}

__device__ __forceinline__ unsigned pack4(int a, int b, int c, int d) {
    return ((unsigned)a & 0xffu) | (((unsigned)b & 0xffu) << 8) | (((unsigned)c & 0xffu) << 16) |
           ((unsigned)d << 24);
}

// int32 tensors live in an MFMA-fragment-tiled layout ("I32T"), not NHWC: blocks of 32 pixels x 32
// channels (4 KB), inside a block the order is [g = (c%32)/8][lane = ((c/4)&1)*32 + m%32][c%4] — exactly
// the accumulator layout of v_mfma_i32_32x32x32_i8 — so that a wave's residual read / int32 write of
// one accumulator group is ONE contiguous 1 KB transaction (64 lanes x 16 B) instead of 32 scattered
// 32-byte pieces.  m = linear pixel index n*P*Q + p*Q + q; rows are padded to a multiple of 32.
// Returns the int index of channel c (c % 4 == 0 for vector access) of pixel m.
__device__ __forceinline__ size_t i32t_index(int m, int c, int Cs) {
    return ((size_t)(m >> 5) * (size_t)(Cs >> 5) + (size_t)(c >> 5)) * 1024u +
           (size_t)(((((c & 31) >> 3) * 64 + ((c >> 2) & 1) * 32 + (m & 31)) << 2) + (c & 3));
}

__device__ __forceinline__ int clamp_sym31(int v) {   // clamp_(max=2^31-1, min=-(2^31-1))
    return max(v, -2147483647);
}

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM convolution on v_mfma_i32_32x32x32_i8.
//
// Block: 256 threads = 4 waves arranged WPX x WCO; tile BM pixels x BN couts, K step BK bytes.
// MFMA roles: A = weights (rows = cout), B = activations (cols = pixels), so that a lane's 4
// consecutive accumulator registers are 4 consecutive output channels of ONE pixel (NHWC-friendly):
//   D reg r of lane l: cout = (r&3) + 8*(r>>2) + 4*(l>>5), pixel = l&31.
// Both operands are read from LDS as 16-byte K-contiguous chunks (lane l: row l&31, chunk
// 2*kk + (l>>5)); whatever the hardware's internal k order is, it is the same for A and B, and the
// integer sum over k is order-independent.
//
// LDS image: rows of BK bytes; 16-byte chunk c of row r is stored at chunk c ^ f(r),
// f(r) = (r / (256/BK)) % (BK/16): the 16 lanes of a ds_read_b128 service group (distinct rows
// mod 16, same logical chunk) then cover all 64 banks exactly once.
// Pipeline: global -> registers (next K step in flight during the MFMAs) -> XOR -> LDS, two LDS
// buffers, one barrier per K step.
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int BK, int WPX, int WCO, bool HAS_PAD, bool HAS_RES>
__global__ void __launch_bounds__(256) conv_igemm_kernel(const ConvArgs a) {
    static_assert(WPX * WCO == 4, "4 waves");
    constexpr int CPR = BK / 16;                  // chunks per row
    constexpr int RPB = 256 / BK;                 // rows per 256-byte bank row
    constexpr int XCH = BM * CPR, WCH = BN * CPR; // 16-byte chunks per tile
    constexpr int XL = (XCH + 255) / 256, WL = (WCH + 255) / 256;
    constexpr int TPX = BM / WPX / 32, TCO = BN / WCO / 32;
    constexpr int KK = BK / 32;
    constexpr int XBYTES = BM * BK, TILE = (BM + BN) * BK;
    static_assert(TPX >= 1 && TCO >= 1, "wave tile");

    __shared__ __attribute__((aligned(16))) char lds[2 * TILE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

    // ---- per-thread gather descriptors (K-loop invariant)
    unsigned xbase[XL], xlds[XL], wbase[WL], wlds[WL];
    int xh0[XL], xw0[XL];   // top-left input coordinate of each gathered row (HAS_PAD only)
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        const int idx = tid + i * 256;
        const int row = idx / CPR, chunk = idx % CPR;
        xlds[i] = row * BK + ((chunk ^ ((row / RPB) % CPR)) << 4);
        const int m = m0 + row;
        xh0[i] = xw0[i] = -(1 << 24);
        xbase[i] = kOOB;
        if (idx < XCH && m < a.M) {
            const int n = m / a.PQ, rem = m - n * a.PQ;
            const int p = rem / a.Q, q = rem - p * a.Q;
            xbase[i] = (unsigned)(n * a.sN + p * a.sP + q * a.sQ + a.origin + chunk * 16);
            xh0[i] = p * a.stride - a.pad;
            xw0[i] = q * a.stride - a.pad;
        }
    }
#pragma unroll
    for (int j = 0; j < WL; ++j) {
        const int idx = tid + j * 256;
        const int row = idx / CPR, chunk = idx % CPR;
        wlds[j] = XBYTES + row * BK + ((chunk ^ ((row / RPB) % CPR)) << 4);
        // rows past coutP fall outside the buffer and read as 0
        wbase[j] = (idx < WCH) ? (unsigned)((co0 + row) * a.ktot + chunk * 16) : kOOB;
    }

    // ---- per-lane fragment addresses
    const int l31 = lane & 31, lh = lane >> 5;
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

    // Residual operand: ALL of the wave tile's int32 rows are requested before the K loop, so their
    // HBM latency overlaps the operand staging and the MFMAs (the in-place out32 store of the same
    // addresses comes later from the same lane).  16 B per lane per (cout tile, pixel tile, group).
    v4i rv[HAS_RES ? TCO : 1][HAS_RES ? TPX : 1][4];
    if (HAS_RES) {
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

    v4i xr[XL], wr[WL];
    const int nk = a.ktot / BK;
    // K-step state (wave-uniform): tap row/col, channel offset inside the tap
    int tr = 0, ts = 0, c0 = 0;

    auto issue_loads = [&](int ks) {
        const unsigned koffx = (unsigned)(tr * a.tapH + ts * a.tapW + c0);
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            unsigned off = xbase[i] + koffx;
            if (HAS_PAD)   // out-of-image tap: fetch through the range check (reads 0)
                off = ((unsigned)(xh0[i] + tr) < (unsigned)a.H && (unsigned)(xw0[i] + ts) < (unsigned)a.W) ? off : kOOB;
            if (i * 256 + 255 < XCH || tid + i * 256 < XCH)
                xr[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
        }
        const unsigned koffw = (unsigned)(ks * BK);
#pragma unroll
        for (int j = 0; j < WL; ++j)
            if (j * 256 + 255 < WCH || tid + j * 256 < WCH)
                wr[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, wbase[j] + koffw, 0, 0);
        // advance to the next K step
        c0 += BK;
        if (c0 == a.CK) {
            c0 = 0; ++ts;
            if (ts == a.kw) { ts = 0; ++tr; }
        }
    };
    auto stage_to_lds = [&](int buf) {
        char* base = lds + buf * TILE;
#pragma unroll
        for (int i = 0; i < XL; ++i)
            if (i * 256 + 255 < XCH || tid + i * 256 < XCH) {
                v4i v = xr[i];
                v.x ^= (int)a.xor_mask; v.y ^= (int)a.xor_mask; v.z ^= (int)a.xor_mask; v.w ^= (int)a.xor_mask;
                *(v4i*)(base + xlds[i]) = v;
            }
#pragma unroll
        for (int j = 0; j < WL; ++j)
            if (j * 256 + 255 < WCH || tid + j * 256 < WCH) *(v4i*)(base + wlds[j]) = wr[j];
    };

    issue_loads(0);
    stage_to_lds(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) issue_loads(ks + 1);
        const char* base = lds + buf * TILE;
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
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
        if (ks + 1 < nk) stage_to_lds(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias -> ReLU -> [align + residual + clamp -> ReLU] -> int32 / requantised int8
    // ReLUs are branch-free floors (INT32_MIN = no ReLU).
    const int floor0 = a.relu0 ? 0 : INT32_MIN, floor1 = a.relu1 ? 0 : INT32_MIN;
#pragma unroll
    for (int i = 0; i < TCO; ++i) {
        const int cot = co0 + wco * (BN / WCO) + i * 32;   // first cout of this 32-wide MFMA tile
        if (cot >= a.coutP) continue;                      // wave-uniform
        v4i bv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[g] = *(const v4i*)(a.bias + cot + 8 * g + 4 * lh);
#pragma unroll
        for (int j = 0; j < TPX; ++j) {
            const int m = m0 + wpx * (BM / WPX) + j * 32 + l31;
            const bool ok = m < a.M;
            const size_t rowo = (size_t)m * (size_t)a.coutP;
            int y[4][4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int v = max((int)((unsigned)acc[i][j][4 * g + e] + (unsigned)bv[g][e]), floor0);
                    if (HAS_RES) {
                        const unsigned s = ((unsigned)v << a.acc_shl) + ((unsigned)rv[i][j][g][e] << a.res_shl);
                        v = max(clamp_sym31((int)s), floor1);
                    }
                    y[g][e] = v;
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
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    d[g] = pack4(requant1(y[g][0], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(y[g][1], a.q[k].n, a.q[k].lo, a.q[k].hi),
                                 requant1(y[g][2], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(y[g][3], a.q[k].n, a.q[k].lo, a.q[k].hi));
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
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int h = h0 + r;
            if ((unsigned)h >= (unsigned)a.H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int w = w0 + s;
                if ((unsigned)w >= (unsigned)a.W) continue;
                const unsigned xv = *(const unsigned*)(a.x + (((size_t)n * a.H + h) * a.W + w) * a.Cs + c);
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
                          requant1(acc[2], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(acc[3], a.q[k].n, a.q[k].lo, a.q[k].hi));
    }
}

// ---------------------------------------------------------------------------------------------
// Max-pool (NHWC).  int32 input: exact max, then any of {int32, two requantised int8} outputs.
// int8 input (already in the single consumer format; requant is monotone so pooling commutes with
// it exactly): per-byte max, signed or unsigned.
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
                        const int xe = a.in_signed ? (int)(signed char)(xv >> (8 * e)) : (int)((xv >> (8 * e)) & 0xffu);
                        mx[e] = max(mx[e], xe);
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
                              requant1(mx[2], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(mx[3], a.q[k].n, a.q[k].lo, a.q[k].hi));
        }
    }
}

// FXQAvgPool2d int branch: int64 sum over H*W, truncate to int32 (fix_quant_ops.py:130-133).
__global__ void __launch_bounds__(256) avgpool_kernel(const AvgArgs a) {
    const int cgs = a.Cs >> 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.N * cgs) return;
    const int cg = idx % cgs, n = idx / cgs, c = cg << 2;
    long long s[4] = {0, 0, 0, 0};
    for (int i = 0; i < a.HW; ++i) {
        const v4i v = *(const v4i*)(a.x + i32t_index(n * a.HW + i, c, a.Cs));
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
    }
    int t[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = (int)(unsigned)(unsigned long long)s[e];
    const size_t o = (size_t)n * a.Cs + c;
    if (a.out32) { v4i v = {t[0], t[1], t[2], t[3]}; *(v4i*)(a.out32 + i32t_index(n, c, a.Cs)) = v; }
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (a.q[k].ptr)
            *(unsigned*)(a.q[k].ptr + o) =
                pack4(requant1(t[0], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(t[1], a.q[k].n, a.q[k].lo, a.q[k].hi),
                      requant1(t[2], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(t[3], a.q[k].n, a.q[k].lo, a.q[k].hi));
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
                          requant1(y[2], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(y[3], a.q[k].n, a.q[k].lo, a.q[k].hi));
    }
}

// Network input: int32 NCHW -> zero-haloed NHWC4 int8 (stem) / NHWC int8 / NHWC int32.
// One thread per (n, h, w); C planes are read coalesced along w.
__global__ void __launch_bounds__(256) input_kernel(const InArgs a) {
    const size_t total = (size_t)a.N * a.H * a.W;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(idx % a.W);
        size_t t = idx / a.W;
        const int h = (int)(t % a.H);
        const int n = (int)(t / a.H);
        const int32_t* xp = a.x + ((size_t)n * a.C * a.H + h) * a.W + w;
        const size_t plane = (size_t)a.H * a.W;
        if (a.stem) {
            int v[4] = {0, 0, 0, 0};
            for (int c = 0; c < a.C; ++c) v[c] = xp[c * plane];
            *(unsigned*)(a.stem + ((((size_t)n * a.Hp + h + a.pad) * a.Wp) + w + a.pad) * 4) = pack4(v[0], v[1], v[2], v[3]);
        }
        if (a.out8) {
            int8_t* o = a.out8 + (((size_t)n * a.H + h) * a.W + w) * a.Cs8;
            for (int c = 0; c < a.C; ++c) o[c] = (int8_t)xp[c * plane];
            for (int c = a.C; c < a.Cs8; ++c) o[c] = 0;
        }
        if (a.out32) {
            const int m = (n * a.H + h) * a.W + w;
            for (int c = 0; c < a.Cs32; ++c) a.out32[i32t_index(m, c, a.Cs32)] = c < a.C ? xp[c * plane] : 0;
        }
    }
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
        if (a.as_float) ((float*)a.out)[idx] = (float)v;
        else ((int*)a.out)[idx] = v;
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

bool pick_conv_tile(int M, int coutP, int ck, bool has_res, ConvTile* t) {
    if (ck % 32 != 0 || coutP % 32 != 0) return false;
    t->bk = (ck % 64 == 0) ? 64 : 32;
    t->bn = coutP >= 96 ? 128 : (coutP > 32 ? 64 : 32);
    t->bm = 128;
    // residual-carrying epilogues hold the int32 operand in registers: the 64-wide tile keeps
    // 4 waves/SIMD resident instead of 2, which is what a streaming epilogue needs
    if (has_res && t->bn == 128) t->bn = 64;
    // not enough workgroups for 256 CUs: shrink the tile (bm first: keeps cout reuse of X rows)
    auto tiles = [&](int bm, int bn) { return (long)((M + bm - 1) / bm) * ((coutP + bn - 1) / bn); };
    if (tiles(t->bm, t->bn) < 512 && t->bn >= 64) t->bm = 64;
    if (tiles(t->bm, t->bn) < 512 && t->bn == 128) t->bn = 64;
    // experiment hooks (tuning only)
    if (const char* e = getenv(has_res ? "F8_RES_BN" : "F8_BN")) { const int v = atoi(e); if (v == 32 || v == 64 || v == 128) t->bn = v; }
    if (const char* e = getenv(has_res ? "F8_RES_BM" : "F8_BM")) { const int v = atoi(e); if (v == 64 || v == 128) t->bm = v; }
    if (t->bm == 64 && t->bn == 32) t->bn = 64;
    return true;
}

int conv_grid(const ConvTile& t, int M, int coutP) {
    return ((M + t.bm - 1) / t.bm) * ((coutP + t.bn - 1) / t.bn);
}

template <int BM, int BN, int BK, int WPX, int WCO>
static hipError_t launch_conv_t(const ConvArgs& a, int grid, hipStream_t s) {
    const bool pad = a.pad > 0, res = a.res != nullptr;
    if (pad && res) hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WPX, WCO, true, true>), dim3(grid), dim3(256), 0, s, a);
    else if (pad) hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WPX, WCO, true, false>), dim3(grid), dim3(256), 0, s, a);
    else if (res) hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WPX, WCO, false, true>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WPX, WCO, false, false>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_conv(const ConvArgs& a, const ConvTile& t, hipStream_t s) {
    const int grid = conv_grid(t, a.M, a.coutP);
#define F8_CASE(BM_, BN_, BK_, WPX_, WCO_) \
    if (t.bm == BM_ && t.bn == BN_ && t.bk == BK_) return launch_conv_t<BM_, BN_, BK_, WPX_, WCO_>(a, grid, s);
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
    const size_t work = (size_t)a.N * a.P * a.Q * (a.Cs >> 2);
    if (a.in_signed) hipLaunchKernelGGL(dwconv3x3_kernel<true>, dim3(grid_for(work)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(dwconv3x3_kernel<false>, dim3(grid_for(work)), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_maxpool(const PoolArgs& a, hipStream_t s) {
    const size_t work = (size_t)a.N * a.P * a.Q * (a.Cs >> 2);
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(work)), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_avgpool(const AvgArgs& a, hipStream_t s) {
    const size_t work = (size_t)a.N * (a.Cs >> 2);
    hipLaunchKernelGGL(avgpool_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_add(const AddArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(add_kernel, dim3(grid_for((size_t)a.M * (a.Cs >> 2))), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_input(const InArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(input_kernel, dim3(grid_for((size_t)a.N * a.H * a.W)), dim3(256), 0, s, a);
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
