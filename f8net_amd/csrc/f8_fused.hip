// f8_fused.hip — one launch for a whole ResNet bottleneck identity block (gfx950).
//
//   x8 (int8 NHWC, C ch) --1x1 C->MID, ReLU--> mid1 --3x3 MID->MID pad 1, ReLU--> mid2 --1x1 MID->C-->
//   + residual (int32, I32T) -> clamp -> ReLU -> y32 (I32T) and/or requantised int8 copies
//
// i.e. IntBlock.forward of /root/reference/models/fix_resnet.py:26-54,77 for a Bottleneck identity block
// (:233-319), with every int_op_only_fix_quant (fix_quant_ops.py:90-114) in place: after the first
// and second conv the values are requantised to the next layer's 8-bit format and stay in LDS.
//
// Why fuse: unfused, the three convs are three launches whose small ones are latency / issue bound
// while the residual-carrying one is HBM bound; fused, the MFMA work of all three hides under the
// residual stream of the block, the two int8 intermediates never touch HBM and two launches per block
// disappear.
//
// Work unit: R output rows x full width W of ONE image (so only a vertical halo exists).
//   P1  mid1 for rows p0-1 .. p0+R (halo rows recomputed), GEMM [(R+2)*W px] x [MID] x K=C,
//       operands streamed through a 2-stage LDS ring by LDS-direct DMA; result requantised and written
//       into an LDS "patch" [(R+2)][W+2][MID] whose border (image edge / halo columns) holds the
//       biased zero, so the 3x3 needs no border classes.
//   P2  3x3 from the patch (tap = constant LDS offset), weights streamed; result -> LDS mid2.
//   P3  1x1 MID->C in chunks of 64 output channels: weights streamed, residual chunk prefetched one
//       chunk ahead, fused epilogue (align, add, clamp, ReLU, int32 + int8 stores).
// 256 threads = 4 waves; MFMA v_mfma_i32_32x32x32_i8 with A = weights, B = activations as in
// conv_igemm_kernel; all LDS rows are XOR-swizzled per 16-byte chunk (see f8_kernels.hip).
#include "f8_device.h"

namespace f8 {

template <int ROWB> struct Swz {                       // rows of ROWB bytes (64 or 128)
    static constexpr int CPR = ROWB / 16, RPB = 256 / ROWB;
    static __device__ __forceinline__ int f(int row) { return (row / RPB) % CPR; }
    static __device__ __forceinline__ unsigned off(int row, int chunk) { return (unsigned)(row * ROWB + ((chunk ^ f(row)) << 4)); }
};

template <int C, int MID, int W, int R>
__global__ void __launch_bounds__(256) fused_bottleneck_kernel(const FusedArgs a) {
    static_assert(MID == 64, "instantiated for MID = 64 (stage 0); MID = 128 needs dynamic LDS");
    constexpr int PW = W + 2, PR = R + 2;
    constexpr int PATCH_PX = PR * PW;
    constexpr int P1_PX = PR * W;                        // mid1 pixels computed (no halo columns)
    constexpr int NP1 = (P1_PX + 31) / 32;               // px tiles of P1
    constexpr int NP1W = (NP1 + 3) / 4;                  // px tiles per wave in P1
    constexpr int OUT_PX = R * W;
    constexpr int NPO = (OUT_PX + 31) / 32;
    static_assert(NPO == 4, "one output px tile per wave");
    constexpr int CM = MID / 32;                         // co tiles of mid
    constexpr int NK1 = C / 64;                          // P1 K steps
    constexpr int NK2 = 9 * (MID / 64);                  // P2 K steps
    constexpr int NC3 = C / 64;                          // P3 chunks of 64 output channels
    constexpr int KK3 = MID / 32;
    constexpr int X1_ROWS = NP1 * 32;
    constexpr int X1_BYTES = X1_ROWS * 64, W_BYTES = MID * 64;
    constexpr int RING = X1_BYTES + W_BYTES;             // largest stage (P1); P2/P3 stages use its first W_BYTES.. bytes
    constexpr int PATCH_BYTES = PATCH_PX * MID, MID2_BYTES = NPO * 32 * MID;
    static_assert(PATCH_BYTES % 16 == 0, "alignment");
    static_assert(PATCH_BYTES + MID2_BYTES + 2 * RING <= 65536, "static LDS");
    constexpr int XS1 = X1_ROWS * 4;                     // 16-byte slots of the P1 X tile
    constexpr int XL1 = (XS1 + 255) / 256;
    constexpr int WS = MID * 4, WL = (WS + 255) / 256;   // W0 / W2 tile slots (rows of 64 B)
    constexpr int W4S = 64 * (MID / 16), W4L = (W4S + 255) / 256;
    static_assert(WS == 256 && W4S == 256, "one weight slot per thread");

    __shared__ __attribute__((aligned(16))) char lds[PATCH_BYTES + MID2_BYTES + 2 * RING];
    char* const patch = lds;
    char* const mid2 = lds + PATCH_BYTES;
    char* const ring = lds + PATCH_BYTES + MID2_BYTES;

    using S64 = Swz<64>;
    using SM = Swz<MID>;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 3;
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- tile: XCD-aware order (consecutive tiles = vertically adjacent row groups share halo rows in L2)
    int t;
    {
        const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
        t = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    }
    const int n = t / a.tiles_per_img, ti = t - n * a.tiles_per_img;
    const int p0 = ti * R;
    const int rows_out = (a.H - p0) < R ? (a.H - p0) : R;
    const int gp1 = (n * a.H + p0 - 1) * W;              // global pixel index of P1 pixel 0 (may be < 0)
    const int m_tile = (n * a.H + p0) * W;               // global pixel index of output pixel 0

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x8, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w0, 0, a.w0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2, 0, a.w2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw4 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w4, 0, a.w4_bytes, 0x00020000);

    // ---- patch <- biased zero everywhere (border and out-of-image rows keep it)
    {
        const unsigned z = a.xor1;
        const v4i zv = {(int)z, (int)z, (int)z, (int)z};
        for (int o = tid * 16; o < PATCH_BYTES; o += 256 * 16) *(v4i*)(patch + o) = zv;
    }

    // ---- P1 gather descriptors
    unsigned xb1[XL1];
#pragma unroll
    for (int i = 0; i < XL1; ++i) {
        const int idx = tid + i * 256;
        const int row = idx >> 2, chunk = (idx & 3) ^ S64::f(row);
        const int pr = row / W;                          // patch row of this P1 pixel
        const int hrow = p0 - 1 + pr;
        const bool ok = idx < XS1 && row < P1_PX && hrow >= 0 && hrow < a.H;
        xb1[i] = ok ? (unsigned)((gp1 + row) * C + chunk * 16) : kOOB;
    }
    const int wrow = tid >> 2, wchunk = (tid & 3) ^ S64::f(wrow);     // weight tile slot (rows of 64 B)
    const unsigned w0b = (unsigned)(wrow * C + wchunk * 16);
    const unsigned w2b = (unsigned)(wrow * (9 * MID) + wchunk * 16);
    const int w4row = tid / (MID / 16), w4chunk = (tid % (MID / 16)) ^ SM::f(w4row);
    const unsigned w4b = (unsigned)(w4row * MID + w4chunk * 16);

    auto issue_p1 = [&](int ks, int slot) {
        char* base = ring + slot * RING;
#pragma unroll
        for (int i = 0; i < XL1; ++i) {
            const unsigned off = xb1[i] == kOOB ? kOOB : xb1[i] + (unsigned)(ks * 64);
            if ((i * 256 + wave * 64) < XS1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(base + i * 4096 + wave * 1024), 16, off, 0, 0, 0);
        }
        const unsigned woff = w0b + (unsigned)(ks * 64);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw0, (__attribute__((address_space(3))) void*)(base + X1_BYTES + wave * 1024), 16, woff, 0, 0, 0);
    };
    auto issue_w2 = [&](int j, int slot) {               // step j: tap j / (MID/64), 64-byte chunk j % (MID/64)
        char* base = ring + slot * RING;
        const unsigned woff = w2b + (unsigned)(j * 64);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw2, (__attribute__((address_space(3))) void*)(base + wave * 1024), 16, woff, 0, 0, 0);
    };
    auto issue_w4 = [&](int c, int slot) {               // 64 output channels x MID bytes
        char* base = ring + slot * RING;
        const unsigned woff = w4b + (unsigned)(c * 64 * MID);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw4, (__attribute__((address_space(3))) void*)(base + wave * 1024), 16, woff, 0, 0, 0);
    };

    // =========================================================================================
    // P1: mid1 = requant(relu(W0 . x8 + b0)) on (R+2) x W pixels  ->  patch
    // =========================================================================================
    {
        v16i acc[NP1W][CM];
#pragma unroll
        for (int j = 0; j < NP1W; ++j)
#pragma unroll
            for (int i = 0; i < CM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0;
        unsigned cof[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) cof[kk] = (unsigned)(((kk * 2 + lh) ^ S64::f(l31)) << 4);

        issue_p1(0, 0);
        for (int ks = 0; ks < NK1; ++ks) {
            wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (ks + 1 < NK1) issue_p1(ks + 1, (ks + 1) & 1);
            const char* base = ring + (ks & 1) * RING;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                v4i wf[CM], xf[NP1W];
#pragma unroll
                for (int i = 0; i < CM; ++i) wf[i] = *(const v4i*)(base + X1_BYTES + (i * 32 + l31) * 64 + cof[kk]);
#pragma unroll
                for (int j = 0; j < NP1W; ++j) {
                    const int pt = wave + 4 * j;
                    if (pt < NP1) xf[j] = *(const v4i*)(base + (pt * 32 + l31) * 64 + cof[kk]);
                }
#pragma unroll
                for (int j = 0; j < NP1W; ++j) {
                    const int pt = wave + 4 * j;
                    if (pt < NP1)
#pragma unroll
                        for (int i = 0; i < CM; ++i) acc[j][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[i], xf[j], acc[j][i], 0, 0, 0);
                }
            }
        }
        // first W2 stage can already travel: its slot was last read two steps ago
        issue_w2(0, NK1 & 1);

        // epilogue: bias, ReLU, requant to body.2's input format, into the patch
        const int floor0 = a.relu_a ? 0 : INT32_MIN;
#pragma unroll
        for (int j = 0; j < NP1W; ++j) {
            const int pt = wave + 4 * j;
            if (pt >= NP1) continue;                     // wave-uniform
            const int pix = pt * 32 + l31;
            const int pr = pix / W, pc = pix - pr * W;
            const int hrow = p0 - 1 + pr;
            const bool ok = pix < P1_PX && hrow >= 0 && hrow < a.H;
            const int ppx = pr * PW + pc + 1;
#pragma unroll
            for (int i = 0; i < CM; ++i) {
                unsigned d[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const v4i bv = *(const v4i*)(a.b0 + i * 32 + 8 * g + 4 * lh);
                    int y[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = requant1(max((int)((unsigned)acc[j][i][4 * g + e] + (unsigned)bv[e]), floor0), a.n1, a.lo1, a.hi1);
                    d[g] = pack4(y[0], y[1], y[2], y[3]) ^ a.xor1;
                }
                auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
                if (ok) {
                    v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
                    *(v4i*)(patch + SM::off(ppx, i * 2 + lh)) = o;
                }
            }
        }
    }

    // =========================================================================================
    // P2: mid2 = requant(relu(conv3x3(mid1) + b2)) on R x W pixels  ->  mid2
    // =========================================================================================
    const int opix = wave * 32 + l31;                    // this lane's output pixel in the tile
    const bool opix_ok = opix < rows_out * W;
    {
        v16i acc[CM];
#pragma unroll
        for (int i = 0; i < CM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0;
        const int oc = opix < OUT_PX ? opix : OUT_PX - 1;   // padding lanes read a valid pixel, result unused
        const int orow = oc / W, ocol = oc - orow * W;
        const int bpx = orow * PW + ocol;                // patch pixel of tap (0,0)
        unsigned cof[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) cof[kk] = (unsigned)(((kk * 2 + lh) ^ S64::f(l31)) << 4);

        constexpr int S0 = NK1 & 1;                      // ring slot of W2 step 0
        int tr = 0, ts = 0;
        for (int j = 0; j < NK2; ++j) {
            wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                // step 0: also "patch complete"
            if (j + 1 < NK2) issue_w2(j + 1, (S0 + j + 1) & 1);
            const char* base = ring + ((S0 + j) & 1) * RING;
            const int ppx = bpx + tr * PW + ts;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const v4i xf = *(const v4i*)(patch + SM::off(ppx, kk * 2 + lh));
#pragma unroll
                for (int i = 0; i < CM; ++i) {
                    const v4i wf = *(const v4i*)(base + (i * 32 + l31) * 64 + cof[kk]);
                    acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, xf, acc[i], 0, 0, 0);
                }
            }
            if (++ts == 3) { ts = 0; ++tr; }
        }
        asm volatile("" ::: "memory");
        issue_w4(0, (S0 + NK2) & 1);
        asm volatile("" ::: "memory");   // the bias / residual loads below must stay behind this DMA (counted wait in P3)

        const int floor0 = a.relu_b ? 0 : INT32_MIN;
#pragma unroll
        for (int i = 0; i < CM; ++i) {
            unsigned d[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4i bv = *(const v4i*)(a.b2 + i * 32 + 8 * g + 4 * lh);
                int y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = requant1(max((int)((unsigned)acc[i][4 * g + e] + (unsigned)bv[e]), floor0), a.n2, a.lo2, a.hi2);
                d[g] = pack4(y[0], y[1], y[2], y[3]) ^ a.xor2;
            }
            auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
            v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
            *(v4i*)(mid2 + SM::off(opix, i * 2 + lh)) = o;
        }
    }

    // =========================================================================================
    // P3: y = clamp((W4 . mid2 + b4) << sa + (x << sr)) [ReLU]  ->  y32 (I32T) / int8 copies
    // =========================================================================================
    {
        constexpr int S0 = (NK1 + NK2) & 1;
        const int m = m_tile + opix;                     // global output pixel of this lane
        const int mc = opix_ok ? m : m_tile;             // padding lanes: any valid pixel (loads only)
        const int floor1 = a.relu1 ? 0 : INT32_MIN;
        unsigned cofm[KK3];
#pragma unroll
        for (int kk = 0; kk < KK3; ++kk) cofm[kk] = (unsigned)(((kk * 2 + lh) ^ SM::f(l31)) << 4);
        v4i xf[KK3];                                     // this wave's mid2 fragments are chunk-invariant: read once
        // (read after the first barrier below)
        v4i rv[2][4], rn[2][4];
        auto load_res = [&](v4i (&dst)[2][4], int c) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) dst[i][g] = *(const v4i*)(a.xr + i32t_index(mc, c * 64 + i * 32 + 8 * g + 4 * lh, C));
        };
        load_res(rv, 0);
        // one chunk of 64 output channels; `cur` holds this chunk's residual, `nxt` receives the next one's
        auto chunk = [&](int c, v4i (&cur)[2][4], v4i (&nxt)[2][4]) {
            // W4 chunk c landed?  Everything issued after it (>= 8 loads: residual prefetch, bias, plus the
            // previous chunk's stores) may stay in flight; all of it is newer than the DMA, so the count is safe.
            wait_vmcnt<8>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                // chunk 0: also "mid2 complete"
            // compiler fences: the counted wait of the NEXT chunk assumes that at least the 8 residual loads below are issued
            // AFTER this DMA in program order; without the fences hipcc is free to hoist those loads above it
            asm volatile("" ::: "memory");
            if (c + 1 < NC3) issue_w4(c + 1, (S0 + c + 1) & 1);
            asm volatile("" ::: "memory");
            const char* base = ring + ((S0 + c) & 1) * RING;
            if (c == 0) {
#pragma unroll
                for (int kk = 0; kk < KK3; ++kk) xf[kk] = *(const v4i*)(mid2 + (wave * 32 + l31) * MID + cofm[kk]);
            }
            v16i acc[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0;
#pragma unroll
            for (int kk = 0; kk < KK3; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const v4i wf = *(const v4i*)(base + (i * 32 + l31) * MID + cofm[kk]);
                    acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, xf[kk], acc[i], 0, 0, 0);
                }
            // prefetch the next chunk's residual (always 8 loads per wave: the counted wait relies on it)
            load_res(nxt, c + 1 < NC3 ? c + 1 : c);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int cot = c * 64 + i * 32;
                int y[4][4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const v4i bv = *(const v4i*)(a.b4 + cot + 8 * g + 4 * lh);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned v = (unsigned)acc[i][4 * g + e] + (unsigned)bv[e];
                        const unsigned s = (v << a.acc_shl) + ((unsigned)cur[i][g][e] << a.res_shl);
                        y[g][e] = max(clamp_sym31((int)s), floor1);
                    }
                }
                if (a.out32 && opix_ok) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        v4i o = {y[g][0], y[g][1], y[g][2], y[g][3]};
                        *(v4i*)(a.out32 + i32t_index(m, cot + 8 * g + 4 * lh, C)) = o;
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
                    if (opix_ok) {
                        v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
                        *(v4i*)(a.q[k].ptr + (size_t)m * C + cot + 16 * lh) = o;
                    }
                }
            }
        };
        static_assert(NC3 % 2 == 0, "chunk loop is unrolled by two (residual ping-pong)");
        for (int c = 0; c < NC3; c += 2) {
            chunk(c, rv, rn);
            chunk(c + 1, rn, rv);
        }
    }
}

hipError_t launch_fused_bottleneck(const FusedArgs& a, hipStream_t s) {
    const int grid = a.N * a.tiles_per_img;
    if (a.C == 256 && a.MID == 64 && a.W == 56 && a.R == 2) {
        hipLaunchKernelGGL((fused_bottleneck_kernel<256, 64, 56, 2>), dim3(grid), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    return hipErrorInvalidValue;
}

bool fused_bottleneck_supported(int C, int MID, int H, int W, int* R) {
    if (C == 256 && MID == 64 && W == 56 && H % 2 == 0) { *R = 2; return true; }
    return false;
}

}  // namespace f8
