// f8_stem.hip — ResNet head in one launch: 7x7 / stride 2 / pad 3 convolution (Cin <= 4, 64 couts) + ReLU + requant to the
// next layer's unsigned 8-bit format + 3x3 / stride 2 / pad 1 max-pool (gfx950 only).
//
// Unfused, the 112x112x64 conv output is written (103 MB per 128 images) and read back by the pool; here it only ever
// exists as a 15x17-pixel int8 tile in LDS.  A workgroup owns 7x8 POOLED pixels = a 15x17 region of conv pixels
// (255 = 8 MFMA pixel tiles, one per wave; 14 % of the conv is recomputed at tile seams):
//   * the 35x40-pixel NHWC4 input patch (5.6 KB) and ALL weights (64 couts x 7 kernel rows x 32 B, rows padded to
//     240 B for conflict-free fragment reads) arrive by LDS-direct DMA once; one kernel row of 8 pixels x 4 channels is
//     one 32-byte K step, so the conv is 7 MFMA steps per tile with no barrier in between;
//   * epilogue -> int8 tile in LDS (conv pixels outside the image hold 0: post-ReLU unsigned values make 0 the identity
//     of max, and every pool window has an in-image tap); pool = packed 16-bit max over 9 LDS reads per 16 channels.
// Arithmetic: the stem of conv_igemm_kernel + maxpool_kernel, bit for bit (reference: fix_resnet.py:354-359; the float
// MaxPool detour there is exact; requant commutes with max because it is monotone).
#include "f8_device.h"
#include <cstdlib>

namespace f8 {

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));

namespace {
constexpr int TP = 7, TQ = 8;                       // pooled pixels per tile
constexpr int RH = 2 * TP + 1, RW = 2 * TQ + 1;     // conv region 15 x 17
constexpr int RPX = RH * RW;                        // 255
constexpr int PH = 2 * (RH - 1) + 7, PWD = 2 * (RW - 1) + 8;   // input patch 35 x 40 pixels (4 B each)
constexpr int PCH = PWD * 4 / 16;                   // 16-byte chunks per patch row (10)
constexpr int PSLOTS = PH * PCH;                    // 350
constexpr int PATCH_BYTES = (PSLOTS * 16 + 1023) / 1024 * 1024;
constexpr int WROW = 240, WCH = WROW / 16;          // weight row: 7 x 32 B + 16 B pad
constexpr int WSLOTS = 64 * WCH;                    // 960
constexpr int W_BYTES = WSLOTS * 16;
constexpr int CT_BYTES = 256 * 64;                  // conv tile: 256 pixel rows x 64 int8 channels
constexpr int LDS_TOTAL = 2 * PATCH_BYTES + W_BYTES + CT_BYTES;      // two patch slots
}

__global__ void __launch_bounds__(512) stem_pool_kernel(const StemPoolArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[LDS_TOTAL];
    char* const wl = lds + 2 * PATCH_BYTES;
    char* const ct = lds + 2 * PATCH_BYTES + W_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
    const int l31 = lane & 31, lh = lane >> 5;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
    const int tq_n = a.Q / TQ, tp_n = a.P / TP;
    const int ntiles = a.N * tp_n * tq_n;

    // PERSISTENT workgroups: weights and biases are fetched once, then the workgroup walks tiles blockIdx, +grid, ... with the
    // next tile's input patch already in flight (two patch slots) while the current one is multiplied, pooled and stored.
    const int pix = wave * 32 + l31;                                 // region pixel of this lane (255 = padding lane)
    const int pixc = pix < RPX ? pix : RPX - 1;
    const int ri = pixc / RW, rj = pixc - ri * RW;
    v4i bq[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[i][g] = *(const v4i*)(a.bias + i * 32 + 8 * g + 4 * lh);
    asm volatile("" ::: "memory");
    {   // weights: slot -> (cout, 16-byte piece); piece 14 of a row is padding (zeros from the range check)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int s = tid + i * 512;
            const int co = s / WCH, pc = s - co * WCH;
            const unsigned off = (s < WSLOTS && pc < 14) ? (unsigned)(co * 224 + pc * 16) : kOOB;
            if ((i * 512 + wave * 64) < WSLOTS)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(wl + i * 8192 + wave * 1024), 16, off, 0, 0, 0);
        }
    }
    const bool dma_wave = wave * 64 < PSLOTS;                        // waves 0..5 carry the patch DMA (one instruction per tile)
    const int ppr = tid / PCH, ppc = tid - ppr * PCH;                // this thread's patch slot: (row, 4-pixel chunk)
    auto tile_of = [&](int t, int* n, int* tp, int* tq) { *tq = t % tq_n; const int r = t / tq_n; *tp = r % tp_n; *n = r / tp_n; };
    auto issue_patch = [&](int t, int slot) {
        // the haloed image carries 2 extra halo pixels on every side (a.org): every chunk of every tile is real memory, 16-byte aligned
        int n, tp, tq; tile_of(t, &n, &tp, &tq);
        const int hr = 2 * (2 * TP * tp - 1) + a.org + ppr, wc = 2 * (2 * TQ * tq - 1) + a.org + ppc * 4;
        unsigned off = kOOB;
        if (tid < PSLOTS && hr >= 0 && hr < a.Hp && wc >= 0 && wc + 4 <= a.Wp) off = (unsigned)((((size_t)n * a.Hp + hr) * a.Wp + wc) * 4);
        if (dma_wave)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(lds + slot * PATCH_BYTES + wave * 1024), 16, off, 0, 0, 0);
    };
    const int t0 = blockIdx.x, step = gridDim.x;
    if (t0 < ntiles) issue_patch(t0, 0);
    // ReLU before a requant whose clamp starts at 0 is absorbed by the clamp (requant is monotone, requant(v <= 0) <= 0):
    // the max below then only runs for formats with a negative lower bound
    const int floor0 = (a.relu0 && a.qlo < 0) ? 0 : INT32_MIN;
    const char* wrow = wl + l31 * WROW + lh * 16;

    int it = 0;
    for (int t = t0; t < ntiles; t += step, ++it) {
        const int cur = it & 1;
        const bool more = t + step < ntiles;
        if (more) issue_patch(t + step, cur ^ 1);                    // that slot's tile was consumed before the previous epilogue barrier
        // patch(t) (and, the first time, the weights) landed: only what this wave issued AFTER it may stay in flight —
        // the next patch (waves 0..5) and the previous tile's pooled store (waves 0..3)
        const int newer = ((more && dma_wave) ? 1 : 0) + ((it > 0 && wave < 4) ? 1 : 0);
        if (newer == 0) wait_vmcnt<0>(); else if (newer == 1) wait_vmcnt<1>(); else wait_vmcnt<2>();
        __builtin_amdgcn_s_barrier();

        int n, tp, tq; tile_of(t, &n, &tp, &tq);
        const int cp0 = 2 * TP * tp - 1, cq0 = 2 * TQ * tq - 1;     // conv pixel of region (0,0); -1 = above / left of the image
        // ---- conv: 7 K steps (kernel rows); B fragment = 16 of the 32 row bytes of this lane's pixel (two 8-byte reads:
        //      the stride-2 pixel pitch makes odd columns 8-byte aligned only)
        const char* xrow = lds + cur * PATCH_BYTES + ((2 * ri) * PWD + 2 * rj + 4 * lh) * 4;
        v16i acc[2];                                                 // accumulators start at the bias: no add in the epilogue
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = bq[i][r >> 2][r & 3];
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            const v2i x0 = *(const v2i*)(xrow + r * PWD * 4), x1 = *(const v2i*)(xrow + r * PWD * 4 + 8);
            const v4i xf = {x0.x, x0.y, x1.x, x1.y};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const v4i wf = *(const v4i*)(wrow + i * 32 * WROW + r * 32);
                acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, xf, acc[i], 0, 0, 0);
            }
        }
        // ---- epilogue -> int8 conv tile in LDS; conv pixels outside the image hold unsigned 0 (plain bytes: the pool
        //      compares them as unsigned; the HBM copy is biased at the very end)
        const int cp = cp0 + ri, cq = cq0 + rj;
        const bool inside = pix < RPX && cp >= 0 && cp < a.Pc && cq >= 0 && cq < a.Qc;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned d[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = requant1(max(acc[i][4 * g + e], floor0), a.qn, a.qlo, a.qhi);
                d[g] = inside ? pack4(y[0], y[1], y[2], y[3]) : 0u;
            }
            auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
            v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
            *(v4i*)(ct + pix * 64 + i * 32 + 16 * lh) = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        // ---- 3x3 / stride 2 max-pool from the tile: thread = (pooled pixel, 16 channels)
        if (tid < TP * TQ * 4) {
            const int pp = tid >> 2, c16 = tid & 3;
            const int pr = pp / TQ, pc = pp - pr * TQ;
            us2 ev[4], od[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { ev[k] = us2{0, 0}; od[k] = us2{0, 0}; }
#pragma unroll
            for (int dr = 0; dr < 3; ++dr)
#pragma unroll
                for (int dc = 0; dc < 3; ++dc) {
                    const v4i v = *(const v4i*)(ct + ((2 * pr + dr) * RW + 2 * pc + dc) * 64 + c16 * 16);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned u = (unsigned)v[k];
                        ev[k] = __builtin_elementwise_max(ev[k], __builtin_bit_cast(us2, u & 0x00ff00ffu));
                        od[k] = __builtin_elementwise_max(od[k], __builtin_bit_cast(us2, (u >> 8) & 0x00ff00ffu));
                    }
                }
            v4i o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (int)((__builtin_bit_cast(unsigned, ev[k]) | (__builtin_bit_cast(unsigned, od[k]) << 8)) ^ a.bias_xor);
            const size_t m = ((size_t)n * a.P + TP * tp + pr) * a.Q + TQ * tq + pc;
            *(v4i*)(a.out + m * 64 + c16 * 16) = o;
        }
        // the next iteration's first barrier separates these ct reads from the next epilogue's ct writes
    }
}

bool stem_pool_supported(int cin, int cout, int k, int stride, int pad, int pool_k, int pool_s, int pool_p, int P, int Q) {
    static const int on = [] { const char* e = getenv("F8_FUSE_STEM"); return e ? atoi(e) : 1; }();
    return on && cin <= 4 && cout == 64 && k == 7 && stride == 2 && pad == 3 && pool_k == 3 && pool_s == 2 && pool_p == 1 && P > 0 && Q > 0 &&
           P % TP == 0 && Q % TQ == 0;
}

hipError_t launch_stem_pool(const StemPoolArgs& a, hipStream_t s) {
    const int ntiles = a.N * (a.P / TP) * (a.Q / TQ);
    static const int wpc = [] { const char* e = getenv("F8_STEM_WPC"); return e ? atoi(e) : 3; }();     // resident workgroups per CU (44 KB LDS each)
    static int ncu = 0;
    if (!ncu) { int dev = 0; hipDeviceProp_t p; ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256; }
    const int grid = ntiles < ncu * wpc ? ntiles : ncu * wpc;
    hipLaunchKernelGGL(stem_pool_kernel, dim3(grid), dim3(512), 0, s, a);
    return hipGetLastError();
}

}  // namespace f8
