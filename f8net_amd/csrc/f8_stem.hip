// f8_stem.hip — ResNet head in one launch: 7x7 / stride 2 / pad 3 convolution (Cin <= 4, 64 couts) + ReLU + requant to the
// next layer's unsigned 8-bit format + 3x3 / stride 2 / pad 1 max-pool (gfx950 only).
//
// Unfused, the 112x112x64 conv output is written (103 MB per 128 images) and read back by the pool; here it only ever
// exists as a 15x17-pixel int8 tile in LDS.  A workgroup owns 7x8 POOLED pixels = a 15x17 region of conv pixels
// (255 = 8 MFMA pixel tiles, one per wave; 14 % of the conv is recomputed at tile seams):
//   * the 35x40-pixel NHWC4 input patch (5.6 KB) and ALL weights (64 couts x 7 kernel rows x 32 B, rows padded to
//     240 B for conflict-free fragment reads) arrive by LDS-direct DMA once; one kernel row of 8 pixels x 4 channels is
//     one 32-byte K step, so the conv is 7 MFMA steps per tile with no barrier in between;
//   * per 32-channel half: bias + ReLU'd int32 accumulators -> LDS tile (conv pixels outside the image hold the identity of
//     max: 0 after a ReLU, INT32_MIN otherwise; every pool window has an in-image tap), 3x3 max over int32, THEN the
//     requantisation(s) on the 56 pooled pixels instead of the 255 conv pixels (requant is monotone, so it commutes with
//     max; pooling first costs 2.7x fewer vector instructions) -> int32 (I32T) and / or int8 outputs.
// Arithmetic: the stem of conv_igemm_kernel + maxpool_kernel, bit for bit (reference: fix_resnet.py:354-359; the float
// MaxPool detour there is exact; requant commutes with max because it is monotone).
#include "f8_device.h"
#include <cstdlib>

namespace f8 {

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));

namespace {
// rint(x * scale) clamped (fix_train.py:683-692 through input_kernel's quant_in)
__device__ __forceinline__ int quant_in_stem(float x, float scale, int lo, int hi) {
    const float r = rintf(__fmul_rn(x, scale));      // one IEEE multiply, as quant_in (f8_kernels.hip)
    return (int)fminf(fmaxf(r, (float)lo), (float)hi);
}
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
constexpr int CT_PITCH = 144;                       // one conv-tile row: 32 int32 channels + 16 B pad (conflict-free column access)
constexpr int CT_BYTES = 256 * CT_PITCH;            // conv tile of ONE 32-channel half
constexpr int LDS_TOTAL = 2 * PATCH_BYTES + W_BYTES + CT_BYTES;      // two patch slots
}

template <int KIND>
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
    // KIND >= 0: the patch is built from the raw NCHW input instead (f8_net_run / _f32 / _u8 without an input launch): this thread's
    // slot = 4 consecutive pixels of one patch row, 3 planes -> 12 element loads one tile ahead (registers), converted, packed to
    // NHWC4 bytes and stored to the other patch slot at the end of the iteration.  Out-of-image elements load 0 through the buffer
    // range check = the (biased) zero of the padding.
    const __amdgpu_buffer_rsrc_t rraw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(KIND == 0 ? (const void*)a.xi : KIND == 1 ? (const void*)a.xf : (const void*)a.xu8), 0,
        (unsigned)((size_t)a.N * a.rC * a.rH * a.rW * (KIND == 2 ? 1 : 4)), 0x00020000);
    int rawv[KIND >= 0 ? 12 : 1];
    unsigned bad = 0;                                    // KIND 0: an int32 input value outside the head's 8-bit format was seen
    unsigned rawok = 0;                                  // bit j: column j of the slot lies inside the image (and so does the row)
    auto load_raw = [&](int t) {
        if constexpr (KIND >= 0) {
            int n, tp, tq; tile_of(t, &n, &tp, &tq);
            const int row = 2 * (2 * TP * tp - 1) - 3 + ppr, col0 = 2 * (2 * TQ * tq - 1) - 3 + ppc * 4;
            const bool rok = tid < PSLOTS && row >= 0 && row < a.rH;
            rawok = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) rawok |= (rok && col0 + j >= 0 && col0 + j < a.rW) ? 1u << j : 0u;
            const unsigned plane = (unsigned)(a.rH * a.rW);
            // element (n, c, row, col0): a slot that runs over the END of its row reads into the next row (masked below; beyond the last
            // element of the buffer the range check returns 0)
            const unsigned e0 = (unsigned)((n * a.rC * a.rH + row) * a.rW + col0);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const unsigned e = rawok && c < a.rC ? e0 + (unsigned)c * plane : kOOB;
                if constexpr (KIND == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) rawv[c * 4 + j] = (int)__builtin_amdgcn_raw_buffer_load_b8(rraw, e == kOOB ? kOOB : e + j, 0, 0);
                } else if (tq == 0) {                    // wave-uniform: the image's left edge — a slot may START before its row (and, on
                                                         // the first row of the first image, before the buffer): element loads
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        rawv[c * 4 + j] = __builtin_amdgcn_raw_buffer_load_b32(rraw, (e != kOOB && ((rawok >> j) & 1u)) ? (e + j) * 4u : kOOB, 0, 0);
                } else {
                    const v4i q4 = __builtin_amdgcn_raw_buffer_load_b128(rraw, e == kOOB ? kOOB : e * 4u, 0, 0);      // 4-byte aligned is enough
#pragma unroll
                    for (int j = 0; j < 4; ++j) rawv[c * 4 + j] = q4[j];
                }
            }
        }
    };
    auto store_raw = [&](int slot) {
        if constexpr (KIND >= 0) {
            int v[3][4];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = rawv[c * 4 + j];
                    int q;
                    if constexpr (KIND == 0) { q = r; bad |= (unsigned)(r - a.chk_lo) > (unsigned)(a.chk_hi - a.chk_lo) ? 1u : 0u; }
                    else if constexpr (KIND == 1) q = quant_in_stem(__builtin_bit_cast(float, r), a.scale, a.qlo, a.qhi);
                    else q = (int)a.lut[c * 256 + (r & 0xff)];
                    v[c][j] = ((rawok >> j) & 1u) && c < a.rC ? q : 0;
                }
            v4i o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (int)(pack4(v[0][j], v[1][j], v[2][j], 0) ^ a.xor8);
            if (tid < PSLOTS) *(v4i*)(lds + slot * PATCH_BYTES + tid * 16) = o;
        }
    };
    const int t0 = blockIdx.x, step = gridDim.x;
    if constexpr (KIND < 0) { if (t0 < ntiles) issue_patch(t0, 0); }
    else if (t0 < ntiles) { load_raw(t0); store_raw(0); }
    const int floor0 = a.relu0 ? 0 : INT32_MIN;
    const char* wrow = wl + l31 * WROW + lh * 16;

    int it = 0;
    for (int t = t0; t < ntiles; t += step, ++it) {
        const int cur = it & 1;
        const bool more = t + step < ntiles;
        if constexpr (KIND < 0) {
            if (more) issue_patch(t + step, cur ^ 1);                // that slot's tile was consumed before the previous epilogue barrier
            // patch(t) (and, the first time, the weights) landed: only the next patch (waves 0..5), issued after it, may stay in
            // flight; the previous tile's output stores are waited for as well (a handful of small stores per wave)
            if (more && dma_wave) wait_vmcnt<1>(); else wait_vmcnt<0>();
        } else {
            __builtin_amdgcn_sched_barrier(0);
            if (more) { load_raw(t + step); __builtin_amdgcn_sched_barrier(0); wait_vmcnt<(KIND == 2 ? 12 : 3)>(); }   // only the next tile's loads stay in flight
            else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this thread's patch stores of the previous iteration
        }
        __builtin_amdgcn_s_barrier();

        int n, tp, tq; tile_of(t, &n, &tp, &tq);
        const int cp0 = 2 * TP * tp - 1, cq0 = 2 * TQ * tq - 1;     // conv pixel of region (0,0); -1 = above / left of the image
        // ---- conv: 7 K steps (kernel rows); B fragment = 16 of the 32 row bytes of this lane's pixel (two 8-byte reads:
        //      the stride-2 pixel pitch makes odd columns 8-byte aligned only)
        const char* xrow = lds + cur * PATCH_BYTES + ((2 * ri) * PWD + 2 * rj + 4 * lh) * 4;
        v16i acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0;
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
        // ---- per 32-channel half: accumulators (+ bias, ReLU) -> int32 tile in LDS -> 3x3 / stride 2 max -> outputs
        const int cp = cp0 + ri, cq = cq0 + rj;
        const bool inside = pix < RPX && cp >= 0 && cp < a.Pc && cq >= 0 && cq < a.Qc;
        const int ident = a.relu0 ? 0 : INT32_MIN;                   // identity of max for conv pixels outside the image
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h) __builtin_amdgcn_s_barrier();                     // half 0's pool reads are done before the tile is rewritten
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v4i o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = inside ? max((int)((unsigned)acc[h][4 * g + e] + (unsigned)bq[h][g][e]), floor0) : ident;
                *(v4i*)(ct + pix * CT_PITCH + (8 * g + 4 * lh) * 4) = o;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (tid < TP * TQ * 8) {                                  // thread = (pooled pixel, 4 channels)
                const int pp = tid >> 3, c4 = tid & 7;
                const int pr = pp / TQ, pc = pp - pr * TQ;
                v4i mx = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
#pragma unroll
                for (int dr = 0; dr < 3; ++dr)
#pragma unroll
                    for (int dc = 0; dc < 3; ++dc) {
                        const v4i v = *(const v4i*)(ct + ((2 * pr + dr) * RW + 2 * pc + dc) * CT_PITCH + c4 * 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e) mx[e] = max(mx[e], v[e]);
                    }
                const int m = (n * a.P + TP * tp + pr) * a.Q + TQ * tq + pc;
                const int c = h * 32 + c4 * 4;
                if (a.out32) *(v4i*)(a.out32 + i32t_index(m, c, 64)) = mx;
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (a.q[k].ptr)
                        *(unsigned*)(a.q[k].ptr + (size_t)m * 64 + c) =
                            pack4(requant1(mx[0], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(mx[1], a.q[k].n, a.q[k].lo, a.q[k].hi),
                                  requant1(mx[2], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(mx[3], a.q[k].n, a.q[k].lo, a.q[k].hi)) ^ a.q[k].bias_xor;
            }
        }
        // the next iteration's first barrier separates these ct reads from the next epilogue's ct writes
        if constexpr (KIND >= 0) { if (more) store_raw(cur ^ 1); }   // that slot's tile was consumed before this iteration's first barrier
    }
    if constexpr (KIND == 0) {                           // the int32 input is NARROWED to the head's 8-bit format: values outside it would wrap silently
        if (a.err && bad) atomicOr(a.err, 1u);
    }
}

bool stem_pool_supported(int cin, int cout, int k, int stride, int pad, int pool_k, int pool_s, int pool_p, int P, int Q) {
    return cin <= 4 && cout == 64 && k == 7 && stride == 2 && pad == 3 && pool_k == 3 && pool_s == 2 && pool_p == 1 && P > 0 && Q > 0 &&
           P % TP == 0 && Q % TQ == 0;
}

hipError_t launch_stem_pool(const StemPoolArgs& a, hipStream_t s) {
    const int ntiles = a.N * (a.P / TP) * (a.Q / TQ);
    const int wpc = a.wpc > 0 ? a.wpc : 2;               // resident workgroups per CU (63 KB LDS each), Options::stem_wpc
    static int ncu = 0;
    if (!ncu) { int dev = 0; hipDeviceProp_t p; ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256; }
    const int grid = ntiles < ncu * wpc ? ntiles : ncu * wpc;
    switch (a.raw_kind) {
        case 0: hipLaunchKernelGGL(stem_pool_kernel<0>, dim3(grid), dim3(512), 0, s, a); break;
        case 1: hipLaunchKernelGGL(stem_pool_kernel<1>, dim3(grid), dim3(512), 0, s, a); break;
        case 2: hipLaunchKernelGGL(stem_pool_kernel<2>, dim3(grid), dim3(512), 0, s, a); break;
        default: hipLaunchKernelGGL(stem_pool_kernel<-1>, dim3(grid), dim3(512), 0, s, a); break;
    }
    return hipGetLastError();
}

}  // namespace f8
