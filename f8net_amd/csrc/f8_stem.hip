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

// ---------------------------------------------------------------------------------------------------------------------------------
// stem_rows_kernel — the same head, organised so that the pool never goes through LDS and the input is always in flight.
//
// stem_pool_kernel above parks every conv tile in LDS as int32 (73 KB written, 129 KB read per 56 pooled pixels) behind four
// barriers per tile, and keeps three 16-byte loads per thread in flight: 80 us for 103 - 180 MB.  Here
//   * a COMPUTE wave owns a strip of 28 pooled columns and walks down its rows: per conv row it multiplies two 32-pixel MFMA
//     tiles — E: the even conv columns 2c, O: the odd ones 2c - 1 (c = the lane's pooled column) — so the pool's horizontal window
//     {2c-1, 2c, 2c+1} is max(O[lane], E[lane], O[lane+1]): one DPP lane shift, no memory; the vertical window is a running max
//     over three consecutive conv rows in registers (the last one is carried into the next pooled row).  Bias rides in the
//     accumulators' start value, ReLU and the requantisations run on the POOLED values (monotone: they commute with max).
//     Conv pixels outside the image (column -1, row -1: the pool's padding) are replaced by a pixel of the same window that lies
//     inside (column 1 / row 1) — max over a window with a duplicate is the max over the window: no masks.  The wave's 7 weight
//     fragments (32 couts x 7 kernel rows x 32 B) live in registers for its whole life; the only LDS traffic is the B
//     operand: one 16-byte read (O) / two 8-byte reads (E) per lane and kernel row from the band's input patch (NHWC4 bytes);
//   * LOADER waves bring in the NEXT band's input rows meanwhile — every load of a band in flight at once (24 x 16 B per thread,
//     ~100 KB per CU: what the HBM latency asks for; registers the compute waves could not spare) — converted from the raw network
//     input (int32 / fp32 / uint8 NCHW planes, as stem_pool_kernel's KIND) or copied from the haloed NHWC4 form into the other
//     patch buffer.  One barrier per band.
// Workgroup = 768 threads = 8 compute waves (32-cout half x 2 strips x 2 half-bands; pooled width <= 28: 1 strip x 4 quarter-bands:
// two per SIMD, 28 weight registers each, so one wave's maxima and stores run under the other's multiplies; the two halves of a
// strip read the same B fragments — LDS has the room) + 4 loader waves, persistent, one per CU; band = 7 pooled rows x the full width of one image (35 input rows, 32 KB per patch buffer); 7 of a
// band's 35 rows are shared with the band above (L2: the bands of one image run on one XCD).
#ifndef F8_STEM_DB
#define F8_STEM_DB 3                                  // B fragments in flight per compute wave (tuning builds override)
#endif
// tuning builds (results INVALID): what the ResNet head is bound by — 1: the B fragments are not read from LDS, 2: no horizontal-pool vector work
// per conv row, 3: the loader waves load nothing after the first band, 4: no output stores (profiles/stem_limiter_r06.md)
#ifndef F8_STEM_ABL
#define F8_STEM_ABL 0
#endif
namespace {
constexpr int RB = 7;                               // pooled rows per band
constexpr int SW = 28;                              // pooled columns per strip (lane 28 of a strip only provides O for lane 27)
constexpr int RB_ROWS = 4 * RB + 7;                 // input rows of a band
constexpr int H2_RB = 14;                           // MobileNet-V2 head: output rows per band (2 * 14 + 5 = 33 input rows)

__device__ __forceinline__ int dpp_next_lane(int v) {   // lane i <- lane i + 1 (across the 16-lane DPP rows; lane 63 keeps its value)
    return __builtin_amdgcn_update_dpp(v, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}
template <class Y>
__device__ __forceinline__ v4i stem_quant16(const Y& y, int n, int lo, int hi, unsigned x_or, bool acc_ok) {
    unsigned d[4];
    if (acc_ok && n > 0 && n <= kRequantU8MaxShift && lo == 0 && hi == 255) {   // the usual case: unsigned 8-bit, three VALU operations per value (wave-uniform branch)
        const float sc = requant_u8_scale(n);
#pragma unroll
        for (int g = 0; g < 4; ++g) d[g] = requant_u8x4(y[4 * g], y[4 * g + 1], y[4 * g + 2], y[4 * g + 3], sc) ^ x_or;
    } else if (n > 0 && n <= 30 && lo == 0 && hi == 255) {   // the same case on INTEGER instructions (option requant_float = 0 — the default since round 5 — or an
                                                     // accumulator the planner cannot bound): v_bfe_u32, v_add3_u32, v_ashr_pk_u8_i32: 2 3/4 per value, exact for every int32.
                                                     // (Round 5: the integer default used to fall through to the general form below, 4 3/4 per value.)
#pragma unroll
        for (int g = 0; g < 4; ++g) d[g] = requant_u8x4_int(y[4 * g], y[4 * g + 1], y[4 * g + 2], y[4 * g + 3], n) ^ x_or;
    } else if (n > 0) {                              // another right shift: four
        const unsigned hf = 1u << (n - 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) d[g] = pack4(requant_shr(y[4 * g], n, hf, 0u, lo, hi), requant_shr(y[4 * g + 1], n, hf, 0u, lo, hi),
                                                 requant_shr(y[4 * g + 2], n, hf, 0u, lo, hi), requant_shr(y[4 * g + 3], n, hf, 0u, lo, hi)) ^ x_or;
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) d[g] = pack4(requant1(y[4 * g], n, lo, hi), requant1(y[4 * g + 1], n, lo, hi), requant1(y[4 * g + 2], n, lo, hi),
                                                 requant1(y[4 * g + 3], n, lo, hi)) ^ x_or;
    }
    auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
    auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
    const v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
    return o;                                        // lane (pixel, half): channels 16 half .. 16 half + 15 of the 32-channel tile
}
}

// H2: the same skeleton (loader waves, double-buffered patch, persistent bands) with a different body for the compute waves: the
// MobileNet-V2 head — 3x3 / 2 conv (3 -> 32, ReLU) -> depthwise 3x3 (ReLU) -> 1x1 (32 -> <= 32) — see the block in front of that body.
template <int KIND, bool H2 = false>
__global__ void __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3))) stem_rows_kernel(const StemPoolArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];      // 2 x [RB_ROWS][PWB pixels][4 B] (patch column pc = input column pc - 5) | 64 biases
    set_fp_round_nearest_even();                                    // stem_quant16 / quant_row may take the float-converter form (f8_device.h)
    const int tid = threadIdx.x;
    const int lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0 .. 7 compute, 8 .. 11 loaders
    const int QW = H2 ? a.rW >> 2 : a.Q;                            // input width / 4
    const int PWB = 4 * QW + 8, ROWB = PWB * 4, SPR = PWB / 4;      // patch row: pixels, bytes, 16-byte slots
    const int PBUF = RB_ROWS * ROWB;
    constexpr int RBK = H2 ? H2_RB : RB;                            // output rows per band
    const int bands = (a.P + RBK - 1) / RBK, ntiles = a.N * bands;
    // XCD-aware order: dispatch slot d (d % 8 = the XCD of a persistent workgroup's every slot) -> band tile; the bands of one image,
    // which share input rows, run on one XCD
    auto tile_of = [&](int d) {
        const int xcd = d & 7, qq = ntiles >> 3, rr = ntiles & 7;
        return (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (d >> 3);
    };
    // loaders' unit: one 16-byte SLOT = 4 pixels of one input row.  Raw planes: slots are image-aligned (columns 4j .. 4j + 3: whole
    // inside the image, 16-byte aligned in memory, so every load is one unconditional b128 / b32 — no branch, hence no wait, between
    // the loads of a band) and land 5 pixels to the right in the patch (four dword stores); the patch columns left and right of the
    // image are written once, below.  Haloed form: slots are patch-aligned (a plain copy).
    const int SPI = KIND < 0 ? SPR : QW;                            // slots per input row (raw: rW / 4)
    struct Band { int n, p0, rp, r0, nslot; };
    auto band_of = [&](int d) {
        const int t = tile_of(d);
        Band B;
        B.n = t / bands; B.p0 = (t - B.n * bands) * RBK;
        B.rp = (a.P - B.p0) < RBK ? (a.P - B.p0) : RBK;             // output rows of this band
        if constexpr (H2) { B.r0 = 2 * B.p0 - 3; B.nslot = (2 * B.rp + 5) * SPI; }     // conv rows p0 - 1 .. p0 + rp: input rows 2 p0 - 3 .. 2 (p0 + rp) + 1
        else { B.r0 = 4 * B.p0 - 5; B.nslot = (4 * B.rp + 7) * SPI; }                  // input rows r0 .. r0 + 4 rp + 6
        return B;
    };
    constexpr int NR = KIND < 0 ? 4 : (KIND == 2 ? 3 : 12);
    constexpr int LSLOTS = 8;                           // slots per loader thread and pass, all in flight (35 rows x 56 slots / 256 threads = 7.7)
    const __amdgpu_buffer_rsrc_t rsrc = KIND < 0 ? __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc((void*)(KIND == 0 ? (const void*)a.xi : KIND == 1 ? (const void*)a.xf : (const void*)a.xu8), 0,
                                            (unsigned)((size_t)a.N * a.rC * a.rH * a.rW * (KIND == 2 ? 1 : 4)), 0x00020000);
    const unsigned plane = (unsigned)(a.rH * a.rW);
    unsigned bad = 0;                                    // KIND 0: an int32 input value outside the head's 8-bit format was seen
    auto slot_issue = [&](const Band& B, int sl, int (&raw)[NR], unsigned& ok) {
        const int pr = sl / SPI, j = sl - pr * SPI;
        if constexpr (KIND < 0) {
            // haloed row / column = input row / column + 3 + org; org = 2 (stem_rows_ok): patch slot j = haloed pixels 4j .. 4j + 3
            const int hr = B.r0 + pr + 5, wc = 4 * j;
            ok = (sl < B.nslot && hr >= 0 && hr < a.Hp && wc + 4 <= a.Wp) ? 1u : 0u;
            const v4i v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? (unsigned)((((size_t)B.n * a.Hp + hr) * a.Wp + wc) * 4) : kOOB, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) raw[q] = v[q];
        } else {
            const int row = B.r0 + pr;
            ok = (sl < B.nslot && row >= 0 && row < a.rH) ? 1u : 0u;
            const unsigned e0 = (unsigned)((B.n * a.rC * a.rH + row) * a.rW + 4 * j);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const unsigned e = (ok && c < a.rC) ? e0 + (unsigned)c * plane : kOOB;
                if constexpr (KIND == 2) raw[c] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, e, 0, 0);
                else {
                    const v4i q4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, e == kOOB ? kOOB : e * 4u, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) raw[c * 4 + q] = q4[q];
                }
            }
        }
    };
    auto slot_commit = [&](char* buf, const Band& B, int sl, const int (&raw)[NR], unsigned ok) {
        const int pr = sl / SPI, j = sl - pr * SPI;
        v4i o;
        if constexpr (KIND < 0) {
            const int z = (int)a.xor8;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = ok ? raw[q] : z;
            if (sl < B.nslot) *(v4i*)(buf + sl * 16) = o;
        } else {
            int v[3][4];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int x;
                    if constexpr (KIND == 0) { x = raw[c * 4 + q]; bad |= (ok && c < a.rC && (unsigned)(x - a.chk_lo) > (unsigned)(a.chk_hi - a.chk_lo)) ? 1u : 0u; }
                    else if constexpr (KIND == 1) x = quant_in_stem(__builtin_bit_cast(float, raw[c * 4 + q]), a.scale, a.qlo, a.qhi);
                    else x = (int)((const short*)(lds + 2 * PBUF + 512))[c * 256 + ((raw[c] >> (8 * q)) & 0xff)];
                    v[c][q] = (ok && c < a.rC) ? x : 0;         // rows outside the image: (biased) zero
                }
            if (sl < B.nslot) {
                int* const dst = (int*)(buf + pr * ROWB + (4 * j + 5) * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q] = (int)(pack4(v[0][q], v[1][q], v[2][q], 0) ^ a.xor8);
            }
        }
    };
    // a band's rows -> `buf`, by `nthr` threads (this one is number `t`): LSLOTS slots per thread in flight
    auto load_band = [&](const Band& B, char* buf, int t, int nthr) {
        for (int s0 = t; s0 < B.nslot; s0 += LSLOTS * nthr) {
            int raw[LSLOTS][NR]; unsigned ok[LSLOTS];
#pragma unroll
            for (int u = 0; u < LSLOTS; ++u) slot_issue(B, s0 + u * nthr, raw[u], ok[u]);
#pragma unroll
            for (int u = 0; u < LSLOTS; ++u) slot_commit(buf, B, s0 + u * nthr, raw[u], ok[u]);
        }
    };
    if constexpr (KIND == 2) {   // the u8 -> head-format table: LDS (a dynamically indexed kernel argument would live in scratch)
        for (int i = tid; i < 3 * 256; i += 768) ((short*)(lds + 2 * PBUF + 512))[i] = a.lut[i];
        __syncthreads();
    }
    if constexpr (KIND >= 0) {   // patch columns outside the image (5 on the left, 3 + padding on the right) of both buffers: biased zero, once
        const int nside = PWB - 4 * QW;                             // 8
        for (int i = tid; i < 2 * RB_ROWS * nside; i += 768) {
            const int r = i / nside, c = i - r * nside;
            *(unsigned*)(lds + r * ROWB + (c < 5 ? c : 4 * QW + c) * 4) = a.xor8;
        }
    }

    const int G = gridDim.x;
    int d = blockIdx.x;
    if (d >= ntiles) return;
    load_band(band_of(d), lds, tid, 768);                           // the first band: every wave loads
    if constexpr (H2) { if (tid < 96) *(int*)(lds + 2 * PBUF + tid * 4) = tid < 32 ? a.bias[tid] : tid < 64 ? a.bd[tid - 32] : a.b1[tid - 64]; }
    else if (tid < 64) *(int*)(lds + 2 * PBUF + tid * 4) = a.bias[tid];
    __syncthreads();

    if (wave >= 8) {
        // =================================================== loader waves: band it + 1 -> the other patch while band it is multiplied
        for (int it = 0; d < ntiles; d += G, ++it) {
            if (F8_STEM_ABL != 3 && d + G < ntiles) load_band(band_of(d + G), lds + ((it & 1) ^ 1) * PBUF, tid - 512, 256);
            __syncthreads();
        }
        if constexpr (KIND == 0) { if (a.err && bad) atomicOr(a.err, 1u); }
        return;
    }
    if constexpr (KIND == 0) { if (a.err && bad) atomicOr(a.err, 1u); }   // (the first band's share of the check)

    if constexpr (H2) {
        // ======================================================= compute waves, MobileNet-V2 head: (strip of 28 columns, half-band of 7 rows)
        // The three convs run on ONE register file per wave, row by row, with nothing but the input patch in LDS:
        //   * head conv 3x3 / 2 (`ref:models/fix_mobilenet_v2.py` head: 3 -> 32, ReLU): lane l <-> conv column c0 - 1 + l; a kernel row is
        //     16 bytes (4 pixels x 4 channels, the 4th pixel's weights are zero), two kernel rows make one 32-byte K step: TWO MFMAs
        //     per conv row; requantised (3 operations, requant_u8x4) and turned by the permlane swap into 16 channels per lane half — which IS the
        //     B operand of a K = 32-channel MFMA step;
        //   * depthwise 3x3 (ReLU): nine MFMAs with diagonal weight fragments (f8_dwmma.hip); horizontal taps = the conv row fragment and
        //     two DPP lane shifts of it, vertical taps = the last three conv rows, sliding; its padding (conv column -1 / 112, conv row
        //     -1 / 112) is the biased zero, written over the lanes / rows that fall outside;
        //   * 1x1 (32 -> <= 32, no ReLU): its B operand is the depthwise row after the same requant + swap: ONE MFMA.
        const int strip = wave & 3, sb = wave >> 2;
        const int cq = strip * SW - 1 + l31;                         // conv column of this lane = depthwise input column
        const bool cq_in = cq >= 0 && cq < a.Qc;
        const int cqa = cq < 0 ? 0 : (cq > a.Qc ? a.Qc : cq);       // for addresses only
        const unsigned offc = (unsigned)(8 * cqa + 16);             // input column 2 cq - 1 = patch column 2 cq + 4
        const int col = strip * SW + l31;                            // output column (lanes 0 .. 27)
        const bool lane_out = l31 < SW && col < a.Q;
        const v4i wh0 = *(const v4i*)(a.w + l31 * 96 + lh * 32);                                   // kernel rows 0 | 1
        const v4i wh1 = lh == 0 ? *(const v4i*)(a.w + l31 * 96 + 64) : v4i{0, 0, 0, 0};            // kernel row 2 | nothing
        v4i wd[9];                                                   // depthwise: diagonal fragments
        {
            const bool mine = (l31 >> 4) == lh;
            const int dsel = (l31 & 15) >> 2, bsh = 8 * (l31 & 3);
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const unsigned wv = (unsigned)(unsigned char)a.wd[tp * 32 + l31];
                const int piece = mine ? (int)(wv << bsh) : 0;
                wd[tp] = v4i{dsel == 0 ? piece : 0, dsel == 1 ? piece : 0, dsel == 2 ? piece : 0, dsel == 3 ? piece : 0};
            }
        }
        const v4i w1f = *(const v4i*)(a.w1 + l31 * 32 + lh * 16);
        const char* const bl = lds + 2 * PBUF + 16 * lh;            // head | depthwise | 1x1 biases, 32 ints each
        const float sca = requant_u8_scale(a.na), scb = requant_u8_scale(a.nb);
        const v4i zq = {(int)0x80808080u, (int)0x80808080u, (int)0x80808080u, (int)0x80808080u};
        auto bias_acc = [&](int which) {
            v16i acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4i b = *(const v4i*)(bl + which * 128 + 8 * g * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[4 * g + q] = b[q];
            }
            return acc;
        };
        const bool rqi = a.rq_int != 0;                             // option requant_float = 0: integer requantisation (wave-uniform branch)
        auto quant_row = [&](const v16i& acc, float sc, int n) {  // ReLU + right shift (1 .. 16) into unsigned 8-bit, 16 channels per lane half
            unsigned dd[4];
            if (rqi) {
#pragma unroll
                for (int g = 0; g < 4; ++g) dd[g] = requant_u8x4_sel<2>(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3], n, 0.0f) ^ 0x80808080u;
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) dd[g] = requant_u8x4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3], sc) ^ 0x80808080u;
            }
            auto s0 = __builtin_amdgcn_permlane32_swap(dd[0], dd[2], false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(dd[1], dd[3], false, false);
            return v4i{(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
        };
        struct Row3 { v4i f[3]; };
        for (int it = 0; d < ntiles; d += G, ++it) {
            const Band B = band_of(d);
            const char* const patch = lds + (it & 1) * PBUF;
            const int p0 = B.p0;
            auto conv_row = [&](int cp) {                            // conv row cp -> the three horizontal tap fragments of the depthwise conv
                Row3 R;
                v4i x = zq;
                if (cp >= 0 && cp < a.Pc) {                          // wave-uniform
                    const char* const r0p = patch + (2 * cp - 1 - B.r0) * ROWB + offc;
                    const char* const rA = r0p + lh * ROWB, * const rB = r0p + 2 * ROWB;
                    const v2i a0 = *(const v2i*)rA, a1 = *(const v2i*)(rA + 8), b0 = *(const v2i*)rB, b1 = *(const v2i*)(rB + 8);
                    v16i acc = bias_acc(0);
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wh0, v4i{a0.x, a0.y, a1.x, a1.y}, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wh1, v4i{b0.x, b0.y, b1.x, b1.y}, acc, 0, 0, 0);
                    x = quant_row(acc, sca, a.na);
                    if (!cq_in) x = zq;
                }
                R.f[0] = x;
#pragma unroll
                for (int k = 0; k < 4; ++k) R.f[1][k] = dpp_next_lane(x[k]);
#pragma unroll
                for (int k = 0; k < 4; ++k) R.f[2][k] = dpp_next_lane(R.f[1][k]);
                return R;
            };
            const int rps = (B.rp + 1) / 2;
            const int pb = sb * rps, pe = (pb + rps) < B.rp ? (pb + rps) : B.rp;
            if (pb < pe) {
                Row3 R0 = conv_row(p0 + pb - 1), R1 = conv_row(p0 + pb);
                for (int p = pb; p < pe; ++p) {
                    const int P = p0 + p;
                    const Row3 R2 = conv_row(P + 1);
                    v16i acc = bias_acc(1);
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wd[kx], R0.f[kx], acc, 0, 0, 0);
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wd[3 + kx], R1.f[kx], acc, 0, 0, 0);
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wd[6 + kx], R2.f[kx], acc, 0, 0, 0);
                    const v4i xb = quant_row(acc, scb, a.nb);
                    v16i acc1 = bias_acc(2);
                    acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1f, xb, acc1, 0, 0, 0);
                    const size_t m = ((size_t)B.n * a.P + P) * a.Q + (lane_out ? col : 0);
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                        if (a.q[k].ptr) {
                            const v4i v = stem_quant16(acc1, a.q[k].n, a.q[k].lo, a.q[k].hi, a.q[k].bias_xor, !rqi);
                            if (lane_out) *(v4i*)(a.q[k].ptr + m * 32 + lh * 16) = v;
                        }
                    R0 = R1; R1 = R2;
                }
            }
            __syncthreads();                                        // patch `it` is consumed, patch `it + 1` is complete
        }
    } else {
    // ======================================================= compute waves: (cout half, strip, sub-band)
    const int half = wave & 1;
    // weights -> registers: A fragment (kernel row r): lane (cout of this wave's half, half of the row's 8 taps)
    v4i wf[7];
#pragma unroll
    for (int r = 0; r < 7; ++r) wf[r] = *(const v4i*)(a.w + (half * 32 + l31) * 224 + r * 32 + lh * 16);
    const int nstrip = a.Q > SW ? 2 : 1, nsb = 4 / nstrip;
    const int strip = nstrip == 2 ? ((wave >> 1) & 1) : 0, sb = nstrip == 2 ? (wave >> 2) : (wave >> 1);
    const int col = strip * SW + l31;                               // pooled column of this lane
    // E: conv column 2 col (lanes past the image repeat its last column: unused), O: conv column 2 col - 1 (column -1, the pool's
    // padding, is replaced by column 1 of the same window)
    const int colE = col < a.Q ? col : a.Q - 1, colO = col == 0 ? 1 : (col < a.Q ? col : a.Q);
    const unsigned offO = (unsigned)(16 * colO + 16 * lh), offE = (unsigned)(16 * colE + 8 + 16 * lh);
    const bool lane_out = l31 < SW && col < a.Q;
    const int floor0 = a.relu0 ? 0 : INT32_MIN;
    const char* const bias_l = lds + 2 * PBUF + half * 128 + 16 * lh;   // this half's 32 biases, behind the patches (kept out of the registers)

    for (int it = 0; d < ntiles; d += G, ++it) {
        const Band B = band_of(d);
        const char* const patch = lds + (it & 1) * PBUF;
        const int p0 = B.p0;

        // one conv row -> its horizontal pool maxima, handed to `sink(register, value)` one by one
        auto conv_row = [&](int cr, auto&& sink) {
            cr = cr < 0 ? 1 : cr;                                   // the row above the image: a row of the same window instead
            v16i e, o;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4i b = *(const v4i*)(bias_l + 8 * g * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) { e[4 * g + q] = b[q]; o[4 * g + q] = b[q]; }
            }
            const char* const base = patch + (2 * (cr - 2 * p0) + 2) * ROWB;
            // B fragments DB - 1 kernel rows ahead of their multiplies, no further (registers)
            constexpr int DB = F8_STEM_DB;
            v4i xo[DB], xe[DB];
            auto rd = [&](int r, v4i& o_, v4i& e_) {
                if constexpr (F8_STEM_ABL == 1) { o_ = v4i{cr + r, (int)offO, r, cr}; e_ = v4i{r, cr, (int)offE, cr ^ r}; (void)base; return; }
                o_ = *(const v4i*)(base + r * ROWB + offO);
                const v2i x0 = *(const v2i*)(base + r * ROWB + offE), x1 = *(const v2i*)(base + r * ROWB + offE + 8);
                e_ = v4i{x0.x, x0.y, x1.x, x1.y};
            };
#pragma unroll
            for (int r = 0; r < DB - 1; ++r) rd(r, xo[r], xe[r]);
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                if (r + DB - 1 < 7) rd(r + DB - 1, xo[(r + DB - 1) % DB], xe[(r + DB - 1) % DB]);
                asm volatile("" : "+v"(xo[r % DB]), "+v"(xe[r % DB]));
                e = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[r], xe[r % DB], e, 0, 0, 0);
                o = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[r], xo[r % DB], o, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if constexpr (F8_STEM_ABL == 2) sink(q, e[q] ^ o[q]);
                else sink(q, max(max(e[q], o[q]), dpp_next_lane(o[q])));
            }
        };

        // PAIR (round 6): the two conv rows of a pooled row (2P, 2P + 1) are multiplied TOGETHER — four accumulators (E / O of both rows) instead of
        // two.  A wave's MFMAs on one accumulator are a dependent chain (v_mfma_i32_32x32x32_i8: 16 passes), and with two chains per wave and two
        // compute waves per SIMD the matrix pipe sat idle between them: ablations of the LDS reads, of the pool's vector work, of the loaders and of
        // the stores each left the launch where it was (profiles/stem_limiter_r06.md).  The rows' input windows overlap (input rows 4P - 3 .. 4P + 3
        // and 4P - 1 .. 4P + 5): nine B-fragment reads feed 28 MFMAs instead of fourteen; the bias is the first MFMA's C operand (16 registers for
        // the wave's life instead of four LDS reads and 32 moves per conv row).
#ifndef F8_STEM_PAIR
#define F8_STEM_PAIR 1
#endif
        auto conv_pair = [&](int crA, const v16i& bz, auto&& sinkA, auto&& sinkB) {    // crA = 2P >= 0
            v16i eA, oA, eB, oB;
            const char* const base = patch + (2 * (crA - 2 * p0) + 2) * ROWB;
            constexpr int DB = 2;                                   // B fragments one input row ahead (three deep: 168 registers and 20 bytes of scratch; measured equal)
            v4i xo[DB], xe[DB];
            auto rd = [&](int r, v4i& o_, v4i& e_) {
                if constexpr (F8_STEM_ABL == 1) { o_ = v4i{crA + r, (int)offO, r, crA}; e_ = v4i{r, crA, (int)offE, crA ^ r}; (void)base; return; }
                o_ = *(const v4i*)(base + r * ROWB + offO);
                const v2i x0 = *(const v2i*)(base + r * ROWB + offE), x1 = *(const v2i*)(base + r * ROWB + offE + 8);
                e_ = v4i{x0.x, x0.y, x1.x, x1.y};
            };
#pragma unroll
            for (int r = 0; r < DB - 1; ++r) rd(r, xo[r], xe[r]);
#pragma unroll
            for (int r = 0; r < 9; ++r) {                           // input row r of the pair's window: kernel row r of conv row A, r - 2 of conv row B
                if (r + DB - 1 < 9) rd(r + DB - 1, xo[(r + DB - 1) % DB], xe[(r + DB - 1) % DB]);
                asm volatile("" : "+v"(xo[r % DB]), "+v"(xe[r % DB]));
                if (r == 0) { eA = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[0], xe[0], bz, 0, 0, 0); oA = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[0], xo[0], bz, 0, 0, 0); }
                else if (r < 7) { eA = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[r], xe[r % DB], eA, 0, 0, 0); oA = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[r], xo[r % DB], oA, 0, 0, 0); }
                if (r == 2) { eB = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[0], xe[r % DB], bz, 0, 0, 0); oB = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[0], xo[r % DB], bz, 0, 0, 0); }
                else if (r > 2) { eB = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[r - 2], xe[r % DB], eB, 0, 0, 0); oB = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[r - 2], xo[r % DB], oB, 0, 0, 0); }
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if constexpr (F8_STEM_ABL == 2) { sinkA(q, eA[q] ^ oA[q]); sinkB(q, eB[q] ^ oB[q]); }
                else { sinkA(q, max(max(eA[q], oA[q]), dpp_next_lane(oA[q]))); sinkB(q, max(max(eB[q], oB[q]), dpp_next_lane(oB[q]))); }
            }
        };

        const int rps = (B.rp + nsb - 1) / nsb;
        const int pb = sb * rps, pe = (pb + rps) < B.rp ? (pb + rps) : B.rp;
        if (pb < pe) {
            v16i carry;
            conv_row(2 * (p0 + pb) - 1, [&](int q, int v) { carry[q] = v; });
            v16i bz;                                                // this half's biases in accumulator layout (PAIR)
            if constexpr (F8_STEM_PAIR) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const v4i b = *(const v4i*)(bias_l + 8 * g * 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) bz[4 * g + q] = b[q];
                }
            }
            for (int p = pb; p < pe; ++p) {
                const int P = p0 + p;
                v16i pm;
                if constexpr (F8_STEM_PAIR) {
                    conv_pair(2 * P, bz, [&](int q, int v) { pm[q] = max(carry[q], v); }, [&](int q, int v) { carry[q] = v; pm[q] = max(max(pm[q], v), floor0); });
                } else {
                    conv_row(2 * P, [&](int q, int v) { pm[q] = max(carry[q], v); });
                    conv_row(2 * P + 1, [&](int q, int v) { carry[q] = v; pm[q] = max(max(pm[q], v), floor0); });
                }
                // ---- pooled row P: outputs
                const int m = (B.n * a.P + P) * a.Q + (lane_out ? col : 0);
                if (lane_out && a.out32) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const v4i v = {pm[4 * g], pm[4 * g + 1], pm[4 * g + 2], pm[4 * g + 3]};
                        *(v4i*)((char*)a.out32 + (size_t)(m >> 5) * (64 * 128) + half * 4096 + g * 1024 + lh * 512 + (m & 31) * 16) = v;
                    }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (a.q[k].ptr) {
                        const v4i v = stem_quant16(pm, a.q[k].n, a.q[k].lo, a.q[k].hi, a.q[k].bias_xor, a.acc_ok != 0 && a.rq_int == 0);
                        if (lane_out && (F8_STEM_ABL != 4 || v[0] == 0x12345678)) *(v4i*)(a.q[k].ptr + (size_t)m * 64 + half * 32 + lh * 16) = v;
                    }
            }
        }
        __syncthreads();                                            // patch `it` is consumed, patch `it + 1` is complete
    }

    }
}

// instances of the row-walking kernel: pooled width <= 56 (two strips), any height; a haloed form must carry 5 halo pixels
static bool stem_rows_ok(const StemPoolArgs& a) {
    if (a.Q < 2 || a.Q > 2 * SW || a.P < 1 || a.Qc != 2 * a.Q || a.Pc != 2 * a.P) return false;
    if (a.raw_kind < 0) return a.org == 2 && a.Wp % 4 == 0;
    return a.rW == 2 * a.Qc && a.rH == 2 * a.Pc;
}

bool stem_pool_supported(int cin, int cout, int k, int stride, int pad, int pool_k, int pool_s, int pool_p, int P, int Q, int rows, int H, int W) {
    if (!(cin <= 4 && cout == 64 && k == 7 && stride == 2 && pad == 3 && pool_k == 3 && pool_s == 2 && pool_p == 1 && P > 0 && Q > 0)) return false;
    // tile kernel | row-walking kernel (launch_stem_pool picks; the latter wants input sides that are multiples of 4: stem_rows_ok)
    return (P % TP == 0 && Q % TQ == 0) || (rows && Q >= 2 && Q <= 2 * SW && H == 4 * P && W == 4 * Q);
}

// MobileNet-V2 head (stem_rows_kernel<KIND, true>): input sides multiples of 4, at most 4 strips of 28 output columns
bool head2_supported(int H, int W) { return H >= 8 && W >= 8 && H % 4 == 0 && W % 4 == 0 && W / 2 <= 4 * SW; }

// compute units of the CURRENT device, cached per device ordinal (a process may drive several GPUs)
static int device_cus() {
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) { int v = 0; cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256; }
    return cus[dev];
}

hipError_t launch_stem_pool(const StemPoolArgs& a, hipStream_t s) {
    if (a.h2) {
        if (!head2_supported(a.rH, a.rW) || a.P != a.rH / 2 || a.Q != a.rW / 2 || a.Pc != a.P || a.Qc != a.Q || a.na < 1 || a.nb < 1 || a.na > kRequantU8MaxShift || a.nb > kRequantU8MaxShift || (!a.acc_ok && !a.rq_int) || a.out32 ||
            (a.raw_kind < 0 && !(a.org == 4 && a.Wp % 4 == 0))) return hipErrorInvalidValue;
        const int lds_bytes = 2 * RB_ROWS * (a.rW + 8) * 4 + 512 + 1536;
        const int ncu3 = device_cus();
        const int ntiles = a.N * ((a.P + H2_RB - 1) / H2_RB);
        const int gdiv = a.grid_div > 0 ? a.grid_div : 1;
        const int gmax = (ncu3 / gdiv + 7) / 8 * 8;
        const int grid = ntiles < gmax ? ntiles : gmax;
        switch (a.raw_kind) {
            case 0: hipLaunchKernelGGL((stem_rows_kernel<0, true>), dim3(grid), dim3(768), lds_bytes, s, a); break;
            case 1: hipLaunchKernelGGL((stem_rows_kernel<1, true>), dim3(grid), dim3(768), lds_bytes, s, a); break;
            case 2: hipLaunchKernelGGL((stem_rows_kernel<2, true>), dim3(grid), dim3(768), lds_bytes, s, a); break;
            default: hipLaunchKernelGGL((stem_rows_kernel<-1, true>), dim3(grid), dim3(768), lds_bytes, s, a); break;
        }
        return hipGetLastError();
    }
    if (a.rows && stem_rows_ok(a)) {
        const int lds_bytes = 2 * RB_ROWS * (4 * a.Q + 8) * 4 + 512 + 1536;   // 67 KB at 224 x 224: patches, biases, the u8 table
        const int ncu2 = device_cus();
        const int ntiles = a.N * ((a.P + RB - 1) / RB);
        // Persistent, one workgroup per CU, on ALL CUs when the launch is write-bound (int32 pooled output: ResNet-18 / 34), on HALF of
        // them otherwise (int8 output: compute-bound): measured, not assumed — with several batches in flight the other batches' launches
        // run on the rest of the chip, and on some boxes of the pool the sclk the firmware grants the launches AFTER a full-chip run of
        // this kernel is 1.6 % lower (2345 vs 2383 MHz at LOWER package power, rocm-smi; tools/clk_probe.sh).  Options::stem_grid_div
        // overrides (DESIGN.md §9 has the sweep).
        const int gdiv = a.grid_div > 0 ? a.grid_div : (a.out32 ? 1 : 2);
        const int gmax = (ncu2 / gdiv + 7) / 8 * 8;                         // a multiple of 8: tile_of's XCD arithmetic
        const int grid = ntiles < gmax ? ntiles : gmax;                     // a workgroup walks slots b, b + grid, ...
        switch (a.raw_kind) {
            case 0: hipLaunchKernelGGL(stem_rows_kernel<0>, dim3(grid), dim3(768), lds_bytes, s, a); break;
            case 1: hipLaunchKernelGGL(stem_rows_kernel<1>, dim3(grid), dim3(768), lds_bytes, s, a); break;
            case 2: hipLaunchKernelGGL(stem_rows_kernel<2>, dim3(grid), dim3(768), lds_bytes, s, a); break;
            default: hipLaunchKernelGGL(stem_rows_kernel<-1>, dim3(grid), dim3(768), lds_bytes, s, a); break;
        }
        return hipGetLastError();
    }
    if (!(a.P % TP == 0 && a.Q % TQ == 0)) return hipErrorInvalidValue;
    const int ntiles = a.N * (a.P / TP) * (a.Q / TQ);
    const int wpc = a.wpc > 0 ? a.wpc : 2;               // resident workgroups per CU (63 KB LDS each), Options::stem_wpc
    const int ncu = device_cus();
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
