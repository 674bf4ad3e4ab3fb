// f8_dwmma.hip — depthwise 3x3 (stride 1 / 2, pad 1) ON THE MATRIX CORES (gfx950).
//
// Depthwise layers (`groups == channels` convs of /root/reference/models/fix_mobilenet_v1.py:14-54 and fix_mobilenet_v2.py:20-48, quantised by
// fix_quant_ops.py:90-114 like every other conv) do not reduce over channels, so on the vector ALUs the four bytes a v_dot4 reduces
// over have to be four TAPS of one channel: dwconv3x3_dot4_kernel (f8_kernels.hip) spends 8 v_perm_b32 + 4 v_dot4 per 4-channel quad
// and tap group on that transpose and is bound by it (~16 vector operations per output; 33 % / 45 % of MobileNet-V2's / -V1's time in
// round 2).  Here a 32-channel tile of a tap is ONE v_mfma_i32_32x32x32_i8 whose weight fragment is DIAGONAL
// (A[c][k] = w[tap][c] for k == c, else 0): the B operand is 32 pixels x 32 channels exactly as NHWC memory holds them, the accumulators
// start at the bias and leave in the layout every conv epilogue of the library turns into 16-byte rows.  A wave owns 32 input columns of one
// image and one channel tile and walks DOWN the rows: one 16-byte load per lane and input row, the three horizontal taps are that row
// fragment and two DPP lane shifts of it (as in the row-walking head, f8_stem.hip), the three vertical taps are the last three row
// fragments, which slide in registers — every input byte is loaded once per strip.  Nine multiplies + ~70 vector operations per 28 x 32
// outputs.  Everything stays in the stored domain: unsigned activations are biased (x ^ 0x80), out-of-image taps are the biased zero,
// 128 * sum(w) sits in the bias (pack_dw_weights).
#include "f8_device.h"

namespace f8 {

namespace {
constexpr int DW_SW = 28;                           // output columns per strip (lanes 28 .. 31 only feed the shifts)
constexpr int DW_BAND = 14;                         // output rows per wave (8 where that leaves the chip short of waves)

__device__ __forceinline__ v4i dw_next_lane(const v4i& v) {   // lane i <- lane i + 1, each dword (= 4 channels of one pixel) on its own
    v4i r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = __builtin_amdgcn_update_dpp(v[k], v[k], 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
    return r;
}
}

// S: stride.  FQ: every int8 output format is a right shift into unsigned 8-bit behind a ReLU (shift 1 .. 16: 3-operation requantisation, f8_device.h; ReLU = the clamp).
// SUBS: output rows per MFMA pixel tile: 1 = 32 lanes along one row (28 outputs), 2 = two rows of 16 lanes (14 outputs each: 14-wide maps)
template <int S, int FQ, int SUBS>
__global__ void __launch_bounds__(256, S == 1 ? 4 : 3) dwconv3x3_mma_kernel(const DwArgs a) {
    constexpr int VW = SUBS == 2 ? 14 : DW_SW;                      // output columns per sub-row
    if constexpr (FQ == 1) set_fp_round_nearest_even();
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 3;
    const int cts = a.Cs >> 5, strips = (a.Q + VW - 1) / VW, bands = (a.P + a.band - 1) / a.band;
    const int sub = SUBS == 2 ? l31 >> 4 : 0, u = SUBS == 2 ? l31 & 15 : l31;      // sub-row of the tile, lane inside it
    // item = (image, band, strip, channel tile), channel tile fastest: the waves of a workgroup read the same pixels' other channels
    const long long item = (long long)blockIdx.x * 4 + wave;
    if (item >= (long long)a.N * bands * strips * cts) return;
    const int ct = (int)(item % cts);
    long long t = item / cts;
    const int strip = (int)(t % strips); t /= strips;
    const int band = (int)(t % bands);
    const int n = (int)(t / bands);
    const int q0 = strip * VW, p0 = band * a.band;
    const int p1 = (p0 + a.band) < a.P ? (p0 + a.band) : a.P;
    const int ch = ct * 32 + 16 * lh;                               // first of this lane's 16 channels (B operand / int8 output row)

    // ---- the nine diagonal weight fragments of this channel tile, the bias in accumulator order
    v4i wa[9];
    {
        const bool mine = (l31 >> 4) == lh;                         // K index == row index: rows 0-15 sit in K half 0, 16-31 in half 1
        const int dsel = (l31 & 15) >> 2, bsh = 8 * (l31 & 3);
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const unsigned wv = (unsigned)(unsigned char)a.w[(size_t)tp * a.Cs + ct * 32 + l31];
            const int piece = mine ? (int)(wv << bsh) : 0;
            wa[tp] = v4i{dsel == 0 ? piece : 0, dsel == 1 ? piece : 0, dsel == 2 ? piece : 0, dsel == 3 ? piece : 0};
        }
    }
    v4i bq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bq[g] = *(const v4i*)(a.bias + ct * 32 + 8 * g + 4 * lh);

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (unsigned)((size_t)a.N * a.H * a.W * a.Cs), 0x00020000);
    const int padv = a.in_signed ? 0 : (int)0x80808080u;
    // input column of this lane: stride 1: q0 - 1 + l (taps kx = 0, 1, 2 are this fragment shifted by 0, 1, 2 lanes);
    // stride 2: O = 2 (q0 + l) - 1 (kx = 0; kx = 2 is O of the next lane), E = 2 (q0 + l) (kx = 1)
    const int colA = S == 1 ? q0 - 1 + u : 2 * (q0 + u) - 1;
    const int colB = 2 * (q0 + u);
    const bool okA = colA >= 0 && colA < a.W, okB = colB < a.W;
    auto row_off = [&](int r, int col, bool ok) -> unsigned {
        return (ok && r >= 0 && r < a.H) ? (unsigned)((((size_t)n * a.H + r) * a.W + col) * a.Cs + ch) : kOOB;
    };
    auto fix = [&](v4i v, unsigned off) { if (off == kOOB) v = v4i{padv, padv, padv, padv}; return v; };

    // fragments of an input row: f[0..2] = taps kx = 0, 1, 2
    struct Row { v4i f[3]; };
    auto make_row = [&](const v4i& va, const v4i& vb) {
        Row R;
        if constexpr (S == 1) { R.f[0] = va; R.f[1] = dw_next_lane(va); R.f[2] = dw_next_lane(R.f[1]); }
        else { R.f[0] = va; R.f[1] = vb; R.f[2] = dw_next_lane(va); }
        return R;
    };
    auto load_row = [&](int r, v4i& va, v4i& vb, unsigned& oa, unsigned& ob) {
        oa = row_off(r, colA, okA);
        va = __builtin_amdgcn_raw_buffer_load_b128(rx, oa, 0, 0);
        if constexpr (S == 2) { ob = row_off(r, colB, okB); vb = __builtin_amdgcn_raw_buffer_load_b128(rx, ob, 0, 0); }
    };
    const int col_out = q0 + u;
    const bool col_ok = u < VW && col_out < a.Q;

    auto emit = [&](const v16i& acc, int p) {
        const bool lane_out = col_ok && p < p1;
        const size_t o = (((size_t)n * a.P + (p < p1 ? p : p0)) * a.Q + (col_ok ? col_out : 0)) * a.Cs + ch;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!a.q[k].ptr) continue;                              // wave-uniform
            unsigned d[4];
            if constexpr (FQ) {
                const float sc = FQ == 1 ? requant_u8_scale(a.q[k].n) : 0.0f;     // FQ == 2: integer requantisation (f8_device.h)
#pragma unroll
                for (int g = 0; g < 4; ++g) d[g] = requant_u8x4_sel<FQ == 2 ? 2 : 1>(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3], a.q[k].n, sc) ^ 0x80808080u;
            } else {
                const int floor0 = a.relu0 ? 0 : INT32_MIN;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    d[g] = pack4(requant1(max(acc[4 * g], floor0), a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(max(acc[4 * g + 1], floor0), a.q[k].n, a.q[k].lo, a.q[k].hi),
                                 requant1(max(acc[4 * g + 2], floor0), a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(max(acc[4 * g + 3], floor0), a.q[k].n, a.q[k].lo, a.q[k].hi)) ^ a.q[k].bias_xor;
            }
            auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
            if (lane_out) {
                const v4i ov = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
                *(v4i*)(a.q[k].ptr + o) = ov;
            }
        }
    };
    auto mac3 = [&](v16i acc, const Row& R, int ky) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wa[ky * 3 + kx], R.f[kx], acc, 0, 0, 0);
        return acc;
    };
    auto acc0 = [&]() {
        v16i acc;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * g + e] = bq[g][e];
        return acc;
    };

    // Output row p + sub reads input rows S (p + sub) - 1 + k, k = 0 .. 2 (fragment k).  A step advances SUBS output rows: fragment k of
    // the next step is fragment k + NEW of this one where that exists (the rows slide in registers), NEW fragments are loaded — one
    // step ahead, under this step's multiplies.
    constexpr int NEW = S == 1 ? SUBS : (SUBS == 1 ? 2 : 3), KEEP = 3 - NEW;
    auto in_row = [&](int p, int k) { return S * (p + sub) - 1 + k; };
    Row R[3];
    {
        v4i va[3], vb[3]; unsigned oa[3], ob[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < 3; ++k) load_row(in_row(p0, k), va[k], vb[k], oa[k], ob[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) R[k] = make_row(fix(va[k], oa[k]), S == 2 ? fix(vb[k], ob[k]) : vb[k]);
    }
    for (int p = p0; p < p1; p += SUBS) {
        v4i na[NEW], nb[NEW]; unsigned noa[NEW], nob[NEW];
        const bool more = p + SUBS < p1;
        if (more) {
#pragma unroll
            for (int j = 0; j < NEW; ++j) { nob[j] = 0; load_row(in_row(p + SUBS, KEEP + j), na[j], nb[j], noa[j], nob[j]); }
        }
        v16i acc = acc0();
#pragma unroll
        for (int k = 0; k < 3; ++k) acc = mac3(acc, R[k], k);
        emit(acc, p + sub);
        if (more) {
#pragma unroll
            for (int k = 0; k < KEEP; ++k) R[k] = R[k + NEW];
#pragma unroll
            for (int j = 0; j < NEW; ++j) R[KEEP + j] = make_row(fix(na[j], noa[j]), S == 2 ? fix(nb[j], nob[j]) : nb[j]);
        }
    }
}

// int8 outputs only, pad 1, stride 1 / 2, whole 32-channel tiles, strips that fill their lanes: output width >= 28 (measured: 14-wide maps,
// 14 of 32 lanes live, are 10 - 15 % SLOWER than the v_dot4 kernel: 17.7 -> 20.0 us; 112 / 56 / 28-wide ones 41 -> 26, 43 -> 27, 27 -> 19 us)
bool dwconv_mma_supported(const DwArgs& a) {
    return !a.out32 && a.w && a.bias4 && a.pad == 1 && (a.stride == 1 || a.stride == 2) && (a.Cs & 31) == 0 && (a.Q >= DW_SW || a.Q == 14) &&
           (a.stride == 1 ? (a.P == a.H && a.Q == a.W) : (a.H == 2 * a.P && a.W == 2 * a.Q)) &&
           (size_t)a.N * a.H * a.W * a.Cs < 0x7fffffffull;
}

hipError_t launch_dwconv_mma(const DwArgs& a0, hipStream_t s) {
    DwArgs a = a0; a.bias = a0.bias4;                               // bias + 128 * sum(w) for unsigned inputs
    int fq = a.relu0 != 0 ? ((a.acc_ok != 0 && !a.rq_int) ? 1 : 2) : 0;      // 1: float converter; 2: integer form of the same requantisation
    for (int k = 0; k < 2; ++k) {
        if (!a.q[k].ptr) continue;
        if (!(a.q[k].n > 0 && a.q[k].n <= 30 && a.q[k].lo == 0 && a.q[k].hi == 255 && a.q[k].bias_xor == 0x80808080u)) fq = 0;
        else if (fq == 1 && a.q[k].n > kRequantU8MaxShift) fq = 2;
    }
    const int subs = a.Q >= DW_SW ? 1 : 2, vw = subs == 2 ? 14 : DW_SW;
    a.band = DW_BAND;
    long long items = (long long)a.N * ((a.P + a.band - 1) / a.band) * ((a.Q + vw - 1) / vw) * (a.Cs >> 5);
    if (items < 4096 && a.P > 8) { a.band = 8; items = (long long)a.N * ((a.P + a.band - 1) / a.band) * ((a.Q + vw - 1) / vw) * (a.Cs >> 5); }   // < 4 waves per SIMD
    const unsigned grid = (unsigned)((items + 3) / 4);
#define F8_DWM(S_, FQ_, SB_) hipLaunchKernelGGL((dwconv3x3_mma_kernel<S_, FQ_, SB_>), dim3(grid), dim3(256), 0, s, a)
#define F8_DWQ(S_, SB_) do { if (fq == 1) F8_DWM(S_, 1, SB_); else if (fq == 2) F8_DWM(S_, 2, SB_); else F8_DWM(S_, 0, SB_); } while (0)
    if (a.stride == 1) { if (subs == 1) F8_DWQ(1, 1); else F8_DWQ(1, 2); }
    else               { if (subs == 1) F8_DWQ(2, 1); else F8_DWQ(2, 2); }
#undef F8_DWQ
#undef F8_DWM
    return hipGetLastError();
}

}  // namespace f8
