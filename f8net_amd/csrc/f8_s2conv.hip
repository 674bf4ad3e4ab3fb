// f8_s2conv.hip — the stride-2 3x3 convolutions of ResNet-50's stage-2 / stage-3 opening blocks (gfx950).
//
// body.2 of the first Bottleneck of stages 2 and 3 (`layer_(res)` of /root/reference/models/fix_resnet.py:34; 3x3 / stride 2 / pad 1,
// 256 -> 256 on 28x28 and 512 -> 512 on 14x14) ran on conv_igemm_kernel at 0.65 - 0.75 POP/s (40.6 / 45.6 us per 128 images): small
// tiles, a gathered A operand with one barrier per 64-byte K step.  The recipe of the 7x7 launch's second phase (f8_p12.hip, P2)
// as a stand-alone kernel:
//   * a workgroup owns 7 output rows of one image (all 7 of a 7x7 map; one half of a 14x14 map): the 15 x (W_in + 1) input entries
//     those rows touch go to LDS ONCE (LDS-direct DMA, out-of-image entries = zeros from the buffer range check, repaired by the
//     border-class bias as in conv_igemm_kernel) and are read-only afterwards; a tap is a constant entry offset (stride 2: the
//     lane's pixel (r, c) reads entry (2 r + tap row, 2 c + tap col));
//   * the weights stream from L2 straight into registers in MFMA-fragment order (host: pack_frag_weights), NBUF batches of 8 K32
//     steps rotating, no barrier in the K loop; every (workgroup, wave) walks the K32 steps of a tap starting at a different one
//     (integer sums are exact in any order) so that the L2 channels do not serialise on the step everybody wants;
//   * 8 waves = 8 output-channel tiles x all pixel tiles of the workgroup (2 x 32 for 49 pixels, 4 x 32 for 98): the 7x7 case splits
//     its 16 channel tiles over a workgroup PAIR (XCD-aware: one XCD streams one half of the weights), the 14x14 case its rows.
// Epilogue: ReLU, requantisation to the consumers' int8 formats (up to two), 16-byte stores.
#include "f8_device.h"

namespace f8 {

template <int CIN, int HO, int WO, int COUT>
struct S2Cfg {
    static constexpr int HI = 2 * HO, WI = 2 * WO;
    static constexpr int RO = 7;                                    // output rows per workgroup
    static constexpr int RSPLIT = HO / RO;                          // row groups per image
    static constexpr int CSPLIT = COUT / 32 / 8;                    // channel-tile groups (8 waves, one tile each)
    static constexpr int NPX = RO * WO, PT = (NPX + 31) / 32;       // output pixels / 32-pixel tiles per workgroup
    static constexpr int PR = 2 * RO + 1, PC = WI + 1, ENT = PR * PC;   // patch rows / columns (the first = the pad row / column) / entries
    static constexpr int PATCH_BYTES = (ENT * CIN + 8191) / 8192 * 8192;   // whole DMA rounds of the 512 threads
    static constexpr int NL = PATCH_BYTES / 16 / 512;               // DMA instructions per thread (whole 1 KB wave instructions)
    static constexpr int SPT = CIN / 32;                            // K32 steps per tap
    static constexpr int NB = 8, NBAT = 9 * SPT / NB;               // steps per weight batch, batches
    static constexpr int NBUF = 3;                                  // (4 for the two-tile instances spills once the B fragments are read a step ahead)
    static_assert(RSPLIT * CSPLIT == 2 && HO % RO == 0 && SPT % NB == 0 && (SPT & (SPT - 1)) == 0 && CIN >= 256, "a workgroup pair per image");
    static_assert(PATCH_BYTES <= 160 * 1024, "LDS");
};

template <int CIN, int HO, int WO, int COUT>
__global__ void __launch_bounds__(512) conv3x3s2_wreg_kernel(const ConvArgs a, const int N) {
    using Cfg = S2Cfg<CIN, HO, WO, COUT>;
    constexpr int HI = Cfg::HI, WI = Cfg::WI, RO = Cfg::RO, NPX = Cfg::NPX, PT = Cfg::PT, PC = Cfg::PC, ENT = Cfg::ENT;
    constexpr int SPT = Cfg::SPT, NB = Cfg::NB, NBAT = Cfg::NBAT, NBUF = Cfg::NBUF, NL = Cfg::NL, CPR = CIN / 16;
    constexpr int NK = 9 * SPT;
    extern __shared__ __attribute__((aligned(16))) char patch[];
    using SX = Swz<CIN>;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
    const int l31 = lane & 31, lh = lane >> 5;
    // workgroup -> (image, half), XCD-aware as in f8_p12.hip: XCDs 0-3 take half 0, XCDs 4-7 half 1
    const int xcd = blockIdx.x & 7, h = xcd >> 2;
    const int n = (blockIdx.x >> 3) * 4 + (xcd & 3);
    if (n >= N) return;
    const int rh = Cfg::RSPLIT == 2 ? h : 0, ch = Cfg::CSPLIT == 2 ? h : 0;
    const int ct = ch * 8 + wave;                        // this wave's output-channel tile
    const int r0 = rh * RO;                              // first output row of the workgroup
    // K-order rotation (integer sums are exact in any order): so that the workgroups do not all ask the L2 for the same kilobyte of the weight
    // stream at the same moment.  Round 6: by whole TAPS (`trot`, wave-uniform: scalar arithmetic once per tap) instead of by K32 steps inside a
    // tap — with a run-time step every B-fragment read paid five to six vector instructions of swizzled-address arithmetic (rocprof_r06_valu.md:
    // 13 - 21 x the essential vector work, 47 - 49 % issue stall); with the step a compile-time constant a read's address is ONE v_xor_b32 of a
    // per-tap base: entry * CIN | ((lh ^ entry % 16) << 4), whose low bits the step's (2 s << 4) never meets with a carry.
#ifndef F8_S2_TAPROT
#define F8_S2_TAPROT 1
#endif
    const int rot = F8_S2_TAPROT ? 0 : ((wave * 2 + (blockIdx.x >> 3) * 3 + (blockIdx.x & 7)) & (SPT - 1));
    const int trot = F8_S2_TAPROT ? __builtin_amdgcn_readfirstlane((int)((unsigned)(wave * 4 + (blockIdx.x >> 3) * 5 + (blockIdx.x & 7) * 2) % 9u)) : 0;
    auto tap_of = [&](int T) { const int t = T + trot; return t >= 9 ? t - 9 : t; };      // the tap multiplied T-th

    // ---- this lane's pixels: accumulators start at the (border-class) bias
    v16i acc[PT];
    int ent0[PT];                                        // patch entry of tap (0, 0) of the lane's pixel in tile j
    int gpx[PT];                                         // global output pixel, -1 beyond the workgroup's pixels
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const int px = j * 32 + l31, oc = px < NPX ? px : NPX - 1;
        const int orow = oc / WO, ocol = oc - orow * WO;
        ent0[j] = 2 * orow * PC + 2 * ocol;
        gpx[j] = px < NPX ? (n * HO + r0 + orow) * WO + ocol : -1;
        const int32_t* bp = a.bias + ct * 32 + 4 * lh;
        if (a.ncc > 0) bp += (size_t)((int)a.rowcls[r0 + orow] * a.ncc + (int)a.colcls[ocol]) * (size_t)a.coutP;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const v4i bv = *(const v4i*)(bp + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[j][4 * g + e] = bv[e];
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- the input patch: entry (pr, pc) = input pixel (2 r0 - 1 + pr, pc - 1), chunk c of entry e stored at chunk c ^ (e % 16)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int idx = tid + i * 512;
        const int e = idx / CPR, chunk = (idx % CPR) ^ SX::f(e);
        const int pr = e / PC, pc = e - pr * PC;
        const int row = 2 * r0 - 1 + pr, col = pc - 1;
        const bool ok = e < ENT && row >= 0 && row < HI && col >= 0;
        const unsigned off = ok ? (unsigned)(((n * HI + row) * WI + col) * CIN + chunk * 16) : kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(patch + i * 8192 + wave * 1024), 16, off, 0, 0, 0);
    }
    // the counted wait below relies on every weight load being YOUNGER than the DMAs
    __builtin_amdgcn_sched_barrier(0);

    const v4i* const wp = (const v4i*)a.w + (size_t)ct * NK * 64 + lane;      // fragment order: [tile][K32 step][lane][16 B]
    // TAPROT: BUFFER loads (resource = this wave's channel tile, the step in the scalar offset, the lane in the one vector offset) — as flat loads every
    // fragment cost a 64-bit vector add (138 v_lshl_add_u64 per wave in the K loop)
    const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc((void*)((const v4i*)a.w + (size_t)ct * NK * 64), 0, (unsigned)(NK * 1024), 0x00020000);
    const unsigned wl16 = (unsigned)lane * 16u;
    v4i wbuf[NBUF][NB];
    auto load_batch = [&](v4i (&dst)[NB], int s0) {      // a batch stays inside a tap
        const int tb = F8_S2_TAPROT ? tap_of(s0 / SPT) * SPT : (s0 & ~(SPT - 1));
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            if constexpr (F8_S2_TAPROT) dst[s] = __builtin_amdgcn_raw_buffer_load_b128(rwt, wl16, (tb + ((s0 + s) & (SPT - 1))) * 1024, 0);
            else dst[s] = wp[(size_t)(tb + ((s0 + s + rot) & (SPT - 1))) * 64];
        }
    };
    constexpr int PRE = NBUF - 1 < NBAT ? NBUF - 1 : NBAT;
    static_for<PRE>([&](auto bc) { constexpr int B = decltype(bc)::value; load_batch(wbuf[B], B * NB); });
    wait_vmcnt<PRE * NB>();                              // the patch landed (the weight batches in flight are newer)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // One flat, fully unrolled K loop (static step index: the tap and the batch are constants).  The B fragments of step G + 1 are read
    // from the patch BEFORE the multiplies of step G (two register sets): the compiler left a `ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma`
    // triple per multiply otherwise (found by scanning the ISA for it: 275 of this kernel's 288 multiplies waited for a read issued right
    // in front of them; no other kernel of the library had the pattern).
    unsigned eoff[PT], esw[PT];
    auto rdx = [&](auto gc, v4i (&xf)[PT]) {
        constexpr int G = decltype(gc)::value, T = G / SPT;
        if constexpr (G % SPT == 0) {                    // a new tap: its patch entries (the fragments of the previous tap are in registers)
            const int t = tap_of(T), dy = (t * 11) >> 5, dx = t - 3 * dy;          // scalar
#pragma unroll
            for (int j = 0; j < PT; ++j) {
                const int ent = ent0[j] + dy * PC + dx;
                eoff[j] = F8_S2_TAPROT ? (unsigned)(ent * CIN) | (((unsigned)lh ^ (unsigned)(ent & 15)) << 4) : (unsigned)(ent * CIN);
                esw[j] = (unsigned)(ent & 15);
            }
        }
        if constexpr (F8_S2_TAPROT) {
#pragma unroll
            for (int j = 0; j < PT; ++j) xf[j] = *(const v4i*)(patch + (eoff[j] ^ (unsigned)((G % SPT) * 32)));
        } else {
            const unsigned c2 = (unsigned)((((G % SPT) + rot) & (SPT - 1)) * 2 + lh);
#pragma unroll
            for (int j = 0; j < PT; ++j) xf[j] = *(const v4i*)(patch + eoff[j] + ((c2 ^ esw[j]) << 4));
        }
    };
    v4i xa[PT], xb[PT];
    rdx(std::integral_constant<int, 0>{}, xa);
    static_for<NK>([&](auto gc) {
        constexpr int G = decltype(gc)::value, B = G / NB, S = G % NB;
        if constexpr (S == 0 && B + NBUF - 1 < NBAT) load_batch(wbuf[(B + NBUF - 1) % NBUF], (B + NBUF - 1) * NB);
        v4i (&cur)[PT] = (G & 1) ? xb : xa;
        v4i (&nxt)[PT] = (G & 1) ? xa : xb;
        if constexpr (G + 1 < NK) rdx(std::integral_constant<int, G + 1>{}, nxt);
#pragma unroll
        for (int j = 0; j < PT; ++j) asm volatile("" : "+v"(cur[j]));
#pragma unroll
        for (int j = 0; j < PT; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wbuf[B % NBUF][S], cur[j], acc[j], 0, 0, 0);
    });

    const int floor0 = a.relu0 ? 0 : INT32_MIN;
#pragma unroll
    for (int j = 0; j < PT; ++j) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!a.q[k].ptr) continue;
            unsigned d[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                d[g] = pack4(requant1(max(acc[j][4 * g + 0], floor0), a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(max(acc[j][4 * g + 1], floor0), a.q[k].n, a.q[k].lo, a.q[k].hi),
                             requant1(max(acc[j][4 * g + 2], floor0), a.q[k].n, a.q[k].lo, a.q[k].hi), requant1(max(acc[j][4 * g + 3], floor0), a.q[k].n, a.q[k].lo, a.q[k].hi)) ^ a.q[k].bias_xor;
            auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
            if (gpx[j] >= 0) {
                const v4i o = {(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
                *(v4i*)(a.q[k].ptr + (size_t)gpx[j] * COUT + ct * 32 + 16 * lh) = o;
            }
        }
    }
}

// 3x3 / stride 2 / pad 1, int8 outputs only: the instances (input bytes per pixel, output map, output channels)
bool conv3x3s2_wreg_supported(int ck, int HO, int WO, int coutP) {
    return (ck == 512 && HO == 7 && WO == 7 && coutP == 512) || (ck == 256 && HO == 14 && WO == 14 && coutP == 256) ||
           (ck == 256 && HO == 7 && WO == 7 && coutP == 512);      // ResNet-18 stage_3_layer_0.body.0
}

template <int CIN, int HO, int WO, int COUT>
static hipError_t launch_s2_t(const ConvArgs& a, int N, hipStream_t s) {
    using Cfg = S2Cfg<CIN, HO, WO, COUT>;
    static unsigned long long attr_done = 0; int attr_dev = -1;
    if (!dyn_lds_opted_in(&attr_done, &attr_dev)) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3s2_wreg_kernel<CIN, HO, WO, COUT>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::PATCH_BYTES);
        if (e != hipSuccess) return e;
        if (attr_dev >= 0) attr_done |= 1ull << attr_dev;
    }
    hipLaunchKernelGGL((conv3x3s2_wreg_kernel<CIN, HO, WO, COUT>), dim3((N + 3) / 4 * 8), dim3(512), Cfg::PATCH_BYTES, s, a, N);
    return hipGetLastError();
}

hipError_t launch_conv3x3s2_wreg(const ConvArgs& a, hipStream_t s) {
    const int HO = a.PQ / a.Q, N = a.M / a.PQ;
    if (a.out32 || a.res || a.x2) return hipErrorInvalidValue;
    if (a.CK == 512 && HO == 7 && a.Q == 7 && a.coutP == 512) return launch_s2_t<512, 7, 7, 512>(a, N, s);
    if (a.CK == 256 && HO == 14 && a.Q == 14 && a.coutP == 256) return launch_s2_t<256, 14, 14, 256>(a, N, s);
    if (a.CK == 256 && HO == 7 && a.Q == 7 && a.coutP == 512) return launch_s2_t<256, 7, 7, 512>(a, N, s);
    return hipErrorInvalidValue;
}

}  // namespace f8
