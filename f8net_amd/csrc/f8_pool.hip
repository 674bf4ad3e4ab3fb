// f8_pool.hip — the last 1x1 convolution of a network (with its residual join, where it has one) AND the average pool behind it in one launch (gfx950).
//
// IntModel.forward ends  ... -> last IntBlock / tail conv -> self.avgpool(x) -> classifier  (/root/reference/models/fix_resnet.py:363-383,
// fix_mobilenet_v2.py:231-241, fix_mobilenet_v1.py:139-147); FXQAvgPool2d's int branch (fix_quant_ops.py:126-134) is `x.sum(-1).sum(-1)` narrowed to
// int32, i.e. a wrapping int32 sum over the 7 x 7 map.  As two launches the conv wrote its int32 result (ResNet-50: 51 MB per 128 images) only for the
// pool to read it back.  Here a workgroup owns 256 output channels and walks a share of the images: an image's 49 int8 input rows sit in LDS, each wave keeps
// the weights of its 32 channels in registers (fragment order, loaded ONCE for all its images), two MFMA pixel tiles cover the map, and the epilogue — ReLU / residual join exactly as
// conv1x1_wstat_kernel's (align shifts, wrapping add, clamp, ReLU) — ends in a sum over the 49 live lanes (wrapping adds: exact in any order), of which
// only the pooled [image][channel] values leave the chip: int32 (I32T, one pixel per image) and / or requantised int8 for the classifier.
#include "f8_device.h"
#include <algorithm>

namespace f8 {

// K: input channels (bytes per pixel row); RES: int32 residual operand (I32T, the block input) joined before the pool
template <int K, bool RES>
__global__ void __launch_bounds__(512) conv1x1_pool_kernel(const ConvArgs a) {
    constexpr int NK = K / 32, XS = K + 16;                       // K32 steps; LDS row stride (padded: conflict-free b128 fragment reads)
    extern __shared__ __attribute__((aligned(16))) char lds[];    // 2 x [64 px][XS]
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
    const int ct = blockIdx.y * 8 + wave;                         // output channel tile of this wave
    const int PQ = a.PQ, NIMG = a.M / a.PQ;                       // pixels per image (<= 64), images

    // ---- this wave's weights (fragment order [tile][K32 step][lane][16 B]) and bias: ONCE; the workgroup then walks images blockIdx.x, + gridDim.x, ...
    //      (one image per workgroup re-streamed the 1 MB of weights per image: 34 us per 128 images for ResNet-50's last join, as slow as the two launches)
    const v4i* const wp = (const v4i*)a.w + (size_t)ct * NK * 64 + lane;
    v4i wf[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) wf[k] = wp[(size_t)k * 64];
    v4i bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = *(const v4i*)(a.bias + ct * 32 + 8 * g + 4 * lh);
    const int floor0 = a.relu0 ? 0 : INT32_MIN;
    const int floor1 = a.relu1 ? 0 : -2147483647;                 // the join's clamp_(min=-(2^31-1)) and the ReLU floor are one max

    // software pipeline over this workgroup's images: while image n is multiplied and pooled, image n + gridDim.x's rows (registers -> the other LDS
    // buffer) and residual values are in flight; one barrier per image
    constexpr int XL = (64 * (K / 16) + 511) / 512;               // 16-byte pieces of an image's rows per thread
    auto load_x = [&](int n, v4i (&xr)[XL]) {
#pragma unroll
        for (int q = 0; q < XL; ++q) {
            const int idx = tid + q * 512, row = idx / (K / 16), c16 = idx - row * (K / 16);
            xr[q] = v4i{0, 0, 0, 0};                              // rows beyond the map: zeros (their lanes are masked out of the sum anyway)
            if (n < NIMG && idx < 64 * (K / 16) && row < PQ) xr[q] = *(const v4i*)(a.x + ((size_t)n * PQ + row) * K + c16 * 16);
        }
    };
    auto store_x = [&](char* buf, const v4i (&xr)[XL]) {
#pragma unroll
        for (int q = 0; q < XL; ++q) { const int idx = tid + q * 512, row = idx / (K / 16), c16 = idx - row * (K / 16); if (idx < 64 * (K / 16)) *(v4i*)(buf + row * XS + c16 * 16) = xr[q]; }
    };
    auto load_res = [&](int n, v4i (&rs)[2][4]) {
        if constexpr (RES) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int px = j * 32 + l31, m = (n < NIMG ? n : 0) * PQ + (px < PQ ? px : 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) rs[j][g] = *(const v4i*)(a.res + i32t_index(m, ct * 32 + 8 * g + 4 * lh, a.coutP));
            }
        }
    };
    v4i xr[XL], rs[2][4], rsn[2][4];
    load_x(blockIdx.x, xr);
    load_res(blockIdx.x, rs);
    store_x(lds, xr);
    __syncthreads();
    int cur = 0;
    for (int n = blockIdx.x; n < NIMG; n += gridDim.x, cur ^= 1) {
        const char* const xb = lds + cur * (64 * XS);
        load_x(n + gridDim.x, xr);                                // the next image (zeros past the end)
        load_res(n + gridDim.x, rsn);
        v16i acc[2];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[0][4 * g + e] = bias[g][e]; acc[1][4 * g + e] = bias[g][e]; }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const v4i x0 = *(const v4i*)(xb + l31 * XS + k * 32 + lh * 16), x1 = *(const v4i*)(xb + (32 + l31) * XS + k * 32 + lh * 16);
            acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[k], x0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[k], x1, acc[1], 0, 0, 0);
        }
        // ---- epilogue: ReLU / join per value, then the pool: sum over the map's pixels = over the live lanes of both tiles
        unsigned tot[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            unsigned s = 0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int v = max(acc[j][r], floor0);
                if constexpr (RES) v = max((int)(((unsigned)v << a.acc_shl) + ((unsigned)rs[j][r >> 2][r & 3] << a.res_shl)), floor1);
                s += (j * 32 + l31 < PQ) ? (unsigned)v : 0u;
            }
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) s += (unsigned)__shfl_xor((int)s, m);     // inside each 32-lane half: the halves hold different channels
            tot[r] = s;
        }
        if (l31 == 0) {                                           // lane halves 0 / 1: channels ct * 32 + 8 g + 4 lh + e
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = ct * 32 + 8 * g + 4 * lh;
                if (a.out32) { const v4i o = {(int)tot[4 * g], (int)tot[4 * g + 1], (int)tot[4 * g + 2], (int)tot[4 * g + 3]}; *(v4i*)(a.out32 + i32t_index(n, c, a.coutP)) = o; }
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (a.q[k].ptr)
                        *(unsigned*)(a.q[k].ptr + (size_t)n * a.coutP + c) =
                            pack4(requant1((int)tot[4 * g], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1((int)tot[4 * g + 1], a.q[k].n, a.q[k].lo, a.q[k].hi),
                                  requant1((int)tot[4 * g + 2], a.q[k].n, a.q[k].lo, a.q[k].hi), requant1((int)tot[4 * g + 3], a.q[k].n, a.q[k].lo, a.q[k].hi)) ^ a.q[k].bias_xor;
            }
        }
        store_x(lds + (cur ^ 1) * (64 * XS), xr);                 // the other buffer: its last readers passed the previous barrier
        if constexpr (RES) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) rs[j][g] = rsn[j][g];
        }
        __syncthreads();
    }
}

// K = padded input channels, coutP = padded output channels, pq = pixels of the pooled map
// (K = 1024 — MobileNet-V1's last pointwise conv — keeps its two launches: 32 weight fragments + the prefetch registers do not fit 256 VGPRs)
bool conv1x1_pool_supported(int ck, int coutP, int pq) { return (ck == 320 || ck == 512) && coutP % 256 == 0 && pq >= 1 && pq <= 64; }

template <int K, bool RES>
static hipError_t launch_pool_t(const ConvArgs& a, hipStream_t s) {
    constexpr int lds = 2 * 64 * (K + 16);                        // two buffers of [64 px][K + 16]
    static unsigned long long attr_done = 0; int attr_dev = -1;
    if (lds > 64 * 1024 && !dyn_lds_opted_in(&attr_done, &attr_dev)) {
        hipError_t e = hipFuncSetAttribute((const void*)conv1x1_pool_kernel<K, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        if (attr_dev >= 0) attr_done |= 1ull << attr_dev;
    }
    // one workgroup per CU: channel groups x image walkers
    int dev = 0, cus = 256;
    static int cu_of[64] = {};
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        if (!cu_of[dev]) { int v = 0; cu_of[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256; }
        cus = cu_of[dev];
    }
    const int groups = a.coutP / 256, nimg = a.M / a.PQ;
    const int walkers = std::max(1, std::min(nimg, cus / groups));
    hipLaunchKernelGGL((conv1x1_pool_kernel<K, RES>), dim3(walkers, groups), dim3(512), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_conv1x1_pool(const ConvArgs& a, hipStream_t s) {
    if (!conv1x1_pool_supported(a.CK, a.coutP, a.PQ) || a.M % a.PQ) return hipErrorInvalidValue;
    const bool res = a.res != nullptr;
    if (a.CK == 320) return res ? launch_pool_t<320, true>(a, s) : launch_pool_t<320, false>(a, s);
    return res ? launch_pool_t<512, true>(a, s) : launch_pool_t<512, false>(a, s);
}

}  // namespace f8
