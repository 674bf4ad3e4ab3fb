// f8_internal.h — kernel argument blocks and launchers shared by f8_kernels.hip and f8_net.cpp.
// Not part of the ABI (include/f8net.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace f8 {

// One requantised int8 output of an epilogue: int_op_only_fix_quant with n = src_fl - dst_fl.
struct QuantOut {
    int8_t* ptr;      // NHWC int8, row stride ld (bytes); nullptr = absent
    int32_t n;        // shift (> 0 right with round-half-even, <= 0 left)
    int32_t lo, hi;   // clamp bounds: [-127,127] or [0,255]
    uint32_t bias_xor; // 0x80808080 for unsigned formats (stored biased: x ^ 0x80), 0 for signed
};

// Implicit-GEMM int8 convolution / linear on v_mfma_i32_32x32x32_i8.
//   D[cout][pixel] = sum_k W[cout][k] * (X[pixel][k] ^ xor_mask)   (+ offset-corrected bias)
// X rows are gathered from an NHWC int8 tensor: K runs over taps (r,s) then CK bytes per tap.
struct ConvArgs {
    const int8_t* x;  uint32_t x_bytes;
    const int8_t* w;  uint32_t w_bytes;    // packed [coutP][ktot], K-contiguous
    const int32_t* bias;                   // [ncls][coutP] border-class bias table (see pack_conv_weights)
    const uint8_t* rowcls; const uint8_t* colcls;   // class of output row p / col q; ncc = #col classes
    int32_t ncc;                           // 0: single class (no padding, or signed input)
    int32_t M;                             // N*P*Q output pixels
    int32_t PQ, Q;
    uint32_t mPQ, mQ; int32_t s1PQ, s2PQ, s1Q, s2Q;   // magic numbers for m / PQ and rem / Q (fast_div)
    int32_t sN, sP, sQ;                    // input byte strides: per image, per output row, per output col
    int32_t origin;                        // byte offset of tap (0,0) at p=q=0 (negative with padding)
    int32_t H, W, stride, pad, kh, kw;
    int32_t CK;                            // bytes per tap (multiple of BK)
    int32_t tapH, tapW;                    // byte offset per tap row / col
    int32_t ktot;                          // kh*kw*CK
    int32_t coutP;                         // padded cout = row stride of NHWC outputs (elements)
    // epilogue
    int32_t relu0;                         // ReLU directly after the conv
    const int32_t* res;                    // int32 NHWC residual operand or nullptr
    int32_t acc_shl, res_shl;              // alignment shifts (one of them is 0)
    int32_t relu1;                         // ReLU after the residual add
    int32_t* out32;                        // NHWC int32 (stride coutP) or nullptr
    QuantOut q[2];
    // dual GEMM (x2 != nullptr): a second 1x1 / pad 0 conv over the same output pixels whose result (+ bias2) is
    // the residual operand of the join; K steps of (x, w) run first, then those of (x2, w2)
    const int8_t* x2; uint32_t x2_bytes;
    const int8_t* w2; uint32_t w2_bytes;   // [coutP][ktot2]
    const int32_t* bias2;                  // [coutP]
    int32_t sN2, sP2, sQ2, ktot2;          // input byte strides of x2 (per image / output row / output col), its K
    void* trace;                           // tuning builds (F8_TRACE) only; nullptr otherwise
};

// Depthwise 3x3 (groups == C), NHWC int8 in, VALU.
struct DwArgs {
    const int8_t* x; const int8_t* w;      // w: [9][Cs] tap-major
    const int32_t* bias;                   // [Cs]
    const int8_t* w4; const int32_t* bias4; // dot4 kernel: [Cs/4][9] tap-transposed dwords, bias + 128*sum(w) for unsigned inputs
    int32_t N, H, W, P, Q, Cs, stride, pad;
    int32_t in_signed;
    int32_t relu0;
    int32_t* out32;
    QuantOut q[2];
};

struct PoolArgs {                          // max-pool, NHWC
    const void* x; int32_t in_is_i8, in_signed;
    int32_t N, H, W, P, Q, Cs, k, stride, pad;
    int32_t* out32;
    QuantOut q[2];                         // when in_is_i8: q[0].ptr is the int8 output, no requant
};

struct AvgArgs {                           // FXQAvgPool2d sum over H*W, NHWC int32 in
    const int32_t* x; int32_t N, HW, Cs;
    int32_t* out32;                        // [N][Cs] or nullptr
    QuantOut q[2];
};

struct AddArgs {                           // standalone align-add (when it cannot be fused) / requant
    const int32_t* a; const int32_t* b; int32_t M, Cs;   // M pixels x Cs channels, int32 in I32T layout
    int32_t a_shl, b_shl, relu;
    int32_t* out32;
    QuantOut q[2];
};

struct InArgs {                            // network input: int32 NCHW -> NHWC forms
    const int32_t* x; int32_t N, C, H, W;
    const float* xf; float scale; int32_t qlo, qhi;   // fp32 images quantised on the fly (x unused): rint(xf * scale) clamped
    uint32_t xor8;                         // 0x80808080 when the int8 consumer format is unsigned (biased storage)
    int8_t* out8;  int32_t Cs8;            // NHWC int8 (Cs8-channel rows), or
    int8_t* stem;  int32_t Hp, Wp, pad;    // zero-haloed NHWC4 for the stem conv
    int32_t* out32; int32_t Cs32;
};

struct OutArgs {                           // NHWC int32 -> NCHW int32 / float32
    const int32_t* x; int32_t N, C, HW, Cs;
    void* out; int32_t as_float;
};

// One launch for a ResNet bottleneck identity block (f8_fused.hip).
struct FusedArgs {
    const int8_t* x8; uint32_t x_bytes;    // block input, int8 NHWC [N*H*W][C] in body.0's input format
    const int32_t* xr;                     // block input, int32 I32T (residual operand)
    const int8_t* w0; const int8_t* w2; const int8_t* w4; uint32_t w0_bytes, w2_bytes, w4_bytes;   // [MID][C], [MID][9][MID], [C][MID]
    const int32_t* b0; const int32_t* b2; const int32_t* b4;                                        // offset-corrected biases
    // DS variant (wsc != nullptr): the join's other operand is the shortcut conv Wsc[COUT][C] . x (+ bsc) instead of xr
    const int8_t* wsc; uint32_t wsc_bytes; const int32_t* bsc; int32_t COUT;
    int32_t N, H, W, C, MID, R, tiles_per_img;
    int32_t n1, lo1, hi1; uint32_t xor1;   // requant body.0 output -> body.2 input format
    int32_t n2, lo2, hi2; uint32_t xor2;   // requant body.2 output -> body.4 input format
    int32_t relu_a, relu_b;                // ReLU after body.0 / body.2
    int32_t acc_shl, res_shl, relu1;       // residual join
    int32_t* out32; QuantOut q[2];
    void* trace;                           // tuning builds (F8_TRACE) only
};

// ResNet head in one launch: 7x7/2 conv + ReLU + requant (unsigned 8-bit) + 3x3/2 max-pool (f8_stem.hip).
struct StemPoolArgs {
    const int8_t* x; uint32_t x_bytes;     // haloed NHWC4 input [N][Hp][Wp][4], halo = conv pad + org pixels
    const int8_t* w; uint32_t w_bytes;     // [64][7][32 B]
    const int32_t* bias;                   // [64], offset-corrected (single class: the halo is biased zero)
    int32_t N, Hp, Wp, org;
    int32_t Pc, Qc, P, Q;                  // conv output size, pooled size
    int32_t relu0;
    int32_t* out32;                        // pooled int32 (I32T, 64 channels) or nullptr
    QuantOut q[2];                         // pooled int8 NHWC (64 channels) in up to two formats
};

struct ConvTile { int bm, bn, bk; };

// Tile choice for a conv; returns false if no kernel instance fits (ck % bk).
bool pick_conv_tile(int M, int coutP, int ck, bool has_res, ConvTile* t);
int  conv_grid(const ConvTile& t, int M, int coutP);
int  conv_deep_nk();

hipError_t launch_conv(const ConvArgs& a, const ConvTile& t, hipStream_t s);
hipError_t launch_fused_bottleneck(const FusedArgs& a, hipStream_t s);
bool fused_bottleneck_supported(int C, int MID, int H, int W, int imgs_per_launch, int* R);
bool fused_ds_supported(int C, int MID, int COUT, int H, int W, int* R);
// 3x3 / stride 1 / pad 1 with the input patch resident in LDS (f8_conv3x3.hip); config = false: no instance
bool conv3x3_patch_config(int cin, int H, int W, int coutP, int* R, int* IMGS, int* BN);
hipError_t launch_conv3x3_patch(const ConvArgs& a, int cin, hipStream_t s);
bool stem_pool_supported(int cin, int cout, int k, int stride, int pad, int pool_k, int pool_s, int pool_p, int P, int Q);
hipError_t launch_stem_pool(const StemPoolArgs& a, hipStream_t s);
hipError_t launch_dwconv(const DwArgs& a, hipStream_t s);
hipError_t launch_maxpool(const PoolArgs& a, hipStream_t s);
hipError_t launch_avgpool(const AvgArgs& a, hipStream_t s);
hipError_t launch_add(const AddArgs& a, hipStream_t s);
hipError_t launch_input(const InArgs& a, hipStream_t s);
hipError_t launch_output(const OutArgs& a, hipStream_t s);
hipError_t launch_quantize_input(const float* x, int32_t* y, size_t n, float scale, int lo, int hi, hipStream_t s);
hipError_t launch_topk_correct(const float* logits, const int64_t* target, int N, int C, const int* ks_dev, int nk, float* correct, hipStream_t s);
hipError_t launch_requant_i32(const int32_t* src, int32_t* dst, size_t n, int sh, int lo, int hi, hipStream_t s);
hipError_t launch_relu_i32(int32_t* x, size_t n, hipStream_t s);
hipError_t launch_add_align_i32(int32_t* res, const int32_t* x, size_t n, int res_shl, int x_shl, hipStream_t s);

}  // namespace f8
