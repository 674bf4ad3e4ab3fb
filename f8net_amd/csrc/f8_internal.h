// f8_internal.h — kernel argument blocks and launchers shared by f8_kernels.hip and f8_net.cpp.
// Not part of the ABI (include/f8net.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace f8 {

// Per-handle tuning options (f8_net_set_option; include/f8net.h lists the keys).  Defaults are the measured best; the
// environment (F8_<KEY>) only seeds the defaults of a new handle, so two nets in one process can differ.
struct Options {
    int split = 2;               // concurrent sub-batches per run = arena copies (1..4)
    int fuse_blocks = 1;         // whole-bottleneck fusion
    int fuse_stages = -1;        // bit mask of stages for the identity-block fusion (-1: stages 0+1, stage 2 when a launch fills half the chip)
    int fuse_dual = 1;           // dual-GEMM downsample join
    int fuse_ds = 1;             // stage-opening block at unchanged resolution in one launch
    int fuse_opener = 1;         // stage-opening block with a stride-2 3x3 in one launch
    int fuse_fc = 1;             // the classifier writes the caller's logits buffer itself (no output launch)
    int fuse_stem = 1;           // stem conv + max-pool in one launch
    int fuse_input = 1;          // ... which also reads the raw network input (no input launch, no haloed NHWC4 copy)
    int wstat = 1;               // weight-stationary 1x1 kernel (f8_wstat.hip) where a launch gives every workgroup >= wstat_min_tiles pixel tiles
    int wstat_min_tiles = 2;
    int wstat_fast = 1;          // 0: always the general epilogue (requant in either direction, explicit ReLU floor)
    int s2wreg = 1;              // stride-2 3x3 convs of the stage-2 / stage-3 openers on conv3x3s2_wreg_kernel (f8_s2conv.hip)
    int wreg = 1;                // late 1x1 convs on conv1x1_wreg_kernel (weights straight to registers, f8_wreg.hip)
    int fuse_bchain = 2;         // consecutive BasicBlock identity blocks of a stage in ONE launch (f8_bchain.hip); 2: with the stage-opening block in front of them
    int fuse_chain = 1;          // all consecutive bottleneck blocks of a stage in ONE launch, int32 residual stream in registers (f8_chain.hip)
                                 // with the same number of rounds (224): groups that finish a round early free their CUs for the next batch's launches
    int fuse_chain7 = 1;         // ... and the identity blocks of a 7x7 bottleneck stage over clusters of eight workgroups (f8_cchain.hip); 0: fused_p12 + the residual-carrying 1x1
    int fuse_pool = 1;           // the network's last 1x1 conv (+ residual join) and the average pool behind it in one launch (f8_pool.hip)
    int fuse_tail = 1;           // ... and the JOIN of a stride-2 stage-opening block as the first block of its stage's chain (its body.0 + body.2 on f8_opener.hip, P12)
    int chain_timeout_ms = 10000; // bound of its halo-exchange spins (another process holding the CUs for longer: sticky error word, logits poisoned, f8_net_check)
    int fuse_p12 = 1;            // 7x7 identity blocks: body.0 + body.2 in one launch, split over a workgroup pair per image (f8_p12.hip)
    int fuse_head2 = 1;          // MobileNet-V2 head (3x3 / 2 conv, depthwise 3x3, 1x1) as one row-walking launch (f8_stem.hip, H2)
    int fuse_ir = 1;             // MobileNet-V2 inverted residual (expand -> depthwise -> project) in one launch: 1 = where it wins, 2 = always
    int patch3x3 = 1;            // LDS-patch 3x3 kernel
    int dual_wide = 2048;        // dual-GEMM joins with at least this many couts use the 128x128 tile
    int deep_nk = 7;             // K loops of at least this many steps use the deepest DMA ring
    int bk128 = 0;               // 128-byte K steps in conv_igemm_kernel
    int dw_dot4 = 1;             // v_dot4 depthwise kernel
    int dw_mma = 1;              // depthwise 3x3 on the matrix cores (f8_dwmma.hip) where it has an instance (output width >= 14, int8 outputs)
    int stem_rows = 1;           // ResNet head: the row-walking kernel (pool in registers) where it has an instance, else the tile kernel
    int stem_grid_div = 0;       // row-walking head on 1 / n of the CUs; 0 = all of them when it writes int32 (write-bound), half otherwise
    int stem_wpc = 3;            // resident stem workgroups per CU (2 / 3 / 4: 84.0 / 84.5 / 84.7 k img/s, same box)
    int opener_stg = 1;          // stride-2 opener: int8 output staged through LDS into 128-byte lines
    int chunk56 = -1, chunk28 = -1, chunk14 = -1;   // images per chunk of the fused blocks (-1: derived from chunk_budget_mb, 0: whole batch)
    int chunk_budget_mb = 96;    // a chunk's int32 stream must fit this much memory-side cache (3/8 of the 256 MiB Infinity Cache)
    int chunk_ds = 1, chunk_opener = 0;             // chunk the stage-opening blocks too (the stride-2 opener runs one workgroup per
                                                    // CU, 7 per image: measured 139 us in one 128-image launch, 161 us in 4 chunks)
    int whole_batch_launches = 0;   // planning hint: launches will cover the whole batch (f8_net_set_pipelined(2)), not max_batch / split images
    int pipeline_depth = 2;      // f8_net_set_pipelined(2): runs in flight (2..4, at most `split` arena copies)
    int shared_streams = 1;      // the internal streams are one set per device, shared by all handles (0: four streams of its own per handle)
    int arena_copies = 0;        // arena copies allocated at upload; 0 = `split`.  More than `split`: that many whole runs may be in flight
                                 // under pipelining mode 2 (pipeline_depth) while a non-pipelined run still cuts the batch `split` ways
    int split_streams = 1;       // 0: same launches serialised on the caller's stream (profiling)
    int graph = 0;               // hipGraph capture / replay of a run
    int stagger = -1, stagger_pipelined = 2;
    int check_device = 1;        // f8_net_run fails if the current device is not the one the handle was uploaded to
    int requant_float = 0;       // 0 (default since round 5: BASELINE north_star — no FP32 multiply in any epilogue): integer shift / round-half-even /
                                 // clamp in every kernel; 1: ReLU -> unsigned 8-bit right shifts of values the planner can bound may run through the
                                 // float converter (v_cvt_f32_i32, v_mul_f32 by 2^-n, v_cvt_pk_u8_f32: exact where planned, f8_device.h)
    int check_input_range = 1;   // int32 inputs that are NARROWED to the head's 8-bit format (no requant) are range-checked on the device; f8_net_check reports
};
void options_from_env(Options* o);                       // f8_net.cpp
int* option_slot(Options* o, const char* key);            // nullptr: unknown key

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
    int32_t deep_nk;                       // Options::deep_nk (ring depth rule of launch_conv_t)
    int32_t no_fast;                       // !Options::wstat_fast
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
    int32_t use_dot4;                      // Options::dw_dot4
    int32_t use_mma;                       // Options::dw_mma
    int32_t band;                          // f8_dwmma.hip: output rows per wave (set by its launcher)
    int32_t acc_ok;                        // every accumulator is provably below 2^31 - 2^16 in magnitude (planner: conv_acc_bounded): the
                                           // float requantisation (requant_u8x4, f8_device.h) equals the wrapping integer one
    int32_t rq_int;                        // Options::requant_float == 0: integer requantisation only
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
    const uint8_t* xu8; int32_t u8_nhwc;              // uint8 images (f8_net_run_u8): NCHW planes or NHWC pixels, through `lut`
    int16_t lut[3 * 256];                             // lut[c * 256 + byte] = the head-format integer of that pixel value
    uint32_t xor8;                         // 0x80808080 when the int8 consumer format is unsigned (biased storage)
    uint32_t* err; int32_t chk_lo, chk_hi; // err != nullptr: int32 inputs narrowed to 8 bits are checked against [chk_lo, chk_hi] (sticky error word)
    int8_t* out8;  int32_t Cs8;            // NHWC int8 (Cs8-channel rows), or
    int8_t* stem;  int32_t Hp, Wp, pad;    // zero-haloed NHWC4 for the stem conv
    int32_t* out32; int32_t Cs32;
};

struct OutArgs {                           // NHWC int32 -> NCHW int32 / float32
    const int32_t* x; int32_t N, C, HW, Cs;
    void* out; int32_t as_float;
    const uint32_t* err; uint32_t epoch;   // the run's stage-chain error word (or nullptr) and tag: when the word carries THIS run's tag, the outputs are poisoned (NaN / INT32_MIN)
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
    int32_t stride2;                       // stage-opening block with a stride-2 3x3 (f8_opener.hip): H, W are the INPUT map
    int32_t stg;                           // Options::opener_stg
    int32_t acc_ok, rq_int;                // P12: both convs' accumulators bounded (conv_acc_bounded) / Options::requant_float == 0 (see DwArgs)
    int32_t p12only;                       // f8_opener.hip: body.0 + body.2 only, q[0] = body.2's output (NHWC int8, MID channels); the join runs as the
                                           // first block of the stage's chain launch (ChainArgs::tail)
};

// One launch for ALL consecutive bottleneck blocks of a ResNet stage (f8_chain.hip): the int32 residual stream of a tile stays in
// registers from block to block.  Weight pointers are the MFMA-fragment-order images (pack_frag_weights).
struct ChainBlk {
    const int8_t* w0; const int8_t* w2; const int8_t* w4; const int8_t* wsc;     // wsc / bsc: stage-opening block only (1x1 shortcut conv)
    const int32_t* b0; const int32_t* b2; const int32_t* b4; const int32_t* bsc;  // offset-corrected, single class
    int32_t nq, loq, hiq; uint32_t xorq;   // requant block input (int32 stream) -> body.0's input format
    int32_t n1, lo1, hi1; uint32_t xor1;   // requant body.0 output -> body.2 input format
    int32_t n2, lo2, hi2; uint32_t xor2;   // requant body.2 output -> body.4 input format
    int32_t relu_a, relu_b, relu1;         // ReLU after body.0 / body.2 / the join
    int32_t acc_shl, res_shl;              // residual join: (conv << acc_shl) + (other << res_shl)
};
constexpr int kChainMaxBlocks = 6;
struct ChainArgs {
    ChainBlk blk[kChainMaxBlocks]; int32_t nblk;
    int32_t acc_ok;                        // body.0 / body.2 accumulators of every block bounded (see DwArgs::acc_ok)
    int32_t stream_ok;                     // ... and so is the int32 stream after every block (and the stream a chain of identity blocks reads): f8_net.cpp tensor_amax
    int32_t rq_int;                        // Options::requant_float == 0: integer requantisation everywhere (no float-converter instance)

    const int32_t* xr;                     // first block an identity block: the stage's int32 stream (I32T) — its int8 form is computed in the launch
    const int8_t* x8in;                    // first block a stage-opening block: its int8 NHWC input [N*H*W][CIN0] in body.0's / the shortcut's format
    // tail != 0: the first block is only the JOIN of a stage-opening block with a stride-2 3x3 (body.0 + body.2 ran in f8_opener.hip, P12):
    // x8in = the block input at TWICE the resolution [N][2H][2W][CIN0] in the SHORTCUT's int8 format (its pixels (2p, 2q) are the 1x1 / 2
    // shortcut's operand), m2in = body.2's output [N*H*W][MID] in body.4's int8 input format; blk[0] carries wsc / bsc / w4 / b4 / the join
    const int8_t* m2in; int32_t tail;
    int32_t N, NG;                         // images; image groups resident at once (grid = NG * tiles per image)
    int32_t* out32; QuantOut q[2];         // forms of the last block's output
    uint32_t* sync;                        // [0] ticket, [16 + workgroup] halo flag; zeroed before every launch
    uint32_t* err;                         // error word (a halo spin timed out): (epoch << 8) | code; read by f8_net_check
    uint32_t* err_host;                    // the same word's mirror in host-visible (pinned, mapped) memory: f8_net_run reads it WITHOUT a synchronisation and
                                           // refuses further runs until f8_net_check has collected the error (nullptr: no mirror)
    uint32_t epoch;                        // this run's tag (1 .. 2^24 - 1): only an error of THIS run ends waits early / poisons the logits —
                                           // a word left by an earlier run stays for f8_net_check to report and changes nothing else
    int8_t* xchg;                          // halo rows between vertically adjacent tiles: [workgroup][parity][side][W * MID]
    uint32_t timeout_ticks;                // bound of every spin (100 MHz wall clock)
    void* trace;
    int32_t R;                             // rows per tile of the instance to launch (chain_shape)
    int32_t pool;                          // f8_cchain.hip only: out32 / q[] are forms of the AVERAGE POOL behind the last block ([N][C]: FXQAvgPool2d's wrapping int32 sum over the map)
};
constexpr int kChainSyncWords = 16 + 512;   // [0] ticket, [1] workgroups out, [16 + workgroup] halo flags (up to two workgroups per CU)
constexpr int kChainErrWord = 1000;          // the sticky error word, inside the first 4096 bytes of the scratch
constexpr size_t kChainXchgBytes = (size_t)512 * 2 * 2 * 3584;

// One launch for consecutive BasicBlock identity blocks of a ResNet-18 / 34 stage (f8_bchain.hip); weights in fragment order.
struct BChainBlk {
    const int8_t* wa; const int8_t* wb;    // first / second 3x3
    const int32_t* ba; const int32_t* bb;  // offset-corrected, single class (the LDS patches carry a biased-zero border)
    int32_t nq, loq, hiq; uint32_t xorq;   // requant block input (int32 stream) -> the first conv's input format
    int32_t n1, lo1, hi1; uint32_t xor1;   // requant first conv output -> second conv's input format
    int32_t relu_a, relu1;                 // ReLU after the first conv / after the join
    int32_t acc_shl, res_shl;              // join: (conv << acc_shl) + (stream << res_shl)
};
constexpr int kBChainMaxBlocks = 6;
struct BChainArgs {
    BChainBlk blk[kBChainMaxBlocks]; int32_t nblk;
    int32_t acc_ok;                        // first-conv accumulators of every block bounded (see DwArgs::acc_ok)
    int32_t stream_ok;                     // ... and the int32 stream after every block / the stream an identity-block chain reads (see ChainArgs)
    int32_t rq_int;                        // Options::requant_float == 0 (see ChainArgs)

    const int32_t* xr;                     // the stage's int32 stream (I32T): chains of identity blocks
    // chains that start with the stage-opening block (blk[0]: wa = 3x3 / 2 over C/2 channels, wb = 3x3, the stream = its 1x1 / 2 shortcut):
    const int8_t* x8in;                    // int8 NHWC input [N][2H][2W][C/2] in the format blk[0].wa reads
    const int8_t* x8sc;                    // ... in the format the shortcut reads (may be the same buffer)
    const int8_t* wsc; const int32_t* bsc; // shortcut conv, fragment order / offset-corrected bias
    int32_t N, NG;
    int32_t* out32; QuantOut q[2];
    uint32_t* sync; uint32_t* err; uint32_t* err_host; uint32_t epoch; int8_t* xchg; uint32_t timeout_ticks;   // as ChainArgs
    void* trace;
};

// One launch for a MobileNet-V2 inverted-residual block: 1x1 expand -> depthwise 3x3 -> 1x1 project [+ int32 residual] (f8_ir.hip).
struct IRArgs {
    int32_t acc_ok;                        // expand / depthwise accumulators bounded (see DwArgs::acc_ok)
    int32_t rq_int;                        // Options::requant_float == 0: integer requantisation only
    const int8_t* x8;                      // block input, int8 NHWC [N*H*W][CIN_S] in the expand conv's input format
    const int32_t* xr;                     // block input, int32 I32T (residual operand) or nullptr
    const int8_t* w0; const int32_t* b0;   // expand  [E32][CIN_S], offset-corrected bias [E32]
    const int8_t* wd4; const int32_t* bd4; // depthwise: dot4 image [E32/4][36 B], bias (+128*sum(w) for unsigned inputs) [E32]
    const int8_t* w4; const int32_t* b4;   // project [COUT_S][E32], offset-corrected bias [COUT_S]
    int32_t N, H, W, Ho, Wo, stride, R, G, tiles_per_img, E32;
    int32_t n1, lo1, hi1; uint32_t xor1;   // requant expand output -> depthwise input format
    int32_t n2, lo2, hi2; uint32_t xor2;   // requant depthwise output -> project input format
    int32_t relu_a, relu_b, relu0;         // ReLU after expand / depthwise / project
    int32_t acc_shl, res_shl, relu1;       // residual join
    int32_t* out32; QuantOut q[2];
    int32_t xp, off_patch, off_mid2, off_w;   // LDS layout (filled by launch_fused_ir)
    // magic numbers (fast_div) for / W, / (H*W), / Wo, / (R*Wo): [magic, sh1, sh2]
    uint32_t mW, mHW, mWo, mRWo; int32_t s1W, s2W, s1HW, s2HW, s1Wo, s2Wo, s1RWo, s2RWo;
};

// ResNet head in one launch: 7x7/2 conv + ReLU + requant (unsigned 8-bit) + 3x3/2 max-pool (f8_stem.hip).
struct StemPoolArgs {
    int32_t acc_ok;                        // conv accumulators bounded (see DwArgs::acc_ok)
    int32_t rq_int;                        // Options::requant_float == 0: integer requantisation only
    const int8_t* x; uint32_t x_bytes;     // haloed NHWC4 input [N][Hp][Wp][4], halo = conv pad + org pixels
    const int8_t* w; uint32_t w_bytes;     // [64][7][32 B]
    const int32_t* bias;                   // [64], offset-corrected (single class: the halo is biased zero)
    int32_t N, Hp, Wp, org;
    int32_t Pc, Qc, P, Q;                  // conv output size, pooled size
    int32_t relu0;
    int32_t* out32;                        // pooled int32 (I32T, 64 channels) or nullptr
    QuantOut q[2];                         // pooled int8 NHWC (64 channels) in up to two formats
    int32_t wpc;                           // Options::stem_wpc
    int32_t rows;                          // Options::stem_rows: the row-walking kernel where it has an instance
    int32_t grid_div;                      // Options::stem_grid_div: that kernel runs on 1 / grid_div of the CUs (0 = by output form)
    // h2 != 0: the MobileNet-V2 head instead (3x3 / 2 conv 3 -> 32 ReLU, depthwise 3x3 ReLU, 1x1 32 -> <= 32): w / bias = the head conv
    // ([32][3][32 B], single class), wd / bd = depthwise ([9][32] tap-major, bias + 128 sum(w)), w1 / b1 = the 1x1 ([32][32]); na / nb =
    // right shifts head -> depthwise input / depthwise -> 1x1 input (both unsigned 8-bit behind a ReLU); Pc = P, Qc = Q = its map; q[] only
    int32_t h2; const int8_t* wd; const int32_t* bd; const int8_t* w1; const int32_t* b1; int32_t na, nb;
    // raw network input read by the stem launch itself (no input launch, no haloed NHWC4 copy): NCHW planes, raw_kind 0 = int32 (xi),
    // 1 = fp32 quantised on the fly (xf, scale, qlo, qhi), 2 = uint8 through `lut`; raw_kind < 0: the haloed form `x`
    int32_t raw_kind, rC, rH, rW;
    const int32_t* xi; const float* xf; const uint8_t* xu8;
    float scale; int32_t qlo, qhi;
    uint32_t xor8;                         // 0x80808080 when the stem's input format is unsigned (stored biased)
    uint32_t* err; int32_t chk_lo, chk_hi; // raw_kind 0: values outside [chk_lo, chk_hi] set the sticky error word (err != nullptr)
    int16_t lut[3 * 256];
};

// Dynamic LDS above 64 KB must be opted into per kernel AND per device (a process that drives several GPUs launches the same
// kernel on each of them).  `done`: one static bit mask per call site (bit = device ordinal); a benign race sets the attribute twice.
inline bool dyn_lds_opted_in(unsigned long long* done, int* dev_out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { *dev_out = -1; return false; }
    *dev_out = dev;
    return ((*done >> dev) & 1ull) != 0;
}

struct ConvTile { int bm, bn, bk; };

// Tile choice for a conv; returns false if no kernel instance fits (ck % bk).
bool pick_conv_tile(int M, int coutP, int ck, bool has_res, bool bk128, ConvTile* t);
int  conv_grid(const ConvTile& t, int M, int coutP);

hipError_t launch_conv(const ConvArgs& a, const ConvTile& t, hipStream_t s);
hipError_t launch_fused_bottleneck(const FusedArgs& a, hipStream_t s);
bool fused_bottleneck_supported(int C, int MID, int H, int W, int imgs_per_launch, int stage_mask, int* R);
bool fused_ds_supported(int C, int MID, int COUT, int H, int W, int* R);
// stage-opening block with a stride-2 3x3 (f8_opener.hip); H, W = input map
bool fused_opener_supported(int C, int MID, int COUT, int H, int W, int* R);
hipError_t launch_fused_opener(const FusedArgs& a, hipStream_t s);
// all consecutive bottleneck blocks of a stage in one launch, residual stream in registers (f8_chain.hip).  cin0 != C: the first
// block is the stage-opening block at unchanged resolution (1x1 shortcut conv from cin0 channels).
bool chain_supported(int C, int MID, int H, int W, int cin0);
bool chain_tail_supported(int C, int MID, int H, int W, int cin0);   // ... a stride-2 opening block's join as the first block (H, W = the stage's resolution)
int chain_max_blocks(int C, int MID, int H, int W, int cin0, bool tail);
// rows per tile (4, or 2: the two-workgroups-per-CU instance of tuning builds, -DF8_CH_R2_S0=1) and resident workgroups per CU of the instance that runs the shape
void chain_shape(int C, int MID, int H, int W, int cin0, bool tail, int* R, int* wg_per_cu);
hipError_t launch_chain(const ChainArgs& a, int C, int MID, int H, int W, int cin0, hipStream_t s, char* launched = nullptr, size_t cap = 0);   // launched: the symbol it started
int chain_kernel_name(char* buf, size_t cap, int C, int MID, int H, int W, int cin0, bool tail, int fast);   // the symbol launch_chain starts (f8_chain.hip)
// the identity blocks of a 7x7 bottleneck stage over clusters of eight workgroups (f8_cchain.hip): reached through chain_supported / launch_chain
bool cchain_supported(int C, int MID, int H, int W, int cin0, bool tail);   // tail: the join of the stride-2 opening block as the first block (cin0 = its input channels)
size_t cchain_xchg_bytes();                                      // exchange scratch of a launch (per arena copy)
int cchain_clusters(int N, int slots);                           // clusters (ChainArgs::NG) a launch over N images starts on `slots` compute units
int cchain_kernel_name(char* buf, size_t cap, int fast);
int chain_fast(const ChainArgs& a);                              // f8_chain.hip: 0 = generic instance, 1 = float-converter requantisation, 2 = integer
hipError_t launch_cchain(const ChainArgs& a, int fast, hipStream_t s);
// consecutive BasicBlock identity blocks of a stage in one launch (f8_bchain.hip)
bool bchain_supported(int C, int H, int W);
bool bchain_ds_supported(int C, int H, int W);
int bchain_tiles_per_img(int C, int H, int W);
hipError_t launch_bchain(const BChainArgs& a, int C, int H, int W, hipStream_t s, char* launched = nullptr, size_t cap = 0);
int bchain_kernel_name(char* buf, size_t cap, int C, int H, int W, bool ds, int fast);                         // the symbol launch_bchain starts (f8_bchain.hip)
// 1x1 -> 3x3 of a 7x7 bottleneck block in one launch (f8_p12.hip); FusedArgs: x8, w0 / b0, w2 / b2, requant 1, q[] = the int8 outputs
bool fused_p12_supported(int C, int MID, int H, int W);
hipError_t launch_fused_p12(const FusedArgs& a, hipStream_t s);
// 1x1 conv with the weights streamed into registers (f8_wreg.hip); ConvArgs::w = the fragment-order image of the weights
bool conv1x1_wreg_supported(int ck, int coutP);
hipError_t launch_conv1x1_wreg(const ConvArgs& a, hipStream_t s);
// classifier: integer linear + int32 -> float32 / int32 [N][classes] into the caller's buffer (f8_fc.hip); ConvArgs::w = fragment order
bool fc_dense_supported(int ck, int coutP);
hipError_t launch_fc_dense(const ConvArgs& a, void* out, int classes, int as_float, const uint32_t* err, uint32_t epoch, hipStream_t s);   // err / epoch: the run's chain error word and tag (logits poisoned when the word carries the tag) or nullptr
// 3x3 / stride 2 / pad 1 with the input patch in LDS and the weights streamed into registers (f8_s2conv.hip); ConvArgs::w = fragment order
bool conv3x3s2_wreg_supported(int ck, int HO, int WO, int coutP);
hipError_t launch_conv3x3s2_wreg(const ConvArgs& a, hipStream_t s);
// weight-stationary 1x1 conv / dual GEMM / residual join (f8_wstat.hip); ConvArgs::w (and w2) = fragment-order images
bool conv1x1_wstat_supported(int k0, int k1, int coutP, bool has_res);
int conv1x1_wstat_waves(int k0, int k1);
bool conv1x1_wstat_fast(const ConvArgs& a);
hipError_t launch_conv1x1_wstat(const ConvArgs& a, int num_cu, hipStream_t s);
// MobileNet-V2 inverted residual (f8_ir.hip): instance for the padded channel pair + a tile (R rows or G whole images) that fits LDS
bool fused_ir_config(int cinS, int coutS, int H, int W, int stride, int* R, int* G);
hipError_t launch_fused_ir(const IRArgs& a, int cinS, int coutS, hipStream_t s);
// 3x3 / stride 1 / pad 1 with the input patch resident in LDS (f8_conv3x3.hip); config = false: no instance
bool conv3x3_patch_config(int cin, int H, int W, int coutP, int* R, int* IMGS, int* BN);
hipError_t launch_conv3x3_patch(const ConvArgs& a, int cin, hipStream_t s);
bool head2_supported(int H, int W);
bool dwconv_mma_supported(const DwArgs& a);
hipError_t launch_dwconv_mma(const DwArgs& a, hipStream_t s);
bool stem_pool_supported(int cin, int cout, int k, int stride, int pad, int pool_k, int pool_s, int pool_p, int P, int Q, int rows, int H, int W);
hipError_t launch_stem_pool(const StemPoolArgs& a, hipStream_t s);
hipError_t launch_dwconv(const DwArgs& a, hipStream_t s);
hipError_t launch_maxpool(const PoolArgs& a, hipStream_t s);
hipError_t launch_avgpool(const AvgArgs& a, hipStream_t s);
// last 1x1 conv (+ residual join) + average pool in one launch (f8_pool.hip); ConvArgs::w = fragment order, out32 / q[] = the POOLED forms [N][coutP]
bool conv1x1_pool_supported(int ck, int coutP, int pq);
hipError_t launch_conv1x1_pool(const ConvArgs& a, hipStream_t s);
hipError_t launch_add(const AddArgs& a, hipStream_t s);
hipError_t launch_input(const InArgs& a, hipStream_t s);
hipError_t launch_output(const OutArgs& a, hipStream_t s);
hipError_t launch_quantize_input(const float* x, int32_t* y, size_t n, float scale, int lo, int hi, hipStream_t s);
struct TopkKs { int k[8]; };                // the k list travels by value in the kernel arguments
hipError_t launch_topk_correct(const float* logits, const int64_t* target, int N, int C, TopkKs ks, int nk, float* correct, hipStream_t s);
hipError_t launch_requant_i32(const int32_t* src, int32_t* dst, size_t n, int sh, int lo, int hi, hipStream_t s);
hipError_t launch_relu_i32(int32_t* x, size_t n, hipStream_t s);
hipError_t launch_add_align_i32(int32_t* res, const int32_t* x, size_t n, int res_shl, int x_shl, hipStream_t s);

}  // namespace f8
