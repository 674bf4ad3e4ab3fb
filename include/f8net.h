/*
 * f8net.h — C ABI of libf8net.so: the MI355X (gfx950) fixed-point-8 integer inference path.
 *
 * The reference (snap-research/F8Net) has no FFI: its `int_op_only` forward is plain nn.Module
 * composition over int32 CPU tensors.  This header declares what a binding for
 * that path binds instead; every entry point cites the reference call site it replaces
 * (paths relative to /root/reference).  INTEGRATION.md shows the reference-side stubs.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.  Status: 0 = ok, < 0 = f8_status.
 *     No exception crosses the ABI; f8_last_error() gives the message for the calling thread.
 *   - All `*_dev` pointers are device (HBM) pointers owned by the caller.  `stream` is a
 *     hipStream_t passed as void*; every call is asynchronous on it and never synchronises.
 *   - Tensors at the boundary use the reference's format: int32, NCHW contiguous, values of
 *     activations in 8-bit range where the reference guarantees it.  The fraction length the
 *     reference carries as the Python attribute `output_fraclen` is an explicit integer here.
 *   - Inside a net, activations live as NHWC int8 / int32 in an arena owned by the handle.
 *   - Thread safety: distinct handles are independent; one handle must not be run concurrently.
 */
#ifndef F8NET_H
#define F8NET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define F8NET_VERSION 200

typedef enum f8_status {
    F8_OK = 0,
    F8_ERR_INVALID = -1,      /* argument the reference would assert on, or malformed graph */
    F8_ERR_UNSUPPORTED = -2,  /* legal in the reference, not built here (e.g. groups not in {1, Cin}) */
    F8_ERR_HIP = -3,          /* HIP runtime error (message in f8_last_error) */
    F8_ERR_NOMEM = -4,
    F8_ERR_STATE = -5         /* call out of order (e.g. run before finalize) */
} f8_status;

const char* f8_status_string(int status);
const char* f8_last_error(void);
int f8_version(void);
/* Number of visible HIP devices (0 without a GPU; never fails). */
int f8_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Op-level seam: element-wise pieces of IntBlock.forward on int32 device tensors (any shape,
 * n elements, contiguous).
 * ------------------------------------------------------------------------------------------ */

/* int_op_only_fix_quant(input, 8, dst_fl, src_fl, signed)  — models/fix_quant_ops.py:90-114.
 * Shift by n = src_fl - dst_fl with round-half-to-even (n > 0) or left shift (n <= 0), then
 * clamp to [-127,127] (signed) or [0,255].  F8_ERR_INVALID where the reference asserts
 * (dst_fl outside [0, 8 - signed], :91-98) or |n| > 30.  src == dst allowed. */
int f8_requant_i32(const int32_t* src_dev, int32_t* dst_dev, size_t n,
                   int src_fl, int dst_fl, int is_signed, void* stream);

/* Input quantisation of forward_loss as a stand-alone op — fix_train.py:683-692:
 *   normalize == 0: dst = round_half_even(255 * src)  (fl, is_signed ignored; output fraclen 8; no clamp, as in the reference)
 *   normalize != 0: dst = clamp(round_half_even(src * 2^fl), [-127,127] if is_signed else [0,255])
 * F8_ERR_INVALID where fix_quant asserts (fl outside [0, 8 - signed], fix_quant_ops.py:66-71). */
int f8_quantize_input_f32(const float* src_dev, int32_t* dst_dev, size_t n, int normalize, int fl, int is_signed, void* stream);

/* Scoring of forward_loss — fix_train.py:697-704: correct_dev[k][n] = 1.0f if target n is among the ks[k] largest
 * logits of image n, else 0 (the rows the reference concatenates behind the loss).  Equal logits rank by lower class
 * index.  logits_dev: float32 [N, classes]; target_dev: int64 [N]; ks: host array of nk <= 8 values. */
int f8_topk_correct_f32(const float* logits_dev, const int64_t* target_dev, int N, int classes,
                        const int* ks, int nk, float* correct_dev, void* stream);

/* nn.ReLU on int32 — models/fix_resnet.py:39,77.  In place. */
int f8_relu_i32(int32_t* x_dev, size_t n, void* stream);

/* Residual align-add-clamp — models/fix_resnet.py:40-54,63-76; fix_mobilenet_v2.py:34-48:
 * the operand with the smaller fraclen is shifted left, res += x (wrapping), clamp to
 * [-(2^31-1), 2^31-1].  res updated in place; *out_fl (host, may be NULL) = max(res_fl, x_fl). */
int f8_add_align_i32(int32_t* res_dev, const int32_t* x_dev, size_t n,
                     int res_fl, int x_fl, int* out_fl, void* stream);

/* ------------------------------------------------------------------------------------------
 * Net-level seam: IntModel.forward — models/fix_resnet.py:352-383,
 * fix_mobilenet_v2.py:207-241, fix_mobilenet_v1.py:120-147 — as a graph of integer ops built
 * once from the exported parameters (state_dict of `Model.int_model()`, fix_resnet.py:526-544)
 * and run many times.  The same builder with one conv / pool / linear node is the op-level
 * replacement for `layer_(res)` (fix_resnet.py:34,59), `self.head[...]` (:356-359),
 * FXQAvgPool2d (fix_quant_ops.py:126-134) and `self.classifier(x)` (:383).
 *
 * Tensor ids are small non-negative ints returned by the builder calls (< 0 = f8_status).
 * A tensor id names the int32-valued tensor the reference would hold at that point, together
 * with its output_fraclen; how it is materialised (int8 for conv consumers — the consumer's
 * int_op_only_fix_quant is fused into the producer's epilogue — int32 for residual / pooling
 * consumers) is decided by f8_net_finalize.
 * ------------------------------------------------------------------------------------------ */

typedef struct f8_net f8_net;

typedef struct f8_conv_desc {
    int32_t cin, cout;
    int32_t kernel;        /* square kernels: 1, 3, 7 in the reference nets */
    int32_t stride, pad;
    int32_t groups;        /* 1 or cin (depthwise, cout == cin) — fix_quant_ops.py:373-390 */
    int32_t weight_fl;     /* buffer `weight_fraclen` (fix_quant_ops.py:710) */
    int32_t input_fl;      /* buffer `input_fraclen`  (fix_quant_ops.py:711) */
    int32_t input_signed;  /* attr `input_symmetric`  (fix_quant_ops.py:709) */
    int32_t quant_input;   /* 1: int_op_only_fix_quant(src -> input_fl) precedes the conv
                              (fix_resnet.py:30-34); 0: src already holds input_fl-format
                              integers (head conv, fix_resnet.py:356-358; op-level use) */
    int32_t relu;          /* an nn.ReLU follows the conv (int_block, fix_resnet.py:314) */
} f8_conv_desc;

typedef struct f8_linear_desc {
    int32_t in_features, out_features;
    int32_t weight_fl, input_fl, input_signed, quant_input;
} f8_linear_desc;

f8_net* f8_net_create(void);
void f8_net_destroy(f8_net* net);

/* Network input: int32 NCHW [N,C,H,W] at run time, tagged with `fraclen` as forward_loss does
 * (fix_train.py:683-692: u8 0..255 at fraclen 8, or signed at head.input_fraclen). */
int f8_net_input(f8_net* net, int C, int H, int W, int fraclen);

/* Integer conv built by ReLUClipFXQConvBN.int_conv (fix_quant_ops.py:680-714).
 * weight: host int32 [cout, cin/groups, k, k] (values must fit int8); bias: host int32 [cout]
 * (32-bit fixed point at fraclen input_fl + weight_fl, :612-614) or NULL.
 * Result tensor: fraclen weight_fl + input_fl (fix_resnet.py:35-37). */
int f8_net_conv(f8_net* net, int src, const f8_conv_desc* desc,
                const int32_t* weight_host, const int32_t* bias_host);

/* Residual join of IntBlock.forward (fix_resnet.py:40-54 / :63-76) [+ post_relu :77]. */
int f8_net_add(f8_net* net, int a, int b, int relu);

/* Head max-pool nn.MaxPool2d(k, stride, pad) via the float detour (fix_resnet.py:358-359) or
 * FXQMaxPool2d (fix_quant_ops.py:150-157); exact integer max. */
int f8_net_maxpool(f8_net* net, int src, int kernel, int stride, int pad);

/* FXQAvgPool2d int branch (fix_quant_ops.py:126-134): sum over H,W in int64, truncate to int32,
 * fraclen += shift (shiftnum = round(log2(kernel_size^2)) = 6 for the nets' FXQAvgPool2d(7)). */
int f8_net_avgpool_sum(f8_net* net, int src, int shift);

/* Integer nn.Linear built by ReLUClipFXQLinear.int_fc (fix_quant_ops.py:1165-1195);
 * src must be [C,1,1]. weight: host int32 [out, in]; bias host int32 [out] or NULL. */
int f8_net_linear(f8_net* net, int src, const f8_linear_desc* desc,
                  const int32_t* weight_host, const int32_t* bias_host);

/* Marks `src` as the (single) network output.  as_float != 0: float32 (the `.float()` of
 * fix_resnet.py:383), else int32.  Layout NCHW [N,C,H,W] ([N,C] for pooled / linear results). */
int f8_net_output(f8_net* net, int src, int as_float);

/* Plans the graph for batches up to max_batch: fuses requant / ReLU / residual into producer
 * epilogues, picks kernels and tiles, packs weights (host side), lays out the arena.
 * Touches no device; works without a GPU. */
int f8_net_finalize(f8_net* net, int max_batch);

/* Human-readable plan (one line per kernel launch); returns bytes needed incl. NUL. */
size_t f8_net_describe(const f8_net* net, char* buf, size_t cap);
int f8_net_num_launches(const f8_net* net);
size_t f8_net_arena_bytes(const f8_net* net);
size_t f8_net_weight_bytes(const f8_net* net);
int f8_net_output_fraclen(const f8_net* net);
/* Output element count per image (C*H*W of the output tensor). */
size_t f8_net_output_elems(const f8_net* net);

/* Allocates device memory on the current device and uploads packed weights.  Implicit in the
 * first f8_net_run; explicit so that callers can keep allocation out of timed regions.  The handle is bound to that
 * device: a run with another device current fails with F8_ERR_STATE (option check_device). */
int f8_net_upload(f8_net* net);

/* Runs the net on `N` images (1 <= N <= max_batch).  input_dev: int32 NCHW [N,C,H,W];
 * output_dev: float32 or int32 [N, output_elems].  Asynchronous on `stream`.
 * F8_ERR_HIP without issuing anything when a stage-chain launch of an EARLIER run of this handle gave up a halo wait (see f8_net_check): the
 * failing launch also writes its code to a host-visible word that this call reads without synchronising; runs are refused until f8_net_check
 * has collected the error (round 5).  The logits of the failed run itself are poisoned (NaN / INT32_MIN); runs issued before the failure
 * became visible are unaffected — the error word carries the failed run's tag and only that run's waits / outputs react to it. */
int f8_net_run(f8_net* net, const int32_t* input_dev, void* output_dev, int N, void* stream);

/* Same net, fed with the fp32 images forward_loss receives (fix_train.py:676-692): the input quantisation
 *   normalize == 0:  x_int = round_half_even(255 * x), fraclen 8        (fix_train.py:689-692; requires an unsigned head at fraclen 8)
 *   normalize != 0:  x_int = clamp(round_half_even(x * 2^fl), +-127 or [0,255]), fl = the head's input fraclen
 *                    (fix_train.py:683-687 through fix_quant, models/fix_quant_ops.py:64-87)
 * is applied inside the input kernel, so the int32 NCHW tensor of the reference never exists in HBM.
 * images_dev: float32 NCHW [N,C,H,W].  F8_ERR_INVALID if normalize == 0 and the net's input fraclen is not 8. */
int f8_net_run_f32(f8_net* net, const float* images_dev, int normalize, void* output_dev, int N, void* stream);

/* Same net, fed with the uint8 pixels a decoder produces (1 byte per pixel and channel instead of 4): the reference's
 * `transforms.ToTensor()` [+ `transforms.Normalize(mean, std)`] (fix_train.py:299-329) and the input quantisation of
 * forward_loss (fix_train.py:683-692) are folded into a 3 x 256 table (built per call on the host in the float32 operations
 * torch executes, so the integers are the reference's bit for bit) that the input kernel looks up while it lays the image out
 * for the head conv.  images_dev: uint8, NCHW [N,3,H,W] (nhwc == 0) or NHWC [N,H,W,3] (nhwc != 0).
 * normalize == 0: x_int = the pixel value, fraclen 8 (needs an unsigned head at fraclen 8); mean / std ignored (may be NULL).
 * normalize != 0: x_int = clamp(round_half_even(((k / 255 - mean[c]) / std[c]) * 2^fl)), fl = the head's input fraclen;
 *                 mean, std: host float[3]. */
int f8_net_run_u8(f8_net* net, const uint8_t* images_dev, int nhwc, int normalize, const float* mean, const float* std,
                  void* output_dev, int N, void* stream);

/* Optional measured tile selection: times every implicit-GEMM convolution launch of a sub-batch of an N-image run on
 * the current device with each tile shape that has a kernel instance (HIP events on `stream`) and keeps the fastest; a
 * tile only replaces the planner's choice for a > 3 % win.  Outputs are bit-identical for every tile.  Blocks until
 * done (tens of milliseconds); returns the number of launches whose tile changed, or a negative status. */
int f8_net_autotune(f8_net* net, int N, void* stream);

/* Pipelined submission (off by default).  By default a run starts after everything enqueued on `stream` before it,
 * which includes the PREVIOUS run's join: consecutive runs execute back to back.  With pipelining on, the caller
 * promises one call of slack on its buffers — the input of run i was complete, and the output buffer of run i free,
 * by the time run i-1 was submitted (static or double-buffered buffers) — and the sub-batches of run i then start as
 * soon as the sub-batches of run i-1 that use the same arena have finished: the tail of one run overlaps the head of
 * the next (its memory-bound early stages with the previous run's compute-bound late stages).  Results still become
 * visible in `stream` order (every run joins `stream`).  Same `stream` for consecutive pipelined runs.
 * on == 2: same contract, different schedule — every run executes UNSPLIT on one of two internal streams / arena copies,
 * alternating, so that two consecutive runs are in flight together: each launch covers the whole batch (twice the
 * workgroups of a sub-batch launch), which is what the latency-bound launches of the late stages need at 128 images;
 * the latency of one run roughly doubles, the throughput of a stream of runs rises. */
int f8_net_set_pipelined(f8_net* net, int on);

/* Orders the NEXT run (any f8_net_run* entry) behind `event` (a hipEvent_t passed as void*; NULL clears): the run's first
 * launch waits for it in addition to its stream dependencies.  This is how a pipelined caller hands over an input that is
 * produced on another stream (an H2D copy, a decoder): record the event behind the producer, call this, then run.  Needed
 * because under f8_net_set_pipelined the run does NOT wait for work queued on `stream` after the previous run's entry.
 * One-shot: consumed by the next run. */
int f8_net_set_input_ready(f8_net* net, void* event);

/* Blocks until the device is idle and reports failures that happened INSIDE kernels of earlier runs of this handle: the
 * stage-chain launches (option fuse_chain) exchange halo rows between workgroups and bound every wait (chain_timeout_ms); a
 * workgroup whose neighbour never arrives stores (run tag << 8 | code) in an error word, the launch runs on without waiting, and THAT run's
 * outputs are invalid — they are POISONED where they leave the library (the classifier / output kernel writes NaN, or INT32_MIN for int32
 * outputs, when the word carries its run's tag), so a caller that never calls this function cannot mistake them for results; later runs are
 * refused by f8_net_run until this function has been called.  F8_OK, or F8_ERR_HIP with the code in f8_last_error — the words of EVERY
 * arena copy are then cleared and the launches' ticket / flag words re-armed from the host (the device is idle).  It also reports (F8_ERR_INVALID) an int32 network input that held
 * values outside the head's 8-bit format in a run since the last check: f8_net_run NARROWS such an input to 8 bits where the reference
 * would feed the full int32 to the head conv (fix_train.py:689 only asserts >= 0), so out-of-format values cannot be honoured; the range
 * is checked inside the input / stem kernel (option check_input_range, default 1).  Never needed for correctness of a healthy run. */
int f8_net_check(f8_net* net);

/* Per-handle tuning options.  A new handle takes its defaults from the environment (F8_<KEY IN CAPITALS>; F8_CHUNK for
 * chunk56) and otherwise the measured best; two handles in one process may differ.  Keys that decide the plan must be set
 * before f8_net_finalize (F8_ERR_STATE afterwards); scheduling keys may change between runs.
 *   planning  : split (1..4 concurrent sub-batches of a run), arena_copies (0 = split; more: that many whole runs in flight under
 *               f8_net_set_pipelined(2) with pipeline_depth), fuse_blocks, fuse_stages (bit mask, -1 = auto), fuse_dual,
 *               fuse_ds, fuse_opener, fuse_fc (the classifier writes the caller's logits buffer itself), fuse_stem, fuse_input (the fused stem launch reads the caller's NCHW buffer itself), fuse_ir (1 = where it wins, 2 = every block), fuse_chain (all consecutive bottleneck blocks of a stage in one launch, the int32 residual stream in registers; it takes precedence over fuse_stages and over the chunk56 / chunk28 / chunk14 keys for the stages it plans: set fuse_chain = 0 to get the per-block launches those keys govern), fuse_tail (the join of a stride-2 stage-opening block opens that launch: its int32 output never exists), fuse_chain7 (the 7x7 bottleneck stage — that join, its identity blocks and the average pool behind them — as one launch over clusters of eight workgroups, f8_cchain.hip; 0: the dual-GEMM / fused_p12 / residual-join launches of rounds 3 - 5), fuse_pool (the network's last 1x1 conv, with its residual join, and the average pool behind it in one launch),
 *               fuse_bchain (the same for BasicBlock stages: 1 = consecutive identity blocks, 2 = with the stage-opening block in front), stem_rows (ResNet head:
 *               row-walking kernel, pool in registers), fuse_head2 (MobileNet-V2: head conv + depthwise + 1x1 as one row-walking launch), dw_mma (depthwise
 *               3x3 on the matrix cores), shared_streams (internal streams are one set per device for all handles), fuse_p12 (7x7 block: first two
 *               convs in one launch), wstat (weight-stationary 1x1 kernel: plain, dual-GEMM and residual-join instances) with
 *               wstat_min_tiles (pixel tiles per workgroup a launch must offer; 0 = always) and wstat_fast (0 = general epilogue), wreg (weights-streamed 1x1 kernel for the
 *               512 -> 256 / 1024 -> 512 reductions of smaller launches),
 *               patch3x3, dual_wide, deep_nk, bk128, dw_dot4, opener_stg, whole_batch_launches (hint: runs will use
 *               f8_net_set_pipelined(2)),
 *               requant_float (default 0: INTEGER shift / round-half-even / clamp in every kernel — fix_quant_ops.py:99-112 literally, on gfx950's
 *               v_ashr_pk_u8_i32; no float instruction in any epilogue; 1: a ReLU -> unsigned-8-bit right shift (1..16) of a value the PLANNER
 *               CAN BOUND — conv accumulators, the int32 stream of a chain launch — runs through the float converter: v_cvt_f32_i32, v_mul_f32
 *               by 2^-n, v_cvt_pk_u8_f32: exact, compared with the reference's arithmetic over all 2^32 inputs on the device; anything
 *               unbounded takes the integer form by itself.  Same results either way, bit for bit)
 *   scheduling: chunk56 / chunk28 / chunk14 (images per chunk of the fused blocks; -1 = derived from chunk_budget_mb, 0 = whole
 *               batch), chunk_budget_mb (memory-side cache a chunk's int32 stream may occupy), chunk_ds, chunk_opener,
 *               split_streams, graph, stagger, stagger_pipelined, stem_wpc, stem_grid_div (row-walking head on 1 / n of the CUs; 0 = by output form), check_device, check_input_range, pipeline_depth (2..4 runs in flight),
 *               chain_timeout_ms (bound of a stage-chain launch's halo waits)
 *   read-only (f8_net_get_option): err_mirror (1 = after upload a page-locked host mirror of the chain error words exists: f8_net_run returns
 *               F8_ERR_HIP without issuing anything once a chain launch of an EARLIER run gave up waiting, until f8_net_check collects the error;
 *               0 = the mirror could not be allocated: time-outs surface only through f8_net_check / poisoned logits.  Multi-rank callers: a rank
 *               that is refused must still enter the step's collective — INTEGRATION.md "rank divergence")
 * F8_ERR_INVALID for an unknown key or a value outside the key's range. */
int f8_net_set_option(f8_net* net, const char* key, int value);
int f8_net_get_option(const f8_net* net, const char* key, int* value);

/* f8_net_run cuts a batch of N into this many independent sub-batches (1..4) that it runs on
 * internal streams forked from / joined to `stream` (no host synchronisation); every planned launch
 * is therefore issued this many times per run, each over N / parts images. */
int f8_net_num_parts(const f8_net* net, int N);

/* Kernel launches planned launch `i` issues in a run of N images under the current schedule: one per sub-batch, times the
 * chunks of images the launches that run chunked (the fused bottleneck blocks; F8_CHUNK*) are cut into.
 * f8_net_run_profiled reports the SUM over these launches for launch i. */
int f8_net_step_launches(const f8_net* net, int i, int N);

/* Same, bracketing every launch with HIP events on `stream`; ms[i] = duration of launch i summed over
 * the sub-batches, which run back to back on `stream` here (i < f8_net_num_launches).  Synchronises the stream before returning. */
int f8_net_run_profiled(f8_net* net, const int32_t* input_dev, void* output_dev, int N,
                        void* stream, float* ms, int cap);

/* Per-launch facts for roofline accounting: name (kernel + layer key), algorithmic bytes and
 * integer ops (2*MAC) for a batch of N.  Returns F8_ERR_INVALID if i is out of range. */
int f8_net_launch_info(const f8_net* net, int i, int N, char* name, size_t name_cap,
                       double* alg_bytes, double* alg_ops);

/* Device kernel symbol of launch i as rocprofv3 --kernel-trace prints it (without the argument list). */
/* ESSENTIAL vector lane-operations of planned launch i for N images: what the reference's element-wise semantics between the matrix products need in
 * their cheapest exact integer form — 3 per int8 value produced (int_op_only_fix_quant, fix_quant_ops.py:99-112: tie bit, rounding add, 1/2 shift-
 * saturate-pack, 1/4 merge, 1/4 bias flip), 2 per joined int32 value (fix_resnet.py:40-54: align-add + clamp / ReLU), 1 per max-pooled value.
 * The profile tooling divides the counters' vector instructions x 64 lanes by it (profiles/rocprof_*_valu.md: issued / essential). */
int f8_net_launch_valu(const f8_net* net, int i, int N, double* essential_lane_ops);
int f8_net_launch_kernel(const f8_net* net, int i, char* buf, size_t cap);

/* Attach a label (e.g. the state_dict key) to the node that produced tensor `t`. */
int f8_net_set_label(f8_net* net, int t, const char* label);

#ifdef __cplusplus
}
#endif
#endif /* F8NET_H */
