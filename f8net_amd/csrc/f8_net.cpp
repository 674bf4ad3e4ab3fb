// f8_net.cpp — graph builder, planner and executor behind include/f8net.h.
//
// The builder records the reference-level integer graph (IntModel.forward,
// /root/reference/models/fix_resnet.py:352-383 and siblings).  f8_net_finalize turns it into a list
// of kernel launches:
//   * every conv's input requantisation (int_op_only_fix_quant, fix_quant_ops.py:90-114) moves into
//     the epilogue of the tensor's PRODUCER, which then emits int8 NHWC in the consumer's format
//     (up to two formats per producer; further ones fall back to a stand-alone requant launch);
//   * ReLU and the residual align-add-clamp (fix_resnet.py:40-54,77) ride in the conv epilogue;
//   * tensors keep an int32 NHWC form only where the reference semantics need 32 bits
//     (residual operands, pooling inputs, network outputs);
//   * a max-pool whose result is only consumed in one int8 format runs on int8 (requant is monotone,
//     so it commutes with max exactly).
// Planning touches no device, so it runs (and is tested) without a GPU.
#include "../../include/f8net.h"
#include "f8_internal.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <mutex>
#include <vector>

using namespace f8;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
int hip_fail(hipError_t e, const char* what) {
    return fail(F8_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline size_t round_up_z(size_t v, size_t m) { return (v + m - 1) / m * m; }

enum FormKind { FORM_I32 = 0, FORM_I8 = 1, FORM_STEM = 2 };
struct Form {
    int kind; int n; int sgn;          // I8: requant shift + signedness of the consumer format
    size_t bytes_per_img = 0;
    size_t slack = 0;                  // I32T: rows padded to a multiple of 32 pixels
    size_t off = 0;                    // arena offset (max_batch sized)
    int first = -1, last = -1;         // step lifetime
    int Hp = 0, Wp = 0, pad = 0;       // STEM
};
struct Tensor {
    int C = 0, H = 0, W = 0, Cs = 0, fl = 0;
    int prod = -1;
    mutable int64_t amax = -2;         // tensor_amax(), cached (-2: not computed yet)
    std::vector<int> consumers;        // node ids
    std::vector<Form> forms;
    std::string label;
    bool dense_out = false;            // written straight to the caller's output buffer
};
enum NodeKind { N_INPUT, N_CONV, N_ADD, N_MAXPOOL, N_AVGPOOL, N_LINEAR };
struct Node {
    int kind; int a = -1, b = -1; int out = -1;
    f8_conv_desc cd{};                 // conv / linear (as 1x1 conv)
    std::vector<int8_t> w; std::vector<int32_t> bias;   // raw OIHW int8 + bias
    mutable int acc_ok = -1;           // conv_acc_bounded(), cached
    mutable int64_t acc_max = -2;      // conv_acc_max(), cached (-2: not computed yet)
    int relu = 0;                      // add
    int pk = 0, pstride = 0, ppad = 0; // maxpool
    int shift = 0;                     // avgpool
    // planning
    int fused_into = -1;               // add: conv node that carries it
    int fused_add = -1;                // conv: add node carried
    bool stem = false, depthwise = false;
    int absorbed_by = -1;              // conv swallowed by a fused bottleneck launch (node id of its last conv)
    int fb_a = -1, fb_b = -1, fb_R = 0;  // last conv of a fused bottleneck: its first two convs, rows per tile
    int sp_pool = -1, sp_conv = -1;      // stem conv <-> max-pool fused into one launch (f8_stem.hip)
    int pool = -1, pool_host = -1;       // 1x1 conv (hosting its residual join or not) <-> the average pool behind it, one launch (f8_pool.hip)
    int fbd_a = -1, fbd_b = -1;          // shortcut conv hosting a fused stage-opening block (DS): body.0 / body.2 (body.4 = `dual`)
    bool fbd_s2 = false;                 // ... whose 3x3 and shortcut have stride 2 (f8_opener.hip)
    bool wreg = false;                   // 1x1 conv on conv1x1_wreg_kernel (f8_wreg.hip)
    bool s2w = false;                    // 3x3 / stride 2 conv on conv3x3s2_wreg_kernel (f8_s2conv.hip)
    bool wstat = false;                  // 1x1 conv / dual GEMM / residual join on conv1x1_wstat_kernel (f8_wstat.hip)
    int chain_into = -1;                 // host conv of a bottleneck block that runs inside a stage-chain launch: the host of the chain's LAST block
    std::vector<int> chain;              // host of the last block of a chain: the hosts of all its blocks, in order (f8_chain.hip)
    int bb_a = -1;                       // second 3x3 of a BasicBlock identity block that runs in a bchain launch: its first 3x3
    int bds_a = -1, bds_b = -1;          // 1x1 / 2 shortcut conv of a stage-opening BasicBlock that opens a bchain launch: its 3x3 / 2 and its second 3x3
    int bchain_into = -1;                // ... the host (second conv) of the chain's LAST block
    std::vector<int> bchain;             // host of the last block of a BasicBlock chain: the hosts of all its blocks (f8_bchain.hip)
    int p12_a = -1;                      // 3x3 conv hosting "1x1 -> 3x3 in one launch" (f8_p12.hip): its 1x1 producer
    bool p12_s2 = false;                 // ... the 3x3 has stride 2: body.0 + body.2 of a stage-opening block on f8_opener.hip (P12); the block's join opens the stage's chain
    bool tail = false;                   // shortcut conv (dual GEMM host) of such a block: its join runs as the FIRST block of a chain launch (ChainArgs::tail)
    int h2_head = -1, h2_dw = -1;        // 1x1 conv that ends the MobileNet-V2 head launch (f8_stem.hip, H2): its 3x3 / 2 head conv and its depthwise conv
    int ir_a = -1, ir_b = -1, ir_R = 0, ir_G = 0;   // project conv of a fused inverted-residual block: its expand / depthwise convs, tile
    int dual = -1;                     // 1x1 conv hosting a join whose other operand is ANOTHER 1x1 conv (node id): one dual-GEMM launch
    int dual_host = -1;                // ... and that other conv: the node that carries it
    int p3_R = 0, p3_imgs = 0, p3_bn = 0;   // 3x3 conv on the LDS-patch kernel (p3_R > 0): rows / images per tile, cout tile
    bool no_classes = false;           // pack a single bias class (the consumer kernel pads with real zeros itself)
    size_t w_off = 0, b_off = 0; int coutP = 0, ck = 0, ktot = 0;
    size_t wf_off = 0;                 // second image of the packed weights in MFMA-fragment order (pack_frag_weights), 0 = none
    size_t rc_off = 0, cc_off = 0; int ncc = 0;      // border-class tables (0 = single class)
    ConvTile tile{};
};
enum StepKind { S_INPUT, S_CONV, S_DW, S_ADD, S_MAXPOOL, S_AVGPOOL, S_REQUANT, S_OUTPUT, S_FUSED, S_STEMPOOL, S_IR, S_P12, S_CHAIN, S_BCHAIN, S_HEAD2 };
struct OutSel { int t = -1; int f32 = -1; int f8[2] = {-1, -1}; };
struct Step {
    int kind; int node;
    int src_t = -1, src_f = -1;        // main input tensor / form
    int res_t = -1, res_f = -1;        // residual (conv) or second operand (add)
    int src2_t = -1, src2_f = -1;      // dual GEMM: input of the second conv
    OutSel out;
    int acc_shl = 0, res_shl = 0, relu0 = 0, relu1 = 0;
    bool dense = false;
    bool raw_input = false;            // S_INPUT: its work is done by the stem launch (S_STEMPOOL with the same flag) unless the run's input is uint8 NHWC
    std::string name;
    mutable std::string kernel;        // device symbol as rocprofv3 prints it; chain steps: corrected by the first run from the instance the launcher really started
    double bytes_per_img = 0, bytes_const = 0, ops_per_img = 0;
    double valu_per_img = 0;           // ESSENTIAL vector lane-operations per image (f8_net_launch_valu): what the reference's semantics need once the MFMAs are done
};

}  // namespace

struct f8_net {
    std::vector<Tensor> tensors;
    std::vector<Node> nodes;
    int out_t = -1, out_float = 0;
    bool finalized = false;
    int max_batch = 0;
    Options opt;                       // per-handle tuning (f8_net_set_option); seeded from the environment at create
    int device = -1;                   // HIP device the arena / weights live on (set by f8_net_upload)
    int num_cu = 0;                    // its compute units (0 before upload: 256 assumed)
    int n_copies = 0;                  // arena copies allocated at upload (= opt.split then)
    hipEvent_t input_ready = nullptr;  // one-shot: the next run's first launch also waits for it (f8_net_set_input_ready)
    std::vector<Step> steps;
    std::vector<uint8_t> wblob;
    size_t arena_bytes = 0;
    size_t arena_stride = 0;           // device arena = kMaxParts copies (one per concurrent sub-batch)
    size_t stem_zero_off = 0, stem_zero_bytes = 0; int stem_zero_val = 0;   // halo = biased zero
    // device
    char* d_arena = nullptr; char* d_w = nullptr; bool uploaded = false;
    uint32_t* d_err = nullptr;         // sticky device error words: [0] an int32 input value outside the head's 8-bit format
    uint32_t* h_err = nullptr; uint32_t* h_err_dev = nullptr;   // host-visible mirror of the chain error words (pinned, mapped): read by f8_net_run without a synchronisation
    uint32_t epoch = 0;                // tag of the run being issued (1 .. 2^24 - 1, f8_net_run): chain error words carry it (ChainArgs::epoch)
    char* d_chain = nullptr; size_t chain_stride = 0;   // per arena copy: sync words + halo exchange rows of the stage-chain launches
    hipEvent_t* events = nullptr; int n_events = 0;
    bool aux_shared = false;           // aux[] belong to the per-device pool (run_common), not to this handle
    hipStream_t aux[4] = {nullptr, nullptr, nullptr, nullptr}; hipEvent_t aux_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t lag_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    const float* in_f32 = nullptr; float in_scale = 0.f; int in_lo = 0, in_hi = 0;   // set by f8_net_run_f32 for the duration of the call
    const uint8_t* in_u8 = nullptr; int in_u8_nhwc = 0; int16_t in_lut[3 * 256];       // set by f8_net_run_u8 for the duration of the call
    // pipelined submission (f8_net_set_pipelined): fork dependency = the event recorded at the PREVIOUS run's entry
    int pipelined = 0; hipEvent_t start_ev[4] = {nullptr, nullptr, nullptr, nullptr}; int start_idx = 0; int pipe_count = 0; hipStream_t prev_stream = nullptr;
    int alt_idx = 0;                   // pipelined == 2: internal stream / arena copy of the next run
    int chunk_off = 0;                 // image offset inside the arena while a chunk group runs (run_steps)
    // hipGraph of one whole run (both sub-batch streams), replayed while (input, output, N, stream) stay the same
    hipGraphExec_t g_exec = nullptr; const void* g_in = nullptr; void* g_out = nullptr; int g_N = 0; hipStream_t g_stream = nullptr; int g_warm = 0;
};

namespace {

int add_form(Tensor& t, int kind, int n, int sgn) {
    for (size_t i = 0; i < t.forms.size(); ++i)
        if (t.forms[i].kind == kind && (kind != FORM_I8 || (t.forms[i].n == n && t.forms[i].sgn == sgn))) return (int)i;
    Form f; f.kind = kind; f.n = n; f.sgn = sgn;
    t.forms.push_back(f);
    return (int)t.forms.size() - 1;
}
int find_form(const Tensor& t, int kind, int n, int sgn) {
    for (size_t i = 0; i < t.forms.size(); ++i)
        if (t.forms[i].kind == kind && (kind != FORM_I8 || (t.forms[i].n == n && t.forms[i].sgn == sgn))) return (int)i;
    return -1;
}

int check_t(const f8_net* net, int t, const char* who) {
    if (!net) return fail(F8_ERR_INVALID, "%s: null net", who);
    if (net->finalized) return fail(F8_ERR_STATE, "%s: net already finalized", who);
    if (t < 0 || t >= (int)net->tensors.size()) return fail(F8_ERR_INVALID, "%s: bad tensor id %d", who, t);
    return 0;
}

int new_tensor(f8_net* net, int C, int H, int W, int fl, int prod) {
    Tensor t; t.C = C; t.H = H; t.W = W; t.Cs = round_up(C, 32); t.fl = fl; t.prod = prod;
    net->tensors.push_back(t);
    return (int)net->tensors.size() - 1;
}

// Is every accumulator of this conv provably below 2^31 - 2^16 in magnitude?  |sum w x + b| <= sum |w| * max |x| + |b| with the 8-bit input
// range (255 unsigned / raw, 127 signed).  Then `v + 2^(n-1)` of the reference's int32 arithmetic (fix_quant_ops.py:100-104) cannot wrap for
// n <= 16 and the three-operation float requantisation (requant_u8x4, f8_device.h) is the same function; real nets are 10^2 - 10^3 below
// the limit, a net that is not keeps the integer form (generic kernel instances).  Cached per node.
constexpr int64_t kAccLimit = (int64_t(1) << 31) - (int64_t(1) << 16) - 1;
// max over the output channels of  sum |w| * max |x| + |b|  (-1: no weights to look at)
static int64_t conv_acc_max(const Node& nd) {
    if (nd.acc_max != -2) return nd.acc_max;
    int64_t m = -1;
    if (nd.cd.cout > 0 && !nd.w.empty() && nd.w.size() % (size_t)nd.cd.cout == 0) {
        const size_t per = nd.w.size() / (size_t)nd.cd.cout;
        const int64_t xmax = (nd.cd.quant_input && nd.cd.input_signed) ? 127 : 255;
        m = 0;
        for (int o = 0; o < nd.cd.cout; ++o) {
            int64_t s = 0;
            for (size_t k = 0; k < per; ++k) { const int v = nd.w[(size_t)o * per + k]; s += v < 0 ? -v : v; }
            const int64_t b = o < (int)nd.bias.size() ? (int64_t)nd.bias[o] : 0;
            m = std::max(m, s * xmax + (b < 0 ? -b : b));
        }
    }
    nd.acc_max = m;
    return m;
}
static bool conv_acc_bounded(const Node& nd) {
    if (nd.acc_ok >= 0) return nd.acc_ok != 0;
    const int64_t m = conv_acc_max(nd);
    nd.acc_ok = (m >= 0 && m <= kAccLimit) ? 1 : 0;
    return nd.acc_ok != 0;
}
// A bound on the magnitude of every value an int32 tensor can hold, from the graph alone (round 4): a conv's output by conv_acc_max (a ReLU behind
// it only shrinks it), a max-pool's by its input's, an align-add's (fix_resnet.py:56-76: the operand with the smaller fraclen is shifted left) by the
// shifted sum of its operands' — which makes the int32 residual STREAM of a ResNet stage a bounded quantity: real nets sit at 2^20 .. 2^27.  -1 =
// unknown or beyond 2^40 (the network input, pooled sums).  The chain launches requantise the stream through the float converter only below
// kAccLimit (ChainArgs::stream_ok); beyond it `v + 2^(n-1)` may wrap as the reference's int32 add does, and the integer instances run.
static int64_t tensor_amax(const f8_net* net, int t, int depth = 0) {
    if (t < 0 || t >= (int)net->tensors.size() || depth > 256) return -1;
    const Tensor& T = net->tensors[t];
    if (T.amax != -2) return T.amax;
    int64_t m = -1;
    if (T.prod >= 0) {
        const Node& nd = net->nodes[T.prod];
        if (nd.kind == N_CONV) m = conv_acc_max(nd);
        else if (nd.kind == N_MAXPOOL) m = tensor_amax(net, nd.a, depth + 1);
        else if (nd.kind == N_ADD) {
            const int64_t a = tensor_amax(net, nd.a, depth + 1), b = tensor_amax(net, nd.b, depth + 1);
            const int sa = T.fl - net->tensors[nd.a].fl, sb = T.fl - net->tensors[nd.b].fl;      // out fraclen = the larger one (f8_net_add)
            constexpr int64_t cap = int64_t(1) << 40;       // the shifted operands stay below 2^41: no int64 overflow (an operand of 2^40 shifted by 31 would wrap)
            if (a >= 0 && b >= 0 && sa >= 0 && sb >= 0 && sa <= 31 && sb <= 31 && a <= (cap >> sa) && b <= (cap >> sb)) m = (a << sa) + (b << sb);
        }
    }
    if (m > (int64_t(1) << 40)) m = -1;
    T.amax = m;
    return m;
}
static bool stream_bounded(const f8_net* net, int t) { const int64_t m = tensor_amax(net, t); return m >= 0 && m <= kAccLimit; }

// shift / clamp of a consumer's int_op_only_fix_quant; validates what the reference asserts
int consumer_format(const Tensor& src, const f8_conv_desc& d, int* n, const char* who) {
    const int maxfl = d.input_signed ? 7 : 8;
    if (d.input_fl < 0 || d.input_fl > maxfl)
        return fail(F8_ERR_INVALID, "%s: input_fl %d outside [0,%d] (fix_quant_ops.py:91-96)", who, d.input_fl, maxfl);
    *n = d.quant_input ? src.fl - d.input_fl : 0;
    if (*n > 30 || *n < -31) return fail(F8_ERR_INVALID, "%s: requant shift %d out of range", who, *n);
    return 0;
}

// round-up magic for unsigned division by d >= 1:  n / d == (t + ((n - t) >> sh1)) >> sh2,
// t = mulhi(n, magic); exact for all 32-bit n (Granlund-Montgomery, branch-free form)
void make_magic(uint32_t d, uint32_t* magic, int32_t* sh1, int32_t* sh2) {
    int l = 0;
    while ((1ull << l) < d) ++l;                       // l = ceil(log2 d)
    *magic = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    *sh1 = l < 1 ? l : 1;
    *sh2 = l > 1 ? l - 1 : 0;
}

int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

void set_q(QuantOut& q, char* ptr, const Form& f) {
    q.ptr = (int8_t*)ptr; q.n = f.n;
    q.lo = f.sgn ? -127 : 0; q.hi = f.sgn ? 127 : 255;
    q.bias_xor = f.sgn ? 0u : 0x80808080u;      // unsigned int8 tensors are stored biased (x ^ 0x80)
}

}  // namespace

namespace f8 {
struct OptKey { const char* key; const char* env; int Options::*slot; int lo, hi; bool planning; };
static const OptKey kOptKeys[] = {
    {"split", "F8_SPLIT", &Options::split, 1, 4, true},
    {"fuse_blocks", "F8_FUSE_BLOCKS", &Options::fuse_blocks, 0, 1, true},
    {"fuse_stages", "F8_FUSE_STAGES", &Options::fuse_stages, -1, 7, true},
    {"fuse_dual", "F8_FUSE_DUAL", &Options::fuse_dual, 0, 1, true},
    {"fuse_ds", "F8_FUSE_DS", &Options::fuse_ds, 0, 1, true},
    {"fuse_opener", "F8_FUSE_OPENER", &Options::fuse_opener, 0, 1, true},
    {"fuse_fc", "F8_FUSE_FC", &Options::fuse_fc, 0, 1, true},
    {"fuse_stem", "F8_FUSE_STEM", &Options::fuse_stem, 0, 1, true},
    {"fuse_input", "F8_FUSE_INPUT", &Options::fuse_input, 0, 1, true},
    {"fuse_ir", "F8_FUSE_IR", &Options::fuse_ir, 0, 2, true},
    {"fuse_head2", "F8_FUSE_HEAD2", &Options::fuse_head2, 0, 1, true},
    {"fuse_p12", "F8_FUSE_P12", &Options::fuse_p12, 0, 1, true},
    {"fuse_chain", "F8_FUSE_CHAIN", &Options::fuse_chain, 0, 1, true},
    {"fuse_tail", "F8_FUSE_TAIL", &Options::fuse_tail, 0, 1, true},
    {"fuse_chain7", "F8_FUSE_CHAIN7", &Options::fuse_chain7, 0, 1, true},
    {"fuse_pool", "F8_FUSE_POOL", &Options::fuse_pool, 0, 1, true},
    {"fuse_bchain", "F8_FUSE_BCHAIN", &Options::fuse_bchain, 0, 2, true},
    {"chain_timeout_ms", "F8_CHAIN_TIMEOUT_MS", &Options::chain_timeout_ms, 0, 1 << 20, false},
    {"wreg", "F8_WREG", &Options::wreg, 0, 1, true},
    {"s2wreg", "F8_S2WREG", &Options::s2wreg, 0, 1, true},
    {"wstat", "F8_WSTAT", &Options::wstat, 0, 1, true},
    {"wstat_min_tiles", "F8_WSTAT_MIN_TILES", &Options::wstat_min_tiles, 0, 1 << 20, true},
    {"wstat_fast", "F8_WSTAT_FAST", &Options::wstat_fast, 0, 1, true},
    {"patch3x3", "F8_PATCH3X3", &Options::patch3x3, 0, 1, true},
    {"dual_wide", "F8_DUAL_WIDE", &Options::dual_wide, 0, 1 << 30, true},
    {"deep_nk", "F8_DEEP_NK", &Options::deep_nk, 1, 1 << 20, true},
    {"bk128", "F8_BK128", &Options::bk128, 0, 1, true},
    {"dw_dot4", "F8_DW_DOT4", &Options::dw_dot4, 0, 1, true},
    {"dw_mma", "F8_DW_MMA", &Options::dw_mma, 0, 1, true},
    {"stem_wpc", "F8_STEM_WPC", &Options::stem_wpc, 1, 8, false},
    {"stem_rows", "F8_STEM_ROWS", &Options::stem_rows, 0, 1, true},
    {"stem_grid_div", "F8_STEM_GRID_DIV", &Options::stem_grid_div, 0, 32, false},
    {"opener_stg", "F8_OPENER_STG", &Options::opener_stg, 0, 1, true},
    {"chunk56", "F8_CHUNK", &Options::chunk56, -1, 1 << 20, false},
    {"chunk28", "F8_CHUNK28", &Options::chunk28, -1, 1 << 20, false},
    {"chunk14", "F8_CHUNK14", &Options::chunk14, -1, 1 << 20, false},
    {"chunk_budget_mb", "F8_CHUNK_BUDGET_MB", &Options::chunk_budget_mb, 1, 1 << 20, false},
    {"chunk_ds", "F8_CHUNK_DS", &Options::chunk_ds, 0, 1, false},
    {"chunk_opener", "F8_CHUNK_OPENER", &Options::chunk_opener, 0, 1, false},
    {"split_streams", "F8_SPLIT_STREAMS", &Options::split_streams, 0, 1, false},
    {"graph", "F8_GRAPH", &Options::graph, 0, 1, false},
    {"stagger", "F8_STAGGER", &Options::stagger, -1, 1 << 20, false},
    {"stagger_pipelined", "F8_STAGGER_PIPELINED", &Options::stagger_pipelined, 0, 1 << 20, false},
    {"check_device", "F8_CHECK_DEVICE", &Options::check_device, 0, 1, false},
    {"check_input_range", "F8_CHECK_INPUT_RANGE", &Options::check_input_range, 0, 1, false},
    {"requant_float", "F8_REQUANT_FLOAT", &Options::requant_float, 0, 1, true},
    {"pipeline_depth", "F8_PIPELINE_DEPTH", &Options::pipeline_depth, 2, 4, false},
    {"arena_copies", "F8_ARENA_COPIES", &Options::arena_copies, 0, 4, true},
    {"shared_streams", "F8_SHARED_STREAMS", &Options::shared_streams, 0, 1, true},
    {"whole_batch_launches", "F8_WHOLE_BATCH_LAUNCHES", &Options::whole_batch_launches, 0, 1, true},
};
static const OptKey* find_opt(const char* key) {
    if (!key) return nullptr;
    for (const auto& k : kOptKeys) if (!strcmp(k.key, key)) return &k;
    return nullptr;
}
void options_from_env(Options* o) {
    for (const auto& k : kOptKeys) {
        const int v = env_int(k.env, o->*(k.slot));
        o->*(k.slot) = v < k.lo ? k.lo : (v > k.hi ? k.hi : v);
    }
}
int* option_slot(Options* o, const char* key) { const OptKey* k = find_opt(key); return k ? &(o->*(k->slot)) : nullptr; }
}  // namespace f8

extern "C" {

const char* f8_status_string(int s) {
    switch (s) {
        case F8_OK: return "ok";
        case F8_ERR_INVALID: return "invalid argument";
        case F8_ERR_UNSUPPORTED: return "unsupported";
        case F8_ERR_HIP: return "HIP error";
        case F8_ERR_NOMEM: return "out of memory";
        case F8_ERR_STATE: return "bad state";
        default: return "unknown";
    }
}
const char* f8_last_error(void) { return g_err.c_str(); }
int f8_version(void) { return F8NET_VERSION; }
int f8_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

// ------------------------------------------------------------------------------ op level
int f8_requant_i32(const int32_t* src, int32_t* dst, size_t n, int src_fl, int dst_fl, int is_signed, void* stream) {
    if (dst_fl < 0 || dst_fl > (is_signed ? 7 : 8))
        return fail(F8_ERR_INVALID, "f8_requant_i32: fl %d outside [0,%d]", dst_fl, is_signed ? 7 : 8);
    const int sh = src_fl - dst_fl;
    if (sh > 30 || sh < -31) return fail(F8_ERR_INVALID, "f8_requant_i32: shift %d out of range", sh);
    if (n == 0) return F8_OK;
    if (!src || !dst) return fail(F8_ERR_INVALID, "f8_requant_i32: null pointer");
    hipError_t e = launch_requant_i32(src, dst, n, sh, is_signed ? -127 : 0, is_signed ? 127 : 255, (hipStream_t)stream);
    return e == hipSuccess ? F8_OK : hip_fail(e, "f8_requant_i32");
}
int f8_quantize_input_f32(const float* src, int32_t* dst, size_t n, int normalize, int fl, int is_signed, void* stream) {
    if (normalize && (fl < 0 || fl > (is_signed ? 7 : 8)))
        return fail(F8_ERR_INVALID, "f8_quantize_input_f32: fl %d outside [0,%d] (fix_quant_ops.py:66-71)", fl, is_signed ? 7 : 8);
    if (n == 0) return F8_OK;
    if (!src || !dst) return fail(F8_ERR_INVALID, "f8_quantize_input_f32: null pointer");
    hipError_t e = normalize ? launch_quantize_input(src, dst, n, (float)(1 << fl), is_signed ? -127 : 0, is_signed ? 127 : 255, (hipStream_t)stream)
                             : launch_quantize_input(src, dst, n, 255.f, INT32_MIN, INT32_MAX, (hipStream_t)stream);
    return e == hipSuccess ? F8_OK : hip_fail(e, "f8_quantize_input_f32");
}
int f8_topk_correct_f32(const float* logits, const int64_t* target, int N, int classes, const int* ks, int nk, float* correct, void* stream) {
    if (N < 0 || classes < 1 || nk < 1 || nk > 8 || !ks) return fail(F8_ERR_INVALID, "f8_topk_correct_f32: bad arguments");
    for (int k = 0; k < nk; ++k) if (ks[k] < 1 || ks[k] > classes) return fail(F8_ERR_INVALID, "f8_topk_correct_f32: k=%d outside [1,%d]", ks[k], classes);
    if (N == 0) return F8_OK;
    if (!logits || !target || !correct) return fail(F8_ERR_INVALID, "f8_topk_correct_f32: null pointer");
    // the k list is tiny and call-specific: it travels by value in the kernel arguments (no device scratch, no copy)
    TopkKs kv{};
    for (int k = 0; k < nk; ++k) kv.k[k] = ks[k];
    const hipError_t e = launch_topk_correct(logits, target, N, classes, kv, nk, correct, (hipStream_t)stream);
    return e == hipSuccess ? F8_OK : hip_fail(e, "f8_topk_correct_f32");
}
int f8_relu_i32(int32_t* x, size_t n, void* stream) {
    if (n == 0) return F8_OK;
    if (!x) return fail(F8_ERR_INVALID, "f8_relu_i32: null pointer");
    hipError_t e = launch_relu_i32(x, n, (hipStream_t)stream);
    return e == hipSuccess ? F8_OK : hip_fail(e, "f8_relu_i32");
}
int f8_add_align_i32(int32_t* res, const int32_t* x, size_t n, int res_fl, int x_fl, int* out_fl, void* stream) {
    const int d = res_fl - x_fl;
    if (d > 31 || d < -31) return fail(F8_ERR_INVALID, "f8_add_align_i32: fraclen gap %d too large", d);
    if (out_fl) *out_fl = std::max(res_fl, x_fl);
    if (n == 0) return F8_OK;
    if (!res || !x) return fail(F8_ERR_INVALID, "f8_add_align_i32: null pointer");
    hipError_t e = launch_add_align_i32(res, x, n, d < 0 ? -d : 0, d > 0 ? d : 0, (hipStream_t)stream);
    return e == hipSuccess ? F8_OK : hip_fail(e, "f8_add_align_i32");
}

// ------------------------------------------------------------------------------ builder
f8_net* f8_net_create(void) {
    f8_net* net = new (std::nothrow) f8_net();
    if (net) options_from_env(&net->opt);
    return net;
}

int f8_net_set_option(f8_net* net, const char* key, int value) {
    if (!net) return fail(F8_ERR_INVALID, "f8_net_set_option: null net");
    const OptKey* k = find_opt(key);
    if (!k) return fail(F8_ERR_INVALID, "f8_net_set_option: unknown key '%s'", key ? key : "(null)");
    if (value < k->lo || value > k->hi) return fail(F8_ERR_INVALID, "f8_net_set_option: %s = %d outside [%d,%d]", key, value, k->lo, k->hi);
    if (k->planning && net->finalized) return fail(F8_ERR_STATE, "f8_net_set_option: '%s' decides the plan; set it before f8_net_finalize", key);
    net->opt.*(k->slot) = value;
    return F8_OK;
}
int f8_net_get_option(const f8_net* net, const char* key, int* value) {
    if (!net || !value) return fail(F8_ERR_INVALID, "f8_net_get_option: null argument");
    // read-only state key: 1 = the host-visible mirror of the chain error words is active (f8_net_run then refuses further runs after a time-out
    // without a synchronisation); 0 = the page-locked allocation failed at upload / not uploaded yet: only f8_net_check reports a time-out
    if (key && !strcmp(key, "err_mirror")) { *value = (net->uploaded && net->h_err) ? 1 : 0; return F8_OK; }
    const OptKey* k = find_opt(key);
    if (!k) return fail(F8_ERR_INVALID, "f8_net_get_option: unknown key '%s'", key ? key : "(null)");
    *value = net->opt.*(k->slot);
    return F8_OK;
}
int f8_net_check(f8_net* net) {
    if (!net || !net->uploaded) return fail(F8_ERR_STATE, "f8_net_check: not uploaded");
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return hip_fail(e, "f8_net_check: hipDeviceSynchronize");
    if (net->d_err) {
        uint32_t w = 0;
        if ((e = hipMemcpy(&w, net->d_err, 4, hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(e, "f8_net_check: hipMemcpy");
        if (w) {
            (void)hipMemset(net->d_err, 0, 4);
            return fail(F8_ERR_INVALID, "f8_net_check: an int32 network input held values outside the head's 8-bit format (the reference would feed them to the "
                                        "head conv as they are; this library narrows the input to 8 bits): outputs of that run are invalid.  Option check_input_range = 0 disables the check");
        }
    }
    if (net->h_err) *(volatile uint32_t*)net->h_err = 0u;      // collected below (the device words stay the authority: one per arena copy)
    uint32_t first_w = 0; int first_p = -1;
    for (int p = 0; net->d_chain && p < net->n_copies; ++p) {
        uint32_t w = 0;
        if ((e = hipMemcpy(&w, net->d_chain + (size_t)p * net->chain_stride + kChainErrWord * 4, 4, hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(e, "f8_net_check: hipMemcpy");
        if (w) {
            // ticket, workgroups-out counter, halo flags and the error word of this arena copy: re-armed from the host (the device is idle —
            // the synchronize above — and a launch that did not run to completion would have left them dirty).  EVERY copy is collected by this call.
            (void)hipMemset(net->d_chain + (size_t)p * net->chain_stride, 0, 4096);
            if (first_p < 0) { first_w = w; first_p = p; }
        }
    }
    if (first_p >= 0)
        return fail(F8_ERR_HIP, "f8_net_check: a stage-chain launch gave up waiting for a neighbouring tile (code 0x%x of run tag %u, arena copy %d): the outputs of that run are invalid", first_w & 0xffu, first_w >> 8, first_p);
    return F8_OK;
}
int f8_net_set_input_ready(f8_net* net, void* event) {
    if (!net) return fail(F8_ERR_INVALID, "f8_net_set_input_ready: null net");
    net->input_ready = (hipEvent_t)event;
    return F8_OK;
}

void f8_net_destroy(f8_net* net) {
    if (!net) return;
    if (net->d_arena) (void)hipFree(net->d_arena);
    if (net->d_w) (void)hipFree(net->d_w);
    if (net->d_chain) (void)hipFree(net->d_chain);
    if (net->d_err) (void)hipFree(net->d_err);
    if (net->h_err) (void)hipHostFree(net->h_err);
    if (net->events) {
        for (int i = 0; i < net->n_events; ++i) (void)hipEventDestroy(net->events[i]);
        delete[] net->events;
    }
    for (int k = 0; k < 4; ++k) if (net->aux[k] && !net->aux_shared) (void)hipStreamDestroy(net->aux[k]);
    for (int k = 0; k < 5; ++k) if (net->aux_ev[k]) (void)hipEventDestroy(net->aux_ev[k]);
    for (int k = 0; k < 4; ++k) if (net->lag_ev[k]) (void)hipEventDestroy(net->lag_ev[k]);
    if (net->g_exec) (void)hipGraphExecDestroy(net->g_exec);
    for (int k = 0; k < 4; ++k) if (net->start_ev[k]) (void)hipEventDestroy(net->start_ev[k]);
    delete net;
}

int f8_net_input(f8_net* net, int C, int H, int W, int fraclen) {
    if (!net) return fail(F8_ERR_INVALID, "f8_net_input: null net");
    if (net->finalized) return fail(F8_ERR_STATE, "f8_net_input: finalized");
    if (C <= 0 || H <= 0 || W <= 0) return fail(F8_ERR_INVALID, "f8_net_input: bad shape");
    for (auto& n : net->nodes) if (n.kind == N_INPUT) return fail(F8_ERR_UNSUPPORTED, "f8_net_input: one input per net");
    Node nd; nd.kind = N_INPUT;
    net->nodes.push_back(nd);
    const int t = new_tensor(net, C, H, W, fraclen, (int)net->nodes.size() - 1);
    net->nodes.back().out = t;
    return t;
}

static int add_conv_node(f8_net* net, int src, const f8_conv_desc& d, const int32_t* w, const int32_t* b,
                         int kind, const char* who) {
    int rc = check_t(net, src, who);
    if (rc) return rc;
    const Tensor& s = net->tensors[src];
    if (d.cin != s.C) return fail(F8_ERR_INVALID, "%s: cin %d != source channels %d", who, d.cin, s.C);
    if (d.kernel < 1 || d.stride < 1 || d.pad < 0 || d.cout < 1) return fail(F8_ERR_INVALID, "%s: bad geometry", who);
    if (!(d.groups == 1 || (d.groups == d.cin && d.cout == d.cin)))
        return fail(F8_ERR_UNSUPPORTED, "%s: groups must be 1 or cin (depthwise)", who);
    if (d.groups != 1 && !(d.kernel == 3 && d.pad == 1))
        return fail(F8_ERR_UNSUPPORTED, "%s: depthwise is built for 3x3 pad 1", who);
    if (d.weight_fl < 0 || d.weight_fl > 31) return fail(F8_ERR_INVALID, "%s: bad weight_fl", who);
    int n;
    rc = consumer_format(s, d, &n, who);
    if (rc) return rc;
    const int P = (s.H + 2 * d.pad - d.kernel) / d.stride + 1, Q = (s.W + 2 * d.pad - d.kernel) / d.stride + 1;
    if (P < 1 || Q < 1) return fail(F8_ERR_INVALID, "%s: kernel larger than padded input", who);
    if (!w) return fail(F8_ERR_INVALID, "%s: null weight", who);
    Node nd; nd.kind = kind; nd.a = src; nd.cd = d;
    const size_t wn = (size_t)d.cout * (d.cin / d.groups) * d.kernel * d.kernel;
    nd.w.resize(wn);
    for (size_t i = 0; i < wn; ++i) {
        if (w[i] < -128 || w[i] > 127) return fail(F8_ERR_INVALID, "%s: weight %d does not fit int8", who, w[i]);
        nd.w[i] = (int8_t)w[i];
    }
    nd.bias.assign(d.cout, 0);
    if (b) for (int i = 0; i < d.cout; ++i) nd.bias[i] = b[i];
    net->nodes.push_back(std::move(nd));
    const int id = (int)net->nodes.size() - 1;
    const int t = new_tensor(net, d.cout, P, Q, d.weight_fl + d.input_fl, id);
    net->nodes[id].out = t;
    net->tensors[src].consumers.push_back(id);
    return t;
}

int f8_net_conv(f8_net* net, int src, const f8_conv_desc* desc, const int32_t* w, const int32_t* b) {
    if (!desc) return fail(F8_ERR_INVALID, "f8_net_conv: null desc");
    return add_conv_node(net, src, *desc, w, b, N_CONV, "f8_net_conv");
}

int f8_net_linear(f8_net* net, int src, const f8_linear_desc* d, const int32_t* w, const int32_t* b) {
    if (!d) return fail(F8_ERR_INVALID, "f8_net_linear: null desc");
    int rc = check_t(net, src, "f8_net_linear");
    if (rc) return rc;
    const Tensor& s = net->tensors[src];
    if (s.H != 1 || s.W != 1) return fail(F8_ERR_INVALID, "f8_net_linear: source must be [C,1,1]");
    f8_conv_desc c{};
    c.cin = d->in_features; c.cout = d->out_features; c.kernel = 1; c.stride = 1; c.pad = 0; c.groups = 1;
    c.weight_fl = d->weight_fl; c.input_fl = d->input_fl; c.input_signed = d->input_signed;
    c.quant_input = d->quant_input; c.relu = 0;
    return add_conv_node(net, src, c, w, b, N_LINEAR, "f8_net_linear");
}

int f8_net_add(f8_net* net, int a, int b, int relu) {
    int rc = check_t(net, a, "f8_net_add");
    if (rc) return rc;
    rc = check_t(net, b, "f8_net_add");
    if (rc) return rc;
    const Tensor &ta = net->tensors[a], &tb = net->tensors[b];
    if (ta.C != tb.C || ta.H != tb.H || ta.W != tb.W) return fail(F8_ERR_INVALID, "f8_net_add: shape mismatch");
    if (a == b) return fail(F8_ERR_UNSUPPORTED, "f8_net_add: operands must differ");
    if (std::abs(ta.fl - tb.fl) > 31) return fail(F8_ERR_INVALID, "f8_net_add: fraclen gap too large");
    Node nd; nd.kind = N_ADD; nd.a = a; nd.b = b; nd.relu = relu;
    net->nodes.push_back(nd);
    const int id = (int)net->nodes.size() - 1;
    const int t = new_tensor(net, ta.C, ta.H, ta.W, std::max(ta.fl, tb.fl), id);
    net->nodes[id].out = t;
    net->tensors[a].consumers.push_back(id);
    net->tensors[b].consumers.push_back(id);
    return t;
}

int f8_net_maxpool(f8_net* net, int src, int k, int stride, int pad) {
    int rc = check_t(net, src, "f8_net_maxpool");
    if (rc) return rc;
    const Tensor& s = net->tensors[src];
    if (k < 1 || stride < 1 || pad < 0 || pad * 2 > k) return fail(F8_ERR_INVALID, "f8_net_maxpool: bad geometry");
    const int P = (s.H + 2 * pad - k) / stride + 1, Q = (s.W + 2 * pad - k) / stride + 1;
    if (P < 1 || Q < 1) return fail(F8_ERR_INVALID, "f8_net_maxpool: window larger than input");
    Node nd; nd.kind = N_MAXPOOL; nd.a = src; nd.pk = k; nd.pstride = stride; nd.ppad = pad;
    net->nodes.push_back(nd);
    const int id = (int)net->nodes.size() - 1;
    const int t = new_tensor(net, s.C, P, Q, s.fl, id);
    net->nodes[id].out = t;
    net->tensors[src].consumers.push_back(id);
    return t;
}

int f8_net_avgpool_sum(f8_net* net, int src, int shift) {
    int rc = check_t(net, src, "f8_net_avgpool_sum");
    if (rc) return rc;
    const Tensor& s = net->tensors[src];
    if (shift < 0 || s.fl + shift > 32)   // fix_quant_ops.py:129 `assert output_fraclen <= 32`
        return fail(F8_ERR_INVALID, "f8_net_avgpool_sum: output fraclen %d > 32", s.fl + shift);
    Node nd; nd.kind = N_AVGPOOL; nd.a = src; nd.shift = shift;
    net->nodes.push_back(nd);
    const int id = (int)net->nodes.size() - 1;
    const int t = new_tensor(net, s.C, 1, 1, s.fl + shift, id);
    net->nodes[id].out = t;
    net->tensors[src].consumers.push_back(id);
    return t;
}

int f8_net_output(f8_net* net, int src, int as_float) {
    int rc = check_t(net, src, "f8_net_output");
    if (rc) return rc;
    if (net->out_t >= 0) return fail(F8_ERR_UNSUPPORTED, "f8_net_output: one output per net");
    net->out_t = src; net->out_float = as_float ? 1 : 0;
    return F8_OK;
}

int f8_net_set_pipelined(f8_net* net, int on) {
    if (!net) return fail(F8_ERR_INVALID, "f8_net_set_pipelined: null net");
    net->pipelined = on == 2 ? 2 : (on ? 1 : 0);
    net->pipe_count = 0;
    return F8_OK;
}

int f8_net_set_label(f8_net* net, int t, const char* label) {
    if (!net || t < 0 || t >= (int)net->tensors.size()) return fail(F8_ERR_INVALID, "f8_net_set_label: bad tensor");
    net->tensors[t].label = label ? label : "";
    return F8_OK;
}

// ------------------------------------------------------------------------------ planner
static void pack_conv_weights(f8_net* net, Node& nd, const Tensor& src, const Tensor& dst) {
    // Packed layout: [coutP][taps][CK] int8, K-contiguous per output channel, zero padded.
    // Generic: tap = (r,s), CK = Cs(src) channels.  Stem (cin <= 4, source = network input held as
    // haloed NHWC4): one "tap" per kernel ROW, CK = 32 bytes = 8 pixels x 4 channels of which
    // the first kw pixels / cin channels carry weights.
    // Bias: unsigned inputs are stored biased (x - 128), so b' = b + 128 * sum of w over the taps that
    // lie INSIDE the image.  Zero padding is fetched as biased 0 (= real 128) and must not be
    // corrected for taps outside; which taps are outside depends only on the output row / column, so
    // rows and columns are classified by their in-image tap mask and the table is
    // bias[rowclass * ncc + colclass][cout].  Signed inputs / unpadded convs have one class.
    const f8_conv_desc& d = nd.cd;
    const int k = d.kernel;
    nd.coutP = round_up(d.cout, 32);
    if (nd.stem) { nd.ck = 32; nd.ktot = k * 32; }
    else { nd.ck = src.Cs; nd.ktot = k * k * src.Cs; }
    nd.w_off = round_up_z(net->wblob.size(), 256);
    const size_t wbytes = (size_t)nd.coutP * nd.ktot;
    net->wblob.resize(nd.w_off + wbytes, 0);
    // per-tap weight sums
    std::vector<long long> tapsum((size_t)d.cout * k * k, 0);
    {
        int8_t* wp = (int8_t*)net->wblob.data() + nd.w_off;
        for (int o = 0; o < d.cout; ++o)
            for (int c = 0; c < d.cin; ++c)
                for (int r = 0; r < k; ++r)
                    for (int s = 0; s < k; ++s) {
                        const int8_t v = nd.w[(((size_t)o * d.cin + c) * k + r) * k + s];
                        tapsum[((size_t)o * k + r) * k + s] += v;
                        size_t idx;
                        if (nd.stem) idx = (size_t)o * nd.ktot + (size_t)r * 32 + (size_t)s * 4 + c;
                        else idx = (size_t)o * nd.ktot + ((size_t)(r * k + s)) * src.Cs + c;
                        wp[idx] = v;
                    }
    }
    // classes
    std::vector<uint32_t> rmasks, cmasks;
    std::vector<uint8_t> rowcls(dst.H, 0), colcls(dst.W, 0);
    const bool classes = !d.input_signed && d.pad > 0 && !nd.stem && !nd.no_classes;
    if (classes) {
        auto classify = [&](int n_out, int n_in, std::vector<uint32_t>& masks, std::vector<uint8_t>& cls) {
            for (int p = 0; p < n_out; ++p) {
                uint32_t mk = 0;
                for (int r = 0; r < k; ++r) { const int h = p * d.stride - d.pad + r; if (h >= 0 && h < n_in) mk |= 1u << r; }
                size_t id = 0;
                while (id < masks.size() && masks[id] != mk) ++id;
                if (id == masks.size()) masks.push_back(mk);
                cls[p] = (uint8_t)id;
            }
        };
        classify(dst.H, src.H, rmasks, rowcls);
        classify(dst.W, src.W, cmasks, colcls);
    } else {
        rmasks.push_back((1u << k) - 1); cmasks.push_back((1u << k) - 1);
    }
    nd.ncc = classes ? (int)cmasks.size() : 0;
    const size_t ncls = rmasks.size() * cmasks.size();
    nd.b_off = round_up_z(net->wblob.size(), 256);
    net->wblob.resize(nd.b_off + ncls * (size_t)nd.coutP * 4, 0);
    int32_t* bp = (int32_t*)(net->wblob.data() + nd.b_off);
    for (size_t rc = 0; rc < rmasks.size(); ++rc)
        for (size_t cc = 0; cc < cmasks.size(); ++cc)
            for (int o = 0; o < d.cout; ++o) {
                long long sum = 0;
                for (int r = 0; r < k; ++r)
                    for (int s = 0; s < k; ++s)
                        if (((rmasks[rc] >> r) & 1u) && ((cmasks[cc] >> s) & 1u)) sum += tapsum[((size_t)o * k + r) * k + s];
                uint32_t b = (uint32_t)nd.bias[o];
                if (!d.input_signed) b += (uint32_t)(128ll * sum);
                bp[(rc * cmasks.size() + cc) * (size_t)nd.coutP + o] = (int32_t)b;
            }
    if (classes) {
        nd.rc_off = round_up_z(net->wblob.size(), 256);
        net->wblob.resize(nd.rc_off + rowcls.size(), 0);
        memcpy(net->wblob.data() + nd.rc_off, rowcls.data(), rowcls.size());
        nd.cc_off = round_up_z(net->wblob.size(), 256);
        net->wblob.resize(nd.cc_off + colcls.size(), 0);
        memcpy(net->wblob.data() + nd.cc_off, colcls.data(), colcls.size());
    }
}

// A second image of a conv's packed weights ([coutP][ktot], K-contiguous) in MFMA-fragment order:
//   [cout tile of 32][K32 step][lane 0..63][16 B],  lane l = row (l & 31) of the tile, bytes [32 * step + 16 * (l >> 5), +16)
// — the A operand of one v_mfma_i32_32x32x32_i8 is one contiguous 1 KB, so a wave fetches it with one coalesced instruction straight
// into registers (f8_p12.hip: no LDS staging, no barrier in the K loop).
static void pack_frag_weights(f8_net* net, Node& nd) {
    const size_t bytes = (size_t)nd.coutP * nd.ktot;
    nd.wf_off = round_up_z(net->wblob.size(), 256);
    net->wblob.resize(nd.wf_off + bytes, 0);
    const int8_t* src = (const int8_t*)net->wblob.data() + nd.w_off;
    int8_t* dst = (int8_t*)net->wblob.data() + nd.wf_off;
    const int steps = nd.ktot / 32;
    for (int t = 0; t < nd.coutP / 32; ++t)
        for (int s = 0; s < steps; ++s)
            for (int l = 0; l < 64; ++l)
                memcpy(dst + (((size_t)t * steps + s) * 64 + l) * 16, src + (size_t)(t * 32 + (l & 31)) * nd.ktot + s * 32 + (l >> 5) * 16, 16);
}

static void pack_dw_weights(f8_net* net, Node& nd, const Tensor& src) {
    // Two images of the depthwise weights:
    //  (1) [9][Cs] tap-major + plain bias            — scalar kernel (int32 outputs, fallback)
    //  (2) [Cs/4][9] dwords for the dot4 kernel: per 4-channel quad, dword e (0..3) = taps 0..3 of channel e,
    //      dword 4+e = taps 4..7 of channel e, dword 8 = tap 8 of the 4 channels; bias + 128 * sum(w) when the
    //      input is unsigned (stored biased; out-of-image taps are replaced by the biased zero in the kernel).
    const f8_conv_desc& d = nd.cd;
    nd.coutP = src.Cs; nd.ck = 0; nd.ktot = 0;
    nd.w_off = round_up_z(net->wblob.size(), 256);
    nd.b_off = round_up_z(nd.w_off + (size_t)9 * src.Cs, 256);
    nd.rc_off = round_up_z(nd.b_off + (size_t)src.Cs * 4, 256);            // dot4 weights
    nd.cc_off = round_up_z(nd.rc_off + (size_t)(src.Cs / 4) * 9 * 4, 256);  // dot4 bias
    net->wblob.resize(nd.cc_off + (size_t)src.Cs * 4, 0);
    int8_t* wp = (int8_t*)net->wblob.data() + nd.w_off;
    int32_t* bp = (int32_t*)(net->wblob.data() + nd.b_off);
    int8_t* w4 = (int8_t*)net->wblob.data() + nd.rc_off;
    int32_t* b4 = (int32_t*)(net->wblob.data() + nd.cc_off);
    for (int c = 0; c < d.cout; ++c) {
        long long sum = 0;
        for (int t = 0; t < 9; ++t) {
            const int8_t v = nd.w[(size_t)c * 9 + t];
            wp[(size_t)t * src.Cs + c] = v;
            sum += v;
            int8_t* quad = w4 + (size_t)(c / 4) * 36;
            const int e = c % 4;
            if (t < 4) quad[e * 4 + t] = v;
            else if (t < 8) quad[16 + e * 4 + (t - 4)] = v;
            else quad[32 + e] = v;
        }
        bp[c] = nd.bias[c];
        uint32_t b = (uint32_t)nd.bias[c];
        if (!d.input_signed) b += (uint32_t)(128ll * sum);
        b4[c] = (int32_t)b;
    }
}

static std::string tname(const f8_net* net, int t) {
    const Tensor& T = net->tensors[t];
    if (!T.label.empty()) return T.label;
    char b[32]; snprintf(b, sizeof b, "t%d", t);
    return b;
}

// choose the forms a producer writes; extra int8 formats beyond two need the int32 form
static void select_outputs(f8_net* net, int t, OutSel* o, std::vector<int>* extra) {
    Tensor& T = net->tensors[t];
    o->t = t;
    int n8 = 0;
    for (size_t i = 0; i < T.forms.size(); ++i) if (T.forms[i].kind == FORM_I8) ++n8;
    if (n8 > 2) add_form(T, FORM_I32, 0, 0);
    o->f32 = find_form(T, FORM_I32, 0, 0);
    int k = 0;
    for (size_t i = 0; i < T.forms.size(); ++i)
        if (T.forms[i].kind == FORM_I8) {
            if (k < 2) o->f8[k++] = (int)i;
            else if (extra) extra->push_back((int)i);
        }
}

// Name (layer key + kernel variant) and device symbol (as rocprofv3 prints it) of a conv step; depends on the tile.
// Essential vector work of a launch (VERDICT r5 #1b: the counters' SQ_INSTS_VALU x 64 divided by this = issued / essential): per value of the
// reference's semantics, with the cheapest exact integer forms of f8_device.h —
//   an int8 value produced (in HBM or only in LDS): round-half-even shift + clamp + pack = 3 (v_bfe_u32, v_add3_u32, 1/2 v_ashr_pk_u8_i32, 1/4 v_perm_b32,
//   1/4 v_xor_b32; the ReLU in front of it is the clamp's lower bound); a joined int32 value: align-add + clamp / ReLU = 2; a max-pooled conv value: 1.
// Addressing, lane swaps, exec masks, halo code, tile padding and recompute are NOT in it: they are what the ratio shows.
static double out_forms8(const Step& st) { return (double)((st.out.f8[0] >= 0) + (st.out.f8[1] >= 0)); }
static void label_conv_step(f8_net* net, Step& st, const Node& nd) {
    auto& ND = net->nodes;
    const f8_conv_desc& d = nd.cd;
    const Tensor& s = net->tensors[nd.a];
    char buf[160];
    if (nd.depthwise) snprintf(buf, sizeof buf, "dwconv3x3s%d:%s", d.stride, tname(net, nd.out).c_str());
    else snprintf(buf, sizeof buf, "conv%dx%ds%d_t%dx%dx%d%s%s:%s", d.kernel, d.kernel, d.stride, nd.tile.bm, nd.tile.bn,
                  nd.tile.bk, nd.stem ? "_stem" : "", st.res_t >= 0 ? "_res" : (nd.dual >= 0 ? "_dual" : ""),
                  (nd.dual >= 0 ? tname(net, ND[nd.dual].out) + "+" + tname(net, nd.out) : tname(net, nd.out)).c_str());
    if (nd.p3_R > 0) snprintf(buf, sizeof buf, "conv3x3s1_patch_R%dx%d_bn%d%s:%s", nd.p3_R, nd.p3_imgs, nd.p3_bn, st.res_t >= 0 ? "_res" : "",
                              tname(net, nd.out).c_str());
    st.name = buf;
    if (nd.depthwise) {
        const Tensor& od = net->tensors[nd.out];         // keep in sync with launch_dwconv / dwconv_mma_supported / launch_dwconv_mma (FQ)
        int fq = d.relu ? ((conv_acc_bounded(nd) && net->opt.requant_float) ? 1 : 2) : 0;
        int n8 = 0;
        for (int k = 0; k < 2; ++k) if (st.out.f8[k] >= 0) { const Form& F = od.forms[st.out.f8[k]]; ++n8; if (!(F.n > 0 && F.n <= 30 && !F.sgn)) fq = 0; else if (fq == 1 && F.n > 16) fq = 2; }
        const bool mma = net->opt.dw_mma && st.out.f32 < 0 && n8 > 0 && d.pad == 1 && (d.stride == 1 || d.stride == 2) && (od.W >= 28 || od.W == 14) &&
                         (d.stride == 1 ? (od.H == s.H && od.W == s.W) : (s.H == 2 * od.H && s.W == 2 * od.W));
        if (mma) snprintf(buf, sizeof buf, "f8::dwconv3x3_mma_kernel<%d, %d, %d>", d.stride, fq, od.W >= 28 ? 1 : 2);
        else snprintf(buf, sizeof buf, "f8::dwconv3x3_dot4_kernel<%d, 2>", d.stride);
    }
    else if (nd.p3_R > 0) snprintf(buf, sizeof buf, "f8::conv3x3_patch_kernel<%d, %d, %d, %d, %d, %d, %s>", d.cin, s.W, nd.p3_R, nd.p3_imgs, nd.p3_bn,
                                   d.cin == 64 ? 64 : (nd.p3_bn == 128 ? 128 : 256), st.res_t >= 0 ? "true" : "false");   // keep in sync with launch_conv3x3_patch
    else {
        const int wpx = (nd.tile.bm == 128 && nd.tile.bn <= 64) ? 4 : 2, wco = 4 / wpx;
        // keep in sync with launch_conv_t (f8_kernels.hip)
        const int tile_b = (nd.tile.bm + nd.tile.bn) * nd.tile.bk;
        const int dst = (4 * tile_b <= 65536) ? 4 : ((3 * tile_b <= 65536) ? 3 : 2);
        const int ksteps = (nd.ktot + (nd.dual >= 0 ? ND[nd.dual].ktot : 0)) / nd.tile.bk;
        const int stages = (dst > 2 && ksteps >= net->opt.deep_nk) ? dst : 2;    // ring depth rule of launch_conv_t
        snprintf(buf, sizeof buf, "f8::conv_igemm_kernel<%d, %d, %d, %d, %d, %s, %s, %d, %s>", nd.tile.bm, nd.tile.bn, nd.tile.bk, wpx, wco,
                 (d.pad > 0 && !nd.stem) ? "true" : "false", (st.res_t >= 0 || nd.dual >= 0) ? "true" : "false", stages,
                 nd.dual >= 0 ? "true" : "false");
    }
    st.kernel = buf;
}

// ================================================================================================================================
// The planner: f8_net_finalize runs these passes in order.  Passes 1 .. 1i only MARK nodes (which launch hosts which conv);
// pass 2 decides the forms each tensor needs in HBM, pass 3 emits the launches, pass 4 lays the arena out.  Round 5: one function per pass
// (round 4: one 1100-line function).
// ================================================================================================================================

// position of a block (by its host conv) inside its stage chain / BasicBlock chain, -1: not chained
static int chain_pos(const f8_net* net, int host) {
    const auto& ND = net->nodes;
    if (host < 0 || ND[host].chain_into < 0) return -1;
    const std::vector<int>& ch = ND[ND[host].chain_into].chain;
    for (size_t k = 0; k < ch.size(); ++k) if (ch[k] == host) return (int)k;
    return -1;
}
static int bchain_pos(const f8_net* net, int host) {
    const auto& ND = net->nodes;
    if (host < 0 || ND[host].bchain_into < 0) return -1;
    const std::vector<int>& ch = ND[ND[host].bchain_into].bchain;
    for (size_t k = 0; k < ch.size(); ++k) if (ch[k] == host) return (int)k;
    return -1;
}

// pass 1
static void plan_residual_joins(f8_net* net, int max_batch) {
    auto& T = net->tensors;
    auto& ND = net->nodes;
    const int nn = (int)ND.size();
    const Options& opt = net->opt;
    const int fuse_blocks = opt.fuse_blocks;
    (void)T; (void)nn; (void)opt; (void)fuse_blocks; (void)max_batch;
    // ---- 1. residual fusion: an add rides in the epilogue of the LATER of its two producers when
    //         that producer is an MFMA conv nobody else reads and the other operand is already there.
    for (int i = 0; i < nn; ++i) {
        Node& ad = ND[i];
        if (ad.kind != N_ADD) continue;
        for (int pass = 0; pass < 2 && ad.fused_into < 0; ++pass) {
            const int x = pass == 0 ? ad.a : ad.b, y = pass == 0 ? ad.b : ad.a;
            const int px = T[x].prod, py = T[y].prod;
            Node& cx = ND[px];
            if (cx.kind != N_CONV || cx.cd.groups != 1 || cx.fused_add >= 0) continue;
            if (T[x].consumers.size() != 1 || x == net->out_t) continue;
            if (py > px) continue;     // other operand must exist before the conv runs
            ad.fused_into = px; cx.fused_add = i;
        }
    }

}

// pass 1b
static void plan_bottleneck_blocks(f8_net* net, int max_batch) {
    auto& T = net->tensors;
    auto& ND = net->nodes;
    const int nn = (int)ND.size();
    const Options& opt = net->opt;
    const int fuse_blocks = opt.fuse_blocks;
    (void)T; (void)nn; (void)opt; (void)fuse_blocks; (void)max_batch;
    // ---- 1b. whole-block fusion: 1x1(ReLU) -> 3x3 pad 1 (ReLU) -> 1x1 + residual with the block input,
    //          all stride 1, intermediates read by nobody else  ->  one launch (f8_fused.hip)
    for (int i = 0; fuse_blocks && i < nn; ++i) {
        Node& c = ND[i];
        if (c.kind != N_CONV || c.fused_add < 0 || c.cd.groups != 1 || c.cd.kernel != 1 || c.cd.stride != 1 || c.cd.pad != 0 || c.cd.relu) continue;
        const Tensor& tb = T[c.a];
        if (tb.consumers.size() != 1 || c.a == net->out_t || !c.cd.quant_input) continue;
        Node& b = ND[tb.prod];
        if (b.kind != N_CONV || b.fused_add >= 0 || b.cd.groups != 1 || b.cd.kernel != 3 || b.cd.stride != 1 || b.cd.pad != 1 || !b.cd.quant_input) continue;
        const Tensor& ta = T[b.a];
        if (ta.consumers.size() != 1 || b.a == net->out_t) continue;
        Node& a0 = ND[ta.prod];
        if (a0.kind != N_CONV || a0.fused_add >= 0 || a0.cd.groups != 1 || a0.cd.kernel != 1 || a0.cd.stride != 1 || a0.cd.pad != 0) continue;
        const Node& ad = ND[c.fused_add];
        const int other = (ad.a == c.out) ? ad.b : ad.a;
        if (other != a0.a) continue;                          // residual operand must be the block input
        const Tensor& x = T[a0.a];
        const int C = a0.cd.cin, MID = a0.cd.cout;
        if (c.cd.cout != C || b.cd.cin != MID || b.cd.cout != MID || c.cd.cin != MID) continue;
        int R = 0;
        const bool chainable = opt.fuse_chain && a0.cd.quant_input && chain_supported(C, MID, x.H, x.W, C) &&    // pass 1f decides; R = 0: no stand-alone launch
                               (opt.fuse_chain7 || !cchain_supported(C, MID, x.H, x.W, C, false));
        if (!fused_bottleneck_supported(C, MID, x.H, x.W, opt.whole_batch_launches ? max_batch : std::max(1, max_batch / opt.split), opt.fuse_stages, &R) && !chainable) {
            // no whole-block instance (the 7x7 maps of stage 3): body.0 + body.2 as one launch, the residual-carrying 1x1 stays
            if (opt.fuse_p12 && fused_p12_supported(C, MID, x.H, x.W) && a0.cd.relu && b.cd.relu && tb.consumers.size() == 1) {
                a0.absorbed_by = tb.prod; b.p12_a = ta.prod; b.no_classes = true;
            }
            continue;
        }
        a0.absorbed_by = i; b.absorbed_by = i; b.no_classes = true;
        c.fb_a = ta.prod; c.fb_b = tb.prod; c.fb_R = R;
    }

}

// pass 1c
static void plan_dual_gemm_joins(f8_net* net, int max_batch) {
    auto& T = net->tensors;
    auto& ND = net->nodes;
    const int nn = (int)ND.size();
    const Options& opt = net->opt;
    const int fuse_blocks = opt.fuse_blocks;
    (void)T; (void)nn; (void)opt; (void)fuse_blocks; (void)max_batch;
    // ---- 1c. downsample join: the add's two operands are both 1x1 / pad 0 convs (body.4 and the shortcut) and the
    //          earlier one feeds nothing else  ->  one dual-GEMM launch, its int32 result never touches HBM
    for (int i = 0; opt.fuse_dual && i < nn; ++i) {
        Node& h = ND[i];
        if (h.kind != N_CONV || h.fused_add < 0 || h.fb_a >= 0 || h.cd.groups != 1 || h.cd.kernel != 1 || h.cd.pad != 0) continue;
        if (h.cd.cin % 64 != 0 || round_up(h.cd.cout, 32) <= 32) continue;          // kernel instances: BK = 64, BN = 64
        const Node& ad = ND[h.fused_add];
        const int other = (ad.a == h.out) ? ad.b : ad.a;
        if (T[other].consumers.size() != 1 || other == net->out_t) continue;
        Node& g = ND[T[other].prod];
        if (g.kind != N_CONV || g.fused_add >= 0 || g.absorbed_by >= 0 || g.fb_a >= 0 || g.cd.groups != 1 || g.cd.kernel != 1 || g.cd.pad != 0 ||
            g.cd.relu || g.cd.cin % 64 != 0 || g.cd.cout != h.cd.cout) continue;
        if (T[g.out].H != T[h.out].H || T[g.out].W != T[h.out].W) continue;
        h.dual = T[other].prod; g.dual_host = i;
    }

}

// pass 1d
static void plan_stage_opening_blocks(f8_net* net, int max_batch) {
    auto& T = net->tensors;
    auto& ND = net->nodes;
    const int nn = (int)ND.size();
    const Options& opt = net->opt;
    const int fuse_blocks = opt.fuse_blocks;
    (void)T; (void)nn; (void)opt; (void)fuse_blocks; (void)max_batch;
    // ---- 1d. stage-opening bottleneck at unchanged resolution: body.0 -> body.2 -> [body.4 + shortcut] in ONE launch
    for (int i = 0; fuse_blocks && i < nn; ++i) {
        Node& h = ND[i];
        if (h.kind != N_CONV || h.dual < 0 || (h.cd.stride != 1 && h.cd.stride != 2) || h.cd.relu || !h.cd.quant_input) continue;
        const int bs = h.cd.stride;                            // stride of the shortcut = stride of the block's 3x3
        Node& g = ND[h.dual];
        if (g.cd.stride != 1 || !g.cd.quant_input) continue;
        const Tensor& tb = T[g.a];
        if (tb.consumers.size() != 1 || g.a == net->out_t) continue;
        Node& b = ND[tb.prod];
        if (b.kind != N_CONV || b.fused_add >= 0 || b.absorbed_by >= 0 || b.cd.groups != 1 || b.cd.kernel != 3 || b.cd.stride != bs || b.cd.pad != 1 ||
            !b.cd.quant_input) continue;
        const Tensor& ta = T[b.a];
        if (ta.consumers.size() != 1 || b.a == net->out_t) continue;
        Node& a0 = ND[ta.prod];
        if (a0.kind != N_CONV || a0.fused_add >= 0 || a0.absorbed_by >= 0 || a0.cd.groups != 1 || a0.cd.kernel != 1 || a0.cd.stride != 1 ||
            a0.cd.pad != 0 || !a0.cd.quant_input || a0.a != h.a || a0.dual_host >= 0 || a0.dual >= 0) continue;
        const Tensor& x = T[h.a];
        int na = 0, nh = 0;
        if (consumer_format(x, a0.cd, &na, "finalize") || consumer_format(x, h.cd, &nh, "finalize")) continue;
        if (na != nh || a0.cd.input_signed != h.cd.input_signed) continue;      // one int8 form of the block input serves both
        const int C = a0.cd.cin, MID = a0.cd.cout;
        if (b.cd.cin != MID || b.cd.cout != MID || g.cd.cin != MID || g.cd.cout != h.cd.cout || h.cd.cin != C) continue;
        int R = 0;
        if (bs == 1 ? !(opt.fuse_ds && fused_ds_supported(C, MID, h.cd.cout, x.H, x.W, &R))
                    : !(opt.fuse_opener && fused_opener_supported(C, MID, h.cd.cout, x.H, x.W, &R))) continue;
        a0.absorbed_by = i; b.absorbed_by = i; b.no_classes = true;
        h.fbd_a = ta.prod; h.fbd_b = tb.prod; h.fb_R = R; h.fbd_s2 = bs == 2;
    }

}

// pass 1f
static void plan_stage_chains(f8_net* net, int max_batch) {
    auto& T = net->tensors;
    auto& ND = net->nodes;
    const int nn = (int)ND.size();
    const Options& opt = net->opt;
    const int fuse_blocks = opt.fuse_blocks;
    (void)T; (void)nn; (void)opt; (void)fuse_blocks; (void)max_batch;
    // ---- 1f. stage chains: consecutive bottleneck blocks at one resolution whose block-to-block tensors are read by nobody else
    //          -> ONE launch for all of them, the int32 residual stream stays in registers (f8_chain.hip).  A chain may start with
    //          the stage-opening block at unchanged resolution (1d) and continues through identity blocks (1b).
    if (opt.fuse_chain && fuse_blocks) {
        struct Blk { int host, in_t, out_t, C, MID, H, W, cin0; bool ds; bool tail = false; };
        auto block_of = [&](int i, Blk* bk) -> bool {
            const Node& h = ND[i];
            if (h.kind != N_CONV || h.fused_add < 0 || h.chain_into >= 0) return false;
            if (h.fb_a >= 0) {
                const Node& a0 = ND[h.fb_a]; const Tensor& x = T[a0.a];
                *bk = Blk{i, a0.a, ND[h.fused_add].out, a0.cd.cin, a0.cd.cout, x.H, x.W, a0.cd.cin, false};
                return true;
            }
            if (h.fbd_a >= 0 && !h.fbd_s2) {
                const Node& a0 = ND[h.fbd_a]; const Tensor& x = T[h.a];
                *bk = Blk{i, h.a, ND[h.fused_add].out, h.cd.cout, a0.cd.cout, x.H, x.W, a0.cd.cin, true};
                return true;
            }
            if (h.fbd_a >= 0 && h.fbd_s2 && opt.fuse_tail) {
                // stage-opening block with a stride-2 3x3 (1d fused it on f8_opener.hip): candidate for body.0 + body.2 there and the JOIN as the
                // first block of the stage's chain (geometry of the chain = the block's OUTPUT map)
                // chain_kernel<TAIL> addresses the shortcut's operand as [N][2H][2W][CIN0]: only an input map of exactly twice the output's sides
                // (a 55x55 map also gives 28x28: the generic dual GEMM takes that one)
                const Node& a0 = ND[h.fbd_a]; const Tensor& y = T[ND[h.fused_add].out]; const Tensor& x = T[h.a];
                if (x.H == 2 * y.H && x.W == 2 * y.W) {
                    *bk = Blk{i, h.a, ND[h.fused_add].out, h.cd.cout, a0.cd.cout, y.H, y.W, a0.cd.cin, true, true};
                    return true;
                }
            }
            if (h.dual >= 0 && h.fbd_a < 0 && h.cd.stride == 2 && h.cd.kernel == 1 && h.cd.quant_input && opt.fuse_tail) {
                // the same for an opening block whose convs run as separate launches (stages 2 / 3 of ResNet-50; stage 1 when body.0 and the shortcut
                // read different int8 forms): the dual-GEMM join (1c) becomes the chain's first block, body.2's int8 output is its mid2
                const Node& g = ND[h.dual]; const Tensor& y = T[ND[h.fused_add].out]; const Tensor& x = T[h.a]; const Tensor& m2 = T[g.a];
                if (g.cd.kernel == 1 && g.cd.stride == 1 && g.cd.quant_input && !g.cd.relu && !h.cd.relu && h.cd.pad == 0 && g.cd.pad == 0 &&
                    x.H == 2 * y.H && x.W == 2 * y.W && m2.H == y.H && m2.W == y.W) {
                    *bk = Blk{i, h.a, ND[h.fused_add].out, h.cd.cout, g.cd.cin, y.H, y.W, h.cd.cin, true, true};
                    return true;
                }
            }
            return false;
        };
        for (int i = 0; i < nn; ++i) {
            Blk first;
            if (!block_of(i, &first)) continue;
            if (first.tail ? !chain_tail_supported(first.C, first.MID, first.H, first.W, first.cin0) : !chain_supported(first.C, first.MID, first.H, first.W, first.cin0)) continue;
            if (!opt.fuse_chain7 && cchain_supported(first.C, first.MID, first.H, first.W, first.cin0, first.tail)) continue;
            std::vector<int> hosts{first.host};
            Blk cur = first;
            const int max_blocks = chain_max_blocks(first.C, first.MID, first.H, first.W, first.cin0, first.tail);
            while ((int)hosts.size() < max_blocks && cur.out_t != net->out_t) {
                const Tensor& y = T[cur.out_t];
                if (y.consumers.size() != 2) break;
                int next = -1;
                for (int j = cur.host + 1; j < nn && next < 0; ++j) {
                    Blk nb;
                    if (block_of(j, &nb) && !nb.ds && nb.in_t == cur.out_t) next = j;
                }
                if (next < 0) break;
                Blk nb; block_of(next, &nb);
                if (nb.C != first.C || nb.MID != first.MID || nb.H != first.H || nb.W != first.W) break;
                const int c0 = y.consumers[0], c1 = y.consumers[1], want0 = ND[next].fb_a, want1 = ND[next].fused_add;
                if (!((c0 == want0 && c1 == want1) || (c0 == want1 && c1 == want0))) break;
                hosts.push_back(next); cur = nb;
            }
            if (hosts.size() < 2) continue;
            const int last = hosts.back();
            for (int h : hosts) ND[h].chain_into = last;
            ND[last].chain = hosts;
            if (first.tail) {
                // the opener becomes: [body.0 + body.2 on f8_opener.hip (P12), host = the 3x3, output = mid2 in body.4's int8 format] + [the chain's first block]
                Node& h = ND[first.host];
                if (h.fbd_a >= 0) {
                    const int ia = h.fbd_a, ib = h.fbd_b;
                    ND[ia].absorbed_by = ib; ND[ib].absorbed_by = -1; ND[ib].p12_a = ia; ND[ib].p12_s2 = true; ND[ib].fb_R = h.fb_R;
                    h.fbd_a = h.fbd_b = -1; h.fbd_s2 = false;
                }
                h.tail = true;
            }
        }
        // identity blocks that only the chain kernel could run and that did not end up in a chain: back to separate launches
        for (int i = 0; i < nn; ++i) {
            Node& c = ND[i];
            if (c.kind == N_CONV && c.fb_a >= 0 && c.fb_R == 0 && c.chain_into < 0) {
                ND[c.fb_a].absorbed_by = -1; ND[c.fb_b].absorbed_by = -1; ND[c.fb_b].no_classes = false;
                c.fb_a = c.fb_b = -1;
            }
        }
    }
}

// pass 1g
static void plan_basic_block_chains(f8_net* net, int max_batch) {
    auto& T = net->tensors;
    auto& ND = net->nodes;
    const int nn = (int)ND.size();
    const Options& opt = net->opt;
    const int fuse_blocks = opt.fuse_blocks;
    (void)T; (void)nn; (void)opt; (void)fuse_blocks; (void)max_batch;
    // ---- 1g. BasicBlock identity blocks (3x3 ReLU -> 3x3 + residual with the block input, all stride 1) -> f8_bchain.hip: ONE launch for
    //          the consecutive ones of a stage, the int32 stream in registers (a single block is a chain of one: both convs in one
    //          launch, `mid` only in LDS)
    if (opt.fuse_bchain && fuse_blocks) {
        auto bblock = [&](int i, int* in_t, int* out_t, int* C) -> bool {
            const Node& c2 = ND[i];
            if (c2.kind != N_CONV || c2.fused_add < 0 || c2.bchain_into >= 0 || c2.fb_a >= 0 || c2.fbd_a >= 0 || c2.dual >= 0 || c2.dual_host >= 0 ||
                c2.absorbed_by >= 0 || c2.cd.groups != 1 || c2.cd.kernel != 3 || c2.cd.stride != 1 || c2.cd.pad != 1 || c2.cd.relu || !c2.cd.quant_input) return false;
            const Tensor& tb = T[c2.a];
            if (tb.consumers.size() != 1 || c2.a == net->out_t) return false;
            const Node& c1 = ND[tb.prod];
            if (c1.kind != N_CONV || c1.fused_add >= 0 || c1.absorbed_by >= 0 || c1.cd.groups != 1 || c1.cd.kernel != 3 || c1.cd.stride != 1 ||
                c1.cd.pad != 1 || !c1.cd.quant_input || c1.dual >= 0 || c1.dual_host >= 0) return false;
            const Node& ad = ND[c2.fused_add];
            const int other = (ad.a == c2.out) ? ad.b : ad.a;
            if (other != c1.a) return false;
            const int cc = c1.cd.cin;
            if (c1.cd.cout != cc || c2.cd.cin != cc || c2.cd.cout != cc || cc % 32) return false;
            if (ND[T[c1.a].prod].kind == N_INPUT) return false;
            if (!bchain_supported(cc, T[c1.a].H, T[c1.a].W)) return false;
            *in_t = c1.a; *out_t = ad.out; *C = cc;
            return true;
        };
        for (int i = 0; i < nn; ++i) {
            int in_t, out_t, C;
            if (!bblock(i, &in_t, &out_t, &C)) continue;
            std::vector<int> hosts{i};
            const bool sgn0 = ND[T[ND[i].a].prod].cd.input_signed;
            int cur_out = out_t;
            while ((int)hosts.size() < kBChainMaxBlocks && cur_out != net->out_t) {
                const Tensor& y = T[cur_out];
                if (y.consumers.size() != 2) break;
                int next = -1, nin = -1, nout = -1, nC = 0;
                for (int j = hosts.back() + 1; j < nn && next < 0; ++j)
                    if (bblock(j, &nin, &nout, &nC) && nin == cur_out) next = j;
                if (next < 0 || nC != C) break;
                const int c1n = T[ND[next].a].prod;
                if (ND[c1n].cd.input_signed != sgn0) break;              // one border pattern for the int8 copy of the stream
                const int k0 = y.consumers[0], k1 = y.consumers[1], w0 = c1n, w1 = ND[next].fused_add;
                if (!((k0 == w0 && k1 == w1) || (k0 == w1 && k1 == w0))) break;
                hosts.push_back(next); cur_out = nout;
            }
            const int lastn = hosts.back();
            for (int h : hosts) {
                Node& c2 = ND[h]; Node& c1 = ND[T[c2.a].prod];
                c2.bchain_into = lastn; c2.bb_a = T[c2.a].prod;
                c1.absorbed_by = h; c1.no_classes = true; c2.no_classes = true;
            }
            // the stage-opening block in front of them (3x3 / 2 ReLU -> 3x3, 1x1 / 2 shortcut, join) joins the launch when the chain is
            // the only reader of its output (its two convs over the block input may read different int8 forms of it)
            [&] {
                if (opt.fuse_bchain < 2 || (int)hosts.size() >= kBChainMaxBlocks || in_t == net->out_t) return;
                const Tensor& y = T[in_t];
                const int c1f = T[ND[i].a].prod, adf = ND[i].fused_add;
                if (y.consumers.size() != 2 || !((y.consumers[0] == c1f && y.consumers[1] == adf) || (y.consumers[0] == adf && y.consumers[1] == c1f))) return;
                if (y.prod < 0 || ND[y.prod].kind != N_ADD || ND[y.prod].fused_into < 0) return;
                const Node& ad = ND[y.prod];
                const int hi = ad.fused_into;
                Node& h = ND[hi];
                if (h.kind != N_CONV || h.fused_add != y.prod || h.bchain_into >= 0 || h.fb_a >= 0 || h.fbd_a >= 0 || h.dual >= 0 || h.dual_host >= 0 ||
                    h.absorbed_by >= 0 || h.cd.groups != 1 || h.cd.kernel != 1 || h.cd.stride != 2 || h.cd.pad != 0 || h.cd.relu || !h.cd.quant_input) return;
                const int other = (ad.a == h.out) ? ad.b : ad.a;
                if (T[other].consumers.size() != 1 || other == net->out_t) return;
                const int gi = T[other].prod;
                Node& g = ND[gi];
                if (g.kind != N_CONV || g.fused_add >= 0 || g.absorbed_by >= 0 || g.dual >= 0 || g.dual_host >= 0 || g.cd.groups != 1 || g.cd.kernel != 3 ||
                    g.cd.stride != 1 || g.cd.pad != 1 || g.cd.relu || !g.cd.quant_input) return;
                if (T[g.a].consumers.size() != 1 || g.a == net->out_t) return;
                const int bi = T[g.a].prod;
                Node& b0 = ND[bi];
                if (b0.kind != N_CONV || b0.fused_add >= 0 || b0.absorbed_by >= 0 || b0.dual >= 0 || b0.dual_host >= 0 || b0.cd.groups != 1 || b0.cd.kernel != 3 ||
                    b0.cd.stride != 2 || b0.cd.pad != 1 || !b0.cd.quant_input || b0.a != h.a) return;
                const Tensor& x = T[h.a];
                if (ND[x.prod].kind == N_INPUT || x.H != 2 * y.H || x.W != 2 * y.W) return;
                if (b0.cd.cin * 2 != C || h.cd.cin * 2 != C || b0.cd.cout != C || g.cd.cin != C || g.cd.cout != C || h.cd.cout != C) return;
                int na = 0, nh = 0;
                if (consumer_format(x, b0.cd, &na, "finalize") || consumer_format(x, h.cd, &nh, "finalize")) return;
                if (!bchain_ds_supported(C, y.H, y.W)) return;
                h.bds_a = bi; h.bds_b = gi; h.bchain_into = lastn;
                b0.absorbed_by = hi; g.absorbed_by = hi; b0.no_classes = true; g.no_classes = true;
                hosts.insert(hosts.begin(), hi);
            }();
            ND[lastn].bchain = hosts;
        }
    }
}

// pass 1e
static void plan_inverted_residuals(f8_net* net, int max_batch) {
    auto& T = net->tensors;
    auto& ND = net->nodes;
    const int nn = (int)ND.size();
    const Options& opt = net->opt;
    const int fuse_blocks = opt.fuse_blocks;
    (void)T; (void)nn; (void)opt; (void)fuse_blocks; (void)max_batch;
    // ---- 1e. MobileNet-V2 inverted residual: 1x1 expand (ReLU) -> depthwise 3x3 (ReLU) -> 1x1 project [+ residual with the
    //          block input], intermediates read by nobody else  ->  one launch (f8_ir.hip), the expanded tensors stay in LDS
    for (int i = 0; opt.fuse_ir && i < nn; ++i) {
        Node& c = ND[i];
        if (c.kind != N_CONV || c.cd.groups != 1 || c.cd.kernel != 1 || c.cd.stride != 1 || c.cd.pad != 0 || !c.cd.quant_input ||
            c.absorbed_by >= 0 || c.fb_a >= 0 || c.dual >= 0 || c.dual_host >= 0) continue;
        const Tensor& tb = T[c.a];
        if (tb.consumers.size() != 1 || c.a == net->out_t) continue;
        Node& b = ND[tb.prod];
        if (b.kind != N_CONV || b.cd.groups == 1 || b.cd.groups != b.cd.cin || b.cd.kernel != 3 || b.cd.pad != 1 || (b.cd.stride != 1 && b.cd.stride != 2) ||
            !b.cd.quant_input || b.fused_add >= 0 || b.absorbed_by >= 0) continue;
        const Tensor& ta = T[b.a];
        if (ta.consumers.size() != 1 || b.a == net->out_t) continue;
        Node& a0 = ND[ta.prod];
        if (a0.kind != N_CONV || a0.cd.groups != 1 || a0.cd.kernel != 1 || a0.cd.stride != 1 || a0.cd.pad != 0 || a0.fused_add >= 0 ||
            a0.absorbed_by >= 0 || a0.dual >= 0 || a0.dual_host >= 0 || !a0.cd.quant_input) continue;
        if (a0.ir_a >= 0 || a0.fb_a >= 0 || a0.fbd_a >= 0) continue;   // already the host of another fused launch (chains of 1x1 / dw / 1x1 / dw ...)
        if (ND[T[a0.a].prod].kind == N_INPUT) continue;          // the network input has its own layouts
        if (c.fused_add >= 0) {                                   // a residual join must be with the block input
            const Node& ad = ND[c.fused_add];
            const int other = (ad.a == c.out) ? ad.b : ad.a;
            if (other != a0.a || b.cd.stride != 1) continue;
        }
        const Tensor& x = T[a0.a];
        int R = 0, G = 0;
        if (!fused_ir_config(x.Cs, round_up(c.cd.cout, 32), x.H, x.W, b.cd.stride, &R, &G)) continue;
        if (opt.fuse_ir == 1) {
            // Where the fused launch wins (measured per block on MobileNet-V2 at 128 images, round 3 — after the depthwise phase moved to
            // the matrix cores and the requantisations to five operations; three launches / fused, us): stage 1 (112x112 / 2 and 56x56
            // maps: the 6x expanded tensor is 154 MB) 139.5 / 118.6 and 126.3 / 108.0; stage 2 (56x56 / 2, 28x28): 138 / 50.6, 69 / 40.4,
            // 68 / 38.7; stages 3 - 4 (28x28 / 2, 14x14): 44 / 30.5, 48 / 34.4, 46 / 33.3, 45 / 30.5, 51 / 37.4 and, with 576 expanded
            // channels, 49.1 / 46.3 and 46.1 / 43.3.  The 7x7 blocks (960 channels = 15 chunks over 49 pixels, 64 workgroups for 128
            // images) stay three launches: 34 - 40 / 90 - 114.  fuse_ir = 2 fuses every block that has an instance.
            if (T[c.out].H * T[c.out].W < 128) continue;
        }
        a0.absorbed_by = i; b.absorbed_by = i;
        c.ir_a = ta.prod; c.ir_b = tb.prod; c.ir_R = R; c.ir_G = G;
    }

}

// pass 1h
static void plan_mobilenet_v2_head(f8_net* net, int max_batch) {
    auto& T = net->tensors;
    auto& ND = net->nodes;
    const int nn = (int)ND.size();
    const Options& opt = net->opt;
    const int fuse_blocks = opt.fuse_blocks;
    (void)T; (void)nn; (void)opt; (void)fuse_blocks; (void)max_batch;
    // ---- 1h. MobileNet-V2 head: network input -> 3x3 / 2 conv (cin <= 4 -> 32, ReLU) -> depthwise 3x3 (ReLU) -> 1x1 (32 -> <= 32), each read
    //          by nobody else  ->  ONE row-walking launch (f8_stem.hip, stem_rows_kernel<KIND, true>): the three convs hand their rows to
    //          each other in registers, the launch reads the caller's buffer itself
    for (int i = 0; opt.fuse_head2 && opt.fuse_stem && i < nn; ++i) {
        Node& c = ND[i];
        if (c.kind != N_CONV || c.cd.groups != 1 || c.cd.kernel != 1 || c.cd.stride != 1 || c.cd.pad != 0 || c.cd.relu || !c.cd.quant_input ||
            c.cd.input_signed || c.fused_add >= 0 || c.absorbed_by >= 0 || c.ir_a >= 0 || c.dual >= 0 || c.dual_host >= 0 || c.out == net->out_t) continue;
        if (c.cd.cin != 32 || round_up(c.cd.cout, 32) != 32) continue;
        const Tensor& tb = T[c.a];
        if (tb.consumers.size() != 1 || c.a == net->out_t) continue;
        Node& b = ND[tb.prod];
        if (b.kind != N_CONV || b.cd.groups != b.cd.cin || b.cd.cin != 32 || b.cd.cout != 32 || b.cd.kernel != 3 || b.cd.stride != 1 || b.cd.pad != 1 ||
            !b.cd.relu || !b.cd.quant_input || b.cd.input_signed || b.fused_add >= 0 || b.absorbed_by >= 0) continue;
        const Tensor& ta = T[b.a];
        if (ta.consumers.size() != 1 || b.a == net->out_t) continue;
        Node& h = ND[ta.prod];
        if (h.kind != N_CONV || h.cd.groups != 1 || h.cd.kernel != 3 || h.cd.stride != 2 || h.cd.pad != 1 || h.cd.cin > 4 || h.cd.cout != 32 || !h.cd.relu ||
            h.fused_add >= 0 || h.absorbed_by >= 0 || h.a == net->out_t) continue;
        const Tensor& x = T[h.a];
        if (x.prod < 0 || ND[x.prod].kind != N_INPUT || x.consumers.size() != 1 || !(!h.cd.quant_input || x.fl == h.cd.input_fl)) continue;
        if (!head2_supported(x.H, x.W) || ta.H * 2 != x.H || ta.W * 2 != x.W) continue;
        int na = 0, nb2 = 0;
        if (consumer_format(ta, b.cd, &na, "finalize") || consumer_format(tb, c.cd, &nb2, "finalize") || na < 1 || nb2 < 1 || na > 16 || nb2 > 16 ||
            false) continue;   // 16: kRequantU8MaxShift (the launch's float-converter form; accumulators the planner cannot bound take its integer form: rq_int)
        bool int8_readers = !T[c.out].consumers.empty() && T[c.out].consumers.size() <= 2;
        for (int u : T[c.out].consumers) if (ND[u].kind != N_CONV || !ND[u].cd.quant_input) int8_readers = false;
        if (!int8_readers) continue;
        h.absorbed_by = i; b.absorbed_by = i;
        c.h2_head = ta.prod; c.h2_dw = tb.prod;
    }

}

// pass 1i
static void plan_last_conv_and_pool(f8_net* net, int max_batch) {
    auto& T = net->tensors;
    auto& ND = net->nodes;
    const int nn = (int)ND.size();
    const Options& opt = net->opt;
    const int fuse_blocks = opt.fuse_blocks;
    (void)T; (void)nn; (void)opt; (void)fuse_blocks; (void)max_batch;
    // ---- 1i. the network's last 1x1 conv (hosting its residual join, or a plain conv + ReLU) and the average pool behind it: ONE launch that sums the map's
    //          pixels in the epilogue (f8_pool.hip); the conv's int32 result — read by nobody but the pool — never exists
    for (int i = 0; opt.fuse_pool && i < nn; ++i) {
        Node& p = ND[i];
        if (p.kind != N_AVGPOOL || p.a == net->out_t || T[p.a].consumers.size() != 1) continue;
        const Tensor& t = T[p.a];
        if (t.prod < 0) continue;
        int ci = t.prod;
        if (ND[ci].kind == N_ADD) { if (ND[ci].fused_into < 0) continue; ci = ND[ci].fused_into; }
        Node& c = ND[ci];
        if (c.kind == N_CONV && c.chain_into == ci && c.fused_add >= 0 && ND[c.fused_add].out == p.a && p.out != net->out_t &&
            cchain_supported(c.cd.cout, c.cd.cin, t.H, t.W, c.cd.cout, false)) {      // (geometry: a chain's last block is an identity block)
            c.pool = i; p.pool_host = ci;                         // the last block of a 7x7 cluster chain: the pool is summed from its stream registers (f8_cchain.hip)
            continue;
        }
        if (c.kind != N_CONV || c.absorbed_by >= 0 || c.dual >= 0 || c.dual_host >= 0 || c.fb_a >= 0 || c.fbd_a >= 0 || c.chain_into >= 0 || c.bchain_into >= 0 ||
            c.ir_a >= 0 || c.p12_a >= 0 || c.bb_a >= 0 || c.h2_head >= 0 || c.cd.groups != 1 || c.cd.kernel != 1 || c.cd.stride != 1 || c.cd.pad != 0 || !c.cd.quant_input) continue;
        if ((c.fused_add >= 0 ? ND[c.fused_add].out : c.out) != p.a) continue;
        if (c.fused_add < 0 && T[c.out].consumers.size() != 1) continue;
        const int ckp = round_up(c.cd.cin, 32), coutP = round_up(c.cd.cout, 32);
        if (!conv1x1_pool_supported(ckp, coutP, t.H * t.W)) continue;
        c.pool = i; p.pool_host = ci;
    }

}

// pass 2
static void plan_tensor_forms(f8_net* net, int max_batch) {
    auto& T = net->tensors;
    auto& ND = net->nodes;
    const int nn = (int)ND.size();
    const Options& opt = net->opt;
    const int fuse_blocks = opt.fuse_blocks;
    (void)T; (void)nn; (void)opt; (void)fuse_blocks; (void)max_batch;
    // ---- 2. which forms does each tensor need?  (reverse order: consumers before producers)
    {
        Tensor& O = T[net->out_t];
        const Node& p = ND[O.prod];
        // the classifier (a linear / 1x1 conv on a 1x1 map that nothing else reads) writes the caller's buffer itself: no int32 form
        const Tensor& ps = T[p.a >= 0 ? p.a : net->out_t];
        O.dense_out = opt.fuse_fc && (p.kind == N_LINEAR || p.kind == N_CONV) && p.cd.kernel == 1 && p.cd.stride == 1 && p.cd.pad == 0 && p.cd.groups == 1 &&
                      O.H == 1 && O.W == 1 && ps.H == 1 && ps.W == 1 && O.consumers.empty() && p.fused_add < 0 && p.absorbed_by < 0 && !p.cd.relu &&
                      fc_dense_supported(ps.Cs, round_up(p.cd.cout, 32));
        if (!O.dense_out) add_form(O, FORM_I32, 0, 0);
    }
    for (int i = nn - 1; i >= 0; --i) {
        Node& nd = ND[i];
        switch (nd.kind) {
            case N_CONV: case N_LINEAR: {
                Tensor& s = T[nd.a];
                int n = 0;
                consumer_format(s, nd.cd, &n, "finalize");
                nd.depthwise = nd.cd.groups != 1;
                if ((nd.absorbed_by >= 0 && ND[nd.absorbed_by].fbd_b == i) || (nd.dual_host >= 0 && ND[nd.dual_host].fbd_a >= 0)) break;   // DS: in LDS
                if (nd.p12_a >= 0) break;                    // the 1x1's output lives in LDS inside the launch
                if (nd.h2_head >= 0 || (nd.absorbed_by >= 0 && ND[nd.absorbed_by].h2_dw == i)) break;   // 1x1 / depthwise of the MobileNet-V2 head launch: rows in registers
                if (nd.absorbed_by >= 0 && ND[nd.absorbed_by].bds_b == i) break;   // second 3x3 of the opening block of a bchain launch: `mid` lives in LDS
                                                                 // (its 3x3 / 2 and its shortcut conv each ask for their int8 form of the block input below)
                if (nd.bb_a >= 0) {                          // second conv of a chained BasicBlock: its source (`mid`) lives in LDS
                    if (bchain_pos(net, i) == 0) {
                        const Node& ad = ND[nd.fused_add];
                        add_form(T[(ad.a == nd.out) ? ad.b : ad.a], FORM_I32, 0, 0);      // the stage's int32 stream: the chain's only input form
                    }
                    break;
                }
                if (nd.absorbed_by >= 0 && ND[nd.absorbed_by].bb_a == i) break;            // first conv of a chained BasicBlock: its int8 input is made in the launch
                {   // stage chain (f8_chain.hip): the tensors between its blocks exist in no form at all; an identity first block
                    // reads only the int32 form of the stage input (its int8 copy is made in the launch)
                    const int host = nd.absorbed_by >= 0 && ND[nd.absorbed_by].fb_a == i ? nd.absorbed_by : -1;   // nd = body.0 of an identity block
                    const int pos = chain_pos(net, host);
                    if (pos > 0) break;
                    if (pos == 0) { add_form(s, FORM_I32, 0, 0); break; }
                }
                if (nd.fb_a >= 0 || (nd.absorbed_by >= 0 && ND[nd.absorbed_by].fb_b == i) || nd.ir_a >= 0 ||
                    (nd.absorbed_by >= 0 && ND[nd.absorbed_by].ir_b == i)) {
                    // source lives in LDS inside the fused launch: no HBM form.  (The block's first conv
                    // still reads the block input from HBM and falls through to the generic case.)
                    if (nd.fused_add >= 0 && chain_pos(net, i) <= 0) {
                        const Node& ad = ND[nd.fused_add];
                        const int other = (ad.a == nd.out) ? ad.b : ad.a;
                        add_form(T[other], FORM_I32, 0, 0);
                    }
                    break;
                }
                nd.stem = !nd.depthwise && nd.cd.cin <= 4 && ND[s.prod].kind == N_INPUT && nd.cd.kernel <= 8 &&
                          s.consumers.size() == 1 && nd.a != net->out_t && n == 0;
                if (nd.stem) {
                    const int P = T[nd.out].H, Q = T[nd.out].W;
                    (void)P;
                    const int f = add_form(s, FORM_STEM, 0, 0);
                    Form& F = s.forms[f];
                    F.sgn = nd.cd.input_signed ? 1 : 0;
                    const bool h2 = nd.absorbed_by >= 0 && ND[nd.absorbed_by].h2_head == i;
                    F.pad = nd.cd.pad + (nd.sp_pool >= 0 ? 2 : 0) + (h2 ? 4 : 0);      // fused stem + pool: 2 more halo pixels keep every tile's patch in memory (head launch of MobileNet-V2: 5 in all)
                    F.Hp = s.H + 2 * F.pad;
                    F.Wp = round_up(std::max(s.W + 2 * F.pad, nd.cd.stride * (Q - 1) + 8), (nd.sp_pool >= 0 || h2) ? 4 : 2);
                } else {
                    add_form(s, FORM_I8, n, nd.cd.input_signed ? 1 : 0);
                }
                if (nd.fused_add >= 0 && nd.dual < 0 && nd.bds_a < 0) {
                    const Node& ad = ND[nd.fused_add];
                    const int other = (ad.a == nd.out) ? ad.b : ad.a;
                    add_form(T[other], FORM_I32, 0, 0);
                }
                break;
            }
            case N_ADD:
                if (nd.fused_into < 0) { add_form(T[nd.a], FORM_I32, 0, 0); add_form(T[nd.b], FORM_I32, 0, 0); }
                break;
            case N_MAXPOOL: {
                Tensor& o = T[nd.out];
                if (!o.forms.empty() && T[nd.a].consumers.size() == 1 && nd.a != net->out_t) {
                    // ResNet head: 7x7/2 stem conv + this pool in one launch (the conv output never leaves LDS)
                    Node& c = ND[T[nd.a].prod];
                    if (c.kind == N_CONV && c.cd.groups == 1 && c.fused_add < 0 && ND[T[c.a].prod].kind == N_INPUT && T[c.a].consumers.size() == 1 &&
                        c.a != net->out_t && (!c.cd.quant_input || T[c.a].fl == c.cd.input_fl) && opt.fuse_stem &&
                        stem_pool_supported(c.cd.cin, c.cd.cout, c.cd.kernel, c.cd.stride, c.cd.pad, nd.pk, nd.pstride, nd.ppad, o.H, o.W, opt.stem_rows, T[c.a].H, T[c.a].W)) {
                        c.sp_pool = i; nd.sp_conv = T[nd.a].prod;
                        break;                                   // no HBM form of the conv output
                    }
                }
                if (o.forms.size() == 1 && o.forms[0].kind == FORM_I8)
                    add_form(T[nd.a], FORM_I8, o.forms[0].n, o.forms[0].sgn);
                else
                    add_form(T[nd.a], FORM_I32, 0, 0);
                break;
            }
            case N_AVGPOOL:
                if (nd.pool_host < 0) add_form(T[nd.a], FORM_I32, 0, 0);      // fused behind its conv (1i): the pool's input has no form
                break;
            default: break;
        }
    }

}

// pass 3
static int emit_steps(f8_net* net, int max_batch) {
    auto& T = net->tensors;
    auto& ND = net->nodes;
    const int nn = (int)ND.size();
    const Options& opt = net->opt;
    const int fuse_blocks = opt.fuse_blocks;
    (void)T; (void)nn; (void)opt; (void)fuse_blocks; (void)max_batch;
    // ---- 3. emit steps in node order
    net->steps.clear();
    auto emit_requants = [&](int t, const std::vector<int>& extra) {
        for (int f : extra) {
            Step st; st.kind = S_REQUANT; st.node = T[t].prod;
            st.src_t = t; st.src_f = find_form(T[t], FORM_I32, 0, 0);
            st.out.t = t; st.out.f8[0] = f;
            st.name = "requant:" + tname(net, t); st.kernel = "f8::add_kernel";
            const double e = (double)T[t].H * T[t].W * T[t].Cs;
            st.bytes_per_img = e * 5;
            net->steps.push_back(st);
        }
    };
    for (int i = 0; i < nn; ++i) {
        Node& nd = ND[i];
        if (nd.kind == N_ADD && nd.fused_into >= 0) continue;
        if (nd.kind == N_CONV && (nd.absorbed_by >= 0 || nd.dual_host >= 0)) continue;
        if (nd.kind == N_CONV && nd.chain_into >= 0 && nd.chain_into != i) continue;      // runs inside the chain launch of a later block
        if (nd.kind == N_CONV && nd.bchain_into >= 0 && nd.bchain_into != i) continue;
        if (nd.kind == N_MAXPOOL && nd.sp_conv >= 0) continue;
        if (nd.kind == N_AVGPOOL && nd.pool_host >= 0) continue;     // runs in its conv's launch
        Step st; st.node = i;
        std::vector<int> extra;
        int out_t = nd.out;
        switch (nd.kind) {
            case N_INPUT: {
                st.kind = S_INPUT;
                Tensor& o = T[nd.out];
                if (o.forms.empty()) add_form(o, FORM_I32, 0, 0);
                // int8 formats that need a real shift (quant_input=1 on the network input) go through the
                // int32 form + a stand-alone requant; the n == 0 format is a plain narrowing in the input kernel
                for (size_t f = 0; f < o.forms.size(); ++f)
                    if (o.forms[f].kind == FORM_I8 && o.forms[f].n != 0) extra.push_back((int)f);
                if (!extra.empty()) add_form(o, FORM_I32, 0, 0);
                st.out.t = nd.out;
                st.name = "input"; st.kernel = "f8::input_kernel";     // refined below once the forms are known
                double b = (double)o.C * o.H * o.W * 4;
                for (auto& F : o.forms) b += (double)o.H * o.W * (F.kind == FORM_I32 ? o.Cs * 4 : (F.kind == FORM_STEM ? 4 : o.Cs));
                st.bytes_per_img = b;
                break;
            }
            case N_CONV: case N_LINEAR: {
                if (nd.h2_head >= 0) {
                    // ---- MobileNet-V2 head: 3x3 / 2 conv + depthwise 3x3 + this 1x1 in one launch
                    Node& hh = ND[nd.h2_head]; Node& hb = ND[nd.h2_dw];
                    Tensor& s = T[hh.a];
                    st.kind = S_HEAD2;
                    st.src_t = hh.a; st.src_f = find_form(s, FORM_STEM, 0, 0);
                    pack_conv_weights(net, hh, s, T[hh.out]);
                    pack_dw_weights(net, hb, T[hb.a]);
                    pack_conv_weights(net, nd, T[nd.a], T[nd.out]);
                    select_outputs(net, out_t, &st.out, &extra);
                    Tensor& o = T[out_t];
                    const double cpx = (double)o.H * o.W;
                    st.ops_per_img = 2.0 * cpx * (32.0 * hh.cd.cin * 9 + 32.0 * 9 + 32.0 * nd.cd.cout);
                    st.valu_per_img = 3.0 * cpx * (32.0 + 32.0 + o.Cs * out_forms8(st));
                    st.bytes_per_img = (double)s.H * s.W * 4 + cpx * o.Cs * ((st.out.f8[0] >= 0) + (st.out.f8[1] >= 0));
                    st.bytes_const = 32.0 * 100 + 32.0 * 13 + 32.0 * 36;
                    st.name = "head3x3s2+dw3x3+1x1:" + tname(net, hh.out) + "+" + tname(net, hb.out) + "+" + tname(net, nd.out);
                    st.kernel = "f8::stem_rows_kernel";
                    break;
                }
                if (nd.sp_pool >= 0 && nd.stem) {
                    // ---- ResNet head: stem conv + max-pool in one launch
                    const Node& pl = ND[nd.sp_pool];
                    Tensor& s = T[nd.a]; Tensor& o = T[pl.out];
                    st.kind = S_STEMPOOL;
                    st.src_t = nd.a; st.src_f = find_form(s, FORM_STEM, 0, 0);
                    st.relu0 = nd.cd.relu;
                    pack_conv_weights(net, nd, s, T[nd.out]);
                    out_t = pl.out;
                    select_outputs(net, out_t, &st.out, &extra);
                    const f8_conv_desc& d = nd.cd;
                    const double cpx = (double)T[nd.out].H * T[nd.out].W;
                    st.ops_per_img = 2.0 * cpx * d.cout * d.cin * d.kernel * d.kernel;
                    st.valu_per_img = cpx * d.cout + 3.0 * (double)o.H * o.W * o.Cs * out_forms8(st);      // one max per conv value; the requantisation runs on the pooled ones
                    st.bytes_per_img = (double)s.H * s.W * 4 + (double)o.H * o.W * o.Cs * ((st.out.f32 >= 0 ? 4 : 0) + (st.out.f8[0] >= 0) + (st.out.f8[1] >= 0));
                    st.bytes_const = (double)nd.coutP * (nd.ktot + 4);
                    st.name = "stem7x7s2+maxpool3x3s2:" + tname(net, nd.out) + "+" + tname(net, pl.out);
                    st.kernel = (opt.stem_rows && T[pl.out].W >= 2 && T[pl.out].W <= 56 && s.W == 4 * T[pl.out].W && s.H == 4 * T[pl.out].H) ? "f8::stem_rows_kernel" : "f8::stem_pool_kernel";    // keep in sync with launch_stem_pool
                    break;
                }
                if (nd.bchain_into == i) {
                    // ---- BasicBlock chain: nd is the second conv of its LAST block
                    const std::vector<int> ch = nd.bchain;
                    Node& f2 = ND[ch[0]];
                    const bool ds = f2.bds_a >= 0;               // the chain starts with the stage-opening block: f2 is its shortcut conv
                    Node& f1 = ND[ds ? f2.bds_a : f2.bb_a];
                    Tensor& x = T[f1.a];
                    st.kind = S_BCHAIN;
                    st.src_t = f1.a;
                    if (ds) {
                        int n0 = 0; consumer_format(x, f1.cd, &n0, "finalize"); st.src_f = find_form(x, FORM_I8, n0, f1.cd.input_signed ? 1 : 0);
                        int n1 = 0; consumer_format(x, f2.cd, &n1, "finalize"); st.res_t = f1.a; st.res_f = find_form(x, FORM_I8, n1, f2.cd.input_signed ? 1 : 0);
                    } else st.src_f = find_form(x, FORM_I32, 0, 0);
                    double ops = 0, wbytes = 0;
                    for (int hi : ch) {
                        const bool hds = ND[hi].bds_a >= 0;
                        Node* cv[3] = {&ND[hds ? ND[hi].bds_a : ND[hi].bb_a], hds ? &ND[ND[hi].bds_b] : &ND[hi], hds ? &ND[hi] : nullptr};
                        for (Node* c : cv) {
                            if (!c) continue;
                            pack_conv_weights(net, *c, T[c->a], T[c->out]);
                            pack_frag_weights(net, *c);
                            ops += 2.0 * T[c->out].H * T[c->out].W * (double)c->cd.kernel * c->cd.kernel * c->cd.cin * c->cd.cout;
                            wbytes += (double)c->coutP * (c->ktot + 4);
                        }
                    }
                    out_t = ND[nd.fused_add].out;
                    select_outputs(net, out_t, &st.out, &extra);
                    Tensor& o = T[out_t];
                    const double px = (double)o.H * o.W;
                    double b = ds ? (double)x.H * x.W * x.Cs + (st.res_f != st.src_f ? px * x.Cs : 0) : px * x.Cs * 4;
                    if (st.out.f32 >= 0) b += px * o.Cs * 4;
                    for (int k = 0; k < 2; ++k) if (st.out.f8[k] >= 0) b += px * o.Cs;
                    st.ops_per_img = ops; st.bytes_per_img = b; st.bytes_const = wbytes;
                    st.valu_per_img = (double)ch.size() * px * o.C * (3.0 + 2.0 + 3.0) - 3.0 * px * o.C + 3.0 * px * o.Cs * out_forms8(st);   // per block: first conv's int8, join, the next block's int8 input; the last block's are the output forms
                    st.name = "basic_chain_x" + std::to_string(ch.size()) + (ds ? "_ds:" : ":") + tname(net, f1.out) + ".." + tname(net, nd.out);
                    char kb[160];
                    // the instance's arithmetic as launch_bchain picks it (bchain_fast): the float converter only for values the planner bounds
                    bool bounded = true;
                    for (size_t k = 0; k < ch.size(); ++k) {
                        const Node& hk = ND[ch[k]];
                        const Node& c1 = ND[hk.bds_a >= 0 ? hk.bds_a : hk.bb_a];
                        bounded = bounded && conv_acc_bounded(c1) && stream_bounded(net, ND[hk.fused_add].out) && (k > 0 || hk.bds_a >= 0 || stream_bounded(net, c1.a));
                    }
                    bchain_kernel_name(kb, sizeof kb, o.C, o.H, o.W, ds, (opt.requant_float && bounded) ? 1 : 2);   // the name comes from f8_bchain.hip, next to the launcher
                    st.kernel = kb;
                    break;
                }
                if (nd.chain_into == i) {
                    // ---- stage chain: nd is the host conv of its LAST block
                    const std::vector<int> ch = nd.chain;
                    Node& hf = ND[ch[0]];
                    const bool tail = hf.tail;                  // first block = the join of a stride-2 opening block (its body.0 + body.2: the S_P12 step in front)
                    const bool ds = hf.fbd_a >= 0 || tail;
                    Tensor& x = T[ds ? hf.a : ND[hf.fb_a].a];
                    st.kind = S_CHAIN;
                    st.src_t = ds ? hf.a : ND[hf.fb_a].a;
                    if (tail) {
                        int n0 = 0; consumer_format(x, hf.cd, &n0, "finalize"); st.src_f = find_form(x, FORM_I8, n0, hf.cd.input_signed ? 1 : 0);      // the shortcut's int8 form
                        const Node& g0 = ND[hf.dual];
                        int n2 = 0; consumer_format(T[g0.a], g0.cd, &n2, "finalize");
                        st.src2_t = g0.a; st.src2_f = find_form(T[g0.a], FORM_I8, n2, g0.cd.input_signed ? 1 : 0);                                      // mid2 in body.4's format
                    }
                    else if (ds) { int n0 = 0; consumer_format(x, ND[hf.fbd_a].cd, &n0, "finalize"); st.src_f = find_form(x, FORM_I8, n0, ND[hf.fbd_a].cd.input_signed ? 1 : 0); }
                    else st.src_f = find_form(x, FORM_I32, 0, 0);
                    double ops = 0, wbytes = 0;
                    for (int hi : ch) {
                        Node& hh = ND[hi];
                        const bool hds = hh.fbd_a >= 0, htl = hh.tail;
                        Node* cv[4] = {htl ? nullptr : &ND[hds ? hh.fbd_a : hh.fb_a], htl ? nullptr : &ND[hds ? hh.fbd_b : hh.fb_b], (hds || htl) ? &ND[hh.dual] : &hh, (hds || htl) ? &hh : nullptr};
                        for (Node* c : cv) {
                            if (!c) continue;
                            pack_conv_weights(net, *c, T[c->a], T[c->out]);
                            pack_frag_weights(net, *c);
                            const double px = (double)T[c->out].H * T[c->out].W;
                            ops += 2.0 * px * c->cd.cin * c->cd.cout * c->cd.kernel * c->cd.kernel;
                            wbytes += (double)c->coutP * (c->ktot + 4);
                        }
                    }
                    out_t = ND[nd.fused_add].out;
                    Tensor& o = T[out_t];                        // the stage's output map (geometry of the chain)
                    if (nd.pool >= 0) {                          // ... summed over its pixels in the launch (1i): the step's outputs are the POOLED tensor's forms
                        out_t = ND[nd.pool].out;
                        if (T[out_t].forms.empty()) add_form(T[out_t], FORM_I32, 0, 0);
                    }
                    select_outputs(net, out_t, &st.out, &extra);
                    const double px = (double)o.H * o.W, opx = nd.pool >= 0 ? 1.0 : px;
                    double b = tail ? px * (x.Cs + T[st.src2_t].Cs) : px * x.Cs * (ds ? 1 : 4);   // the stage input, once (tail: the shortcut's pixels + mid2)
                    if (st.out.f32 >= 0) b += opx * o.Cs * 4;
                    for (int k = 0; k < 2; ++k) if (st.out.f8[k] >= 0) b += opx * o.Cs;
                    st.ops_per_img = ops; st.bytes_per_img = b; st.bytes_const = wbytes;
                    {   // per block: body.0's and body.2's int8 outputs (the TAIL join has none), the join, the next block's int8 input (last block: the output forms)
                        const double mid = (double)(tail ? ND[hf.dual].cd.cin : ND[ds ? hf.fbd_a : hf.fb_a].cd.cout);
                        st.valu_per_img = px * ((double)(ch.size() - (tail ? 1 : 0)) * 2.0 * mid * 3.0 + (double)ch.size() * o.C * 2.0 + (double)(ch.size() - 1) * o.C * 3.0) +
                                          opx * 3.0 * o.Cs * out_forms8(st) + (nd.pool >= 0 ? px * o.C : 0.0);
                    }
                    const Node& a0 = tail ? ND[hf.dual] : ND[ds ? hf.fbd_a : hf.fb_a];
                    st.name = "stage_chain_x" + std::to_string(ch.size()) + (tail ? "_tail" : (ds ? "_ds" : "")) + (nd.pool >= 0 ? "+avgpool:" : ":") + tname(net, a0.out) + ".." + tname(net, nd.out);
                    char kb[160];
                    const int C = o.C, MID = tail ? a0.cd.cin : a0.cd.cout;
                    int cR = 4, cW = 1;
                    chain_shape(C, MID, o.H, o.W, tail ? hf.cd.cin : a0.cd.cin, tail, &cR, &cW);
                    // the instance's arithmetic as launch_chain picks it (chain_fast): the float converter only for values the planner bounds
                    bool bounded = true;
                    for (size_t k = 0; k < ch.size(); ++k) {
                        const Node& hh = ND[ch[k]];
                        bounded = bounded && stream_bounded(net, ND[hh.fused_add].out);
                        if (hh.tail) continue;
                        const bool hds = hh.fbd_a >= 0;
                        const Node& na = ND[hds ? hh.fbd_a : hh.fb_a]; const Node& nb = ND[hds ? hh.fbd_b : hh.fb_b];
                        bounded = bounded && conv_acc_bounded(na) && conv_acc_bounded(nb) && (k > 0 || hds || stream_bounded(net, na.a));
                    }
                    chain_kernel_name(kb, sizeof kb, C, MID, o.H, o.W, tail ? hf.cd.cin : a0.cd.cin, tail, (opt.requant_float && bounded) ? 1 : 2);   // the name comes from f8_chain.hip, next to the launcher
                    st.kernel = kb;
                    break;
                }
                if (nd.fbd_a >= 0) {
                    // ---- fused stage-opening block (DS): nd is the shortcut conv, nd.dual the block's last body conv
                    Node& na = ND[nd.fbd_a]; Node& nb = ND[nd.fbd_b]; Node& ng = ND[nd.dual];
                    Tensor& x = T[nd.a];
                    st.kind = S_FUSED;
                    st.src_t = nd.a;
                    int n0 = 0; consumer_format(x, na.cd, &n0, "finalize");
                    st.src_f = find_form(x, FORM_I8, n0, na.cd.input_signed ? 1 : 0);
                    pack_conv_weights(net, na, x, T[na.out]);
                    pack_conv_weights(net, nb, T[nb.a], T[nb.out]);
                    pack_conv_weights(net, ng, T[ng.a], T[ng.out]);
                    pack_conv_weights(net, nd, x, T[nd.out]);
                    const Node& ad = ND[nd.fused_add];
                    const int dfl = T[nd.out].fl - T[ng.out].fl;          // > 0: body.4's result shifts left
                    st.acc_shl = dfl < 0 ? -dfl : 0; st.res_shl = dfl > 0 ? dfl : 0;
                    st.relu1 = ad.relu;
                    out_t = ad.out;
                    select_outputs(net, out_t, &st.out, &extra);
                    Tensor& o = T[out_t];
                    const double px = (double)x.H * x.W, pxo = (double)o.H * o.W;   // stride-2 opener: body.0 runs on the input map
                    st.ops_per_img = 2.0 * (px * (double)na.cd.cin * na.cd.cout + pxo * (9.0 * nb.cd.cin * nb.cd.cout + (double)ng.cd.cin * ng.cd.cout +
                                                                                         (double)nd.cd.cin * nd.cd.cout));
                    st.valu_per_img = 3.0 * (px * na.cd.cout + pxo * nb.cd.cout) + pxo * o.C * 2.0 + 3.0 * pxo * o.Cs * out_forms8(st);
                    double b = px * x.Cs;                                           // int8 input once
                    if (st.out.f32 >= 0) b += pxo * o.Cs * 4;
                    for (int k = 0; k < 2; ++k) if (st.out.f8[k] >= 0) b += pxo * o.Cs;
                    st.bytes_per_img = b;
                    st.bytes_const = (double)na.coutP * (na.ktot + 4) + (double)nb.coutP * (nb.ktot + 4) + (double)ng.coutP * (ng.ktot + 4) +
                                     (double)nd.coutP * (nd.ktot + 4);
                    st.name = std::string(nd.fbd_s2 ? "fused_opener_s2_R" : "fused_bottleneck_ds_R") + std::to_string(nd.fb_R) + ":" + tname(net, na.out) + "+" +
                              tname(net, nb.out) + "+" + tname(net, ng.out) + "+" + tname(net, nd.out);
                    char kb[160];
                    if (nd.fbd_s2) snprintf(kb, sizeof kb, "f8::fused_opener_kernel<%d, %d, %d, %d, %d, %s>", na.cd.cin, na.cd.cout, x.W, nd.fb_R, nd.cd.cout,
                                            (opt.opener_stg && st.out.f8[0] >= 0) ? "true" : "false");
                    else snprintf(kb, sizeof kb, "f8::fused_bottleneck_kernel<%d, %d, %d, %d, %d, true>", na.cd.cin, na.cd.cout, x.W, nd.fb_R, nd.cd.cout);
                    st.kernel = kb;
                    break;
                }
                if (nd.p12_a >= 0) {
                    // ---- 1x1 -> 3x3 of a 7x7 bottleneck block in one launch: nd is the 3x3
                    Node& na = ND[nd.p12_a];
                    Tensor& x = T[na.a];
                    st.kind = S_P12;
                    st.src_t = na.a;
                    int n0 = 0; consumer_format(x, na.cd, &n0, "finalize");
                    st.src_f = find_form(x, FORM_I8, n0, na.cd.input_signed ? 1 : 0);
                    pack_conv_weights(net, na, x, T[na.out]);
                    pack_conv_weights(net, nd, T[nd.a], T[nd.out]);
                    if (!nd.p12_s2) { pack_frag_weights(net, na); pack_frag_weights(net, nd); }      // f8_opener.hip streams the plain [cout][K] images
                    st.relu0 = nd.cd.relu;
                    select_outputs(net, out_t, &st.out, &extra);
                    if (st.out.f32 >= 0) return fail(F8_ERR_UNSUPPORTED, "finalize: the fused 1x1 -> 3x3 launch writes int8 forms only");
                    if (nd.p12_s2 && (st.out.f8[0] < 0 || st.out.f8[1] >= 0)) return fail(F8_ERR_UNSUPPORTED, "finalize: the stride-2 1x1 -> 3x3 launch writes exactly one int8 form");
                    Tensor& o = T[out_t];
                    const double px = (double)x.H * x.W, pxo = (double)o.H * o.W;
                    st.ops_per_img = 2.0 * (px * (double)na.cd.cin * na.cd.cout + pxo * 9.0 * nd.cd.cin * nd.cd.cout);
                    st.valu_per_img = 3.0 * (px * na.cd.cout + pxo * o.Cs * out_forms8(st));
                    double b = px * x.Cs;
                    for (int k = 0; k < 2; ++k) if (st.out.f8[k] >= 0) b += pxo * o.Cs;
                    st.bytes_per_img = b;
                    st.bytes_const = (double)na.coutP * (na.ktot + 4) + (double)nd.coutP * (nd.ktot + 4);
                    st.name = std::string(nd.p12_s2 ? "fused_opener_s2_p12_R" + std::to_string(nd.fb_R) + ":" : "fused_p12:") + tname(net, na.out) + "+" + tname(net, nd.out);
                    char kb[96];
                    if (nd.p12_s2) snprintf(kb, sizeof kb, "f8::fused_opener_kernel<%d, %d, %d, %d, %d, false, true, %d>", na.cd.cin, na.cd.cout, x.W, nd.fb_R, 4 * na.cd.cout, opt.requant_float ? 1 : 2);
                    else snprintf(kb, sizeof kb, "f8::fused_p12_kernel<%d, %d>", na.cd.cin, na.cd.cout);
                    st.kernel = kb;
                    break;
                }
                if (nd.ir_a >= 0) {
                    // ---- fused inverted residual: nd is the project conv
                    Node& na = ND[nd.ir_a]; Node& nb = ND[nd.ir_b];
                    Tensor& x = T[na.a];
                    st.kind = S_IR;
                    st.src_t = na.a;
                    int n0 = 0; consumer_format(x, na.cd, &n0, "finalize");
                    st.src_f = find_form(x, FORM_I8, n0, na.cd.input_signed ? 1 : 0);
                    pack_conv_weights(net, na, x, T[na.out]);
                    nb.depthwise = true;
                    pack_dw_weights(net, nb, T[nb.a]);
                    pack_conv_weights(net, nd, T[nd.a], T[nd.out]);
                    st.relu0 = nd.cd.relu;
                    if (nd.fused_add >= 0) {
                        const Node& ad = ND[nd.fused_add];
                        st.res_t = na.a; st.res_f = find_form(x, FORM_I32, 0, 0);
                        const int dfl = T[nd.out].fl - x.fl;
                        st.acc_shl = dfl < 0 ? -dfl : 0; st.res_shl = dfl > 0 ? dfl : 0;
                        st.relu1 = ad.relu;
                        out_t = ad.out;
                    }
                    select_outputs(net, out_t, &st.out, &extra);
                    Tensor& o = T[out_t];
                    const double px = (double)x.H * x.W, pxo = (double)o.H * o.W;
                    st.ops_per_img = 2.0 * (px * na.cd.cin * na.cd.cout + pxo * 9.0 * nb.cd.cout + pxo * (double)nd.cd.cin * nd.cd.cout);
                    st.valu_per_img = 3.0 * (px * na.cd.cout + pxo * nb.cd.cout) + (st.res_t >= 0 ? 2.0 * pxo * o.C : 0.0) + 3.0 * pxo * o.Cs * out_forms8(st);
                    double b = px * x.Cs + (st.res_t >= 0 ? px * x.Cs * 4 : 0);
                    if (st.out.f32 >= 0) b += pxo * o.Cs * 4;
                    for (int k = 0; k < 2; ++k) if (st.out.f8[k] >= 0) b += pxo * o.Cs;
                    st.bytes_per_img = b;
                    st.bytes_const = (double)na.coutP * (na.ktot + 4) + (double)T[nb.a].Cs * 13 + (double)nd.coutP * (nd.ktot + 4);
                    char kb[200];
                    snprintf(kb, sizeof kb, "fused_ir_s%d_%s:", nb.cd.stride, nd.ir_G > 1 ? ("G" + std::to_string(nd.ir_G)).c_str() : ("R" + std::to_string(nd.ir_R)).c_str());
                    st.name = std::string(kb) + tname(net, na.out) + "+" + tname(net, nb.out) + "+" + tname(net, nd.out);
                    {   // keep in sync with launch_fused_ir (FQ) and f8_ir.hip (P2MMA)
                        int n1 = 0, n2 = 0;
                        consumer_format(T[nb.a], nb.cd, &n1, "finalize"); consumer_format(T[nd.a], nd.cd, &n2, "finalize");
                        const bool fqf = na.cd.relu && nb.cd.relu && !nb.cd.input_signed && !nd.cd.input_signed && n1 > 0 && n2 > 0 && n1 <= 30 && n2 <= 30 && nd.coutP <= 96;
                        const int fq = !fqf ? 0 : ((opt.requant_float && n1 <= 16 && n2 <= 16 && conv_acc_bounded(na) && conv_acc_bounded(nb)) ? 1 : 2);
                        snprintf(kb, sizeof kb, "f8::fused_ir_kernel<%d, %d, %d, %s, %d>", x.Cs, nd.coutP, fq, nd.coutP <= 96 ? "true" : "false", nd.coutP <= 96 ? 8 : 4);
                    }
                    st.kernel = kb;
                    break;
                }
                if (nd.fb_a >= 0) {
                    // ---- fused bottleneck block: nd is its last conv
                    Node& na = ND[nd.fb_a]; Node& nb = ND[nd.fb_b];
                    Tensor& x = T[na.a];
                    st.kind = S_FUSED;
                    st.src_t = na.a;
                    int n0 = 0; consumer_format(x, na.cd, &n0, "finalize");
                    st.src_f = find_form(x, FORM_I8, n0, na.cd.input_signed ? 1 : 0);
                    pack_conv_weights(net, na, x, T[na.out]);
                    pack_conv_weights(net, nb, T[nb.a], T[nb.out]);
                    pack_conv_weights(net, nd, T[nd.a], T[nd.out]);
                    const Node& ad = ND[nd.fused_add];
                    st.res_t = na.a; st.res_f = find_form(x, FORM_I32, 0, 0);
                    const int dfl = T[nd.out].fl - x.fl;
                    st.acc_shl = dfl < 0 ? -dfl : 0; st.res_shl = dfl > 0 ? dfl : 0;
                    st.relu1 = ad.relu;
                    out_t = ad.out;
                    select_outputs(net, out_t, &st.out, &extra);
                    Tensor& o = T[out_t];
                    const double px = (double)x.H * x.W;
                    st.ops_per_img = 2.0 * px * ((double)na.cd.cin * na.cd.cout + 9.0 * nb.cd.cin * nb.cd.cout + (double)nd.cd.cin * nd.cd.cout);
                    st.valu_per_img = px * (3.0 * (na.cd.cout + nb.cd.cout) + 2.0 * o.C + 3.0 * o.Cs * out_forms8(st));
                    double b = px * x.Cs * (1 + 4);                                 // int8 input + int32 residual, once each
                    if (st.out.f32 >= 0) b += px * o.Cs * 4;
                    for (int k = 0; k < 2; ++k) if (st.out.f8[k] >= 0) b += px * o.Cs;
                    st.bytes_per_img = b;
                    st.bytes_const = (double)na.coutP * (na.ktot + 4) + (double)nb.coutP * (nb.ktot + 4) + (double)nd.coutP * (nd.ktot + 4);
                    st.name = "fused_bottleneck_R" + std::to_string(nd.fb_R) + ":" + tname(net, na.out) + "+" + tname(net, nb.out) + "+" + tname(net, nd.out);
                    char kb[160];
                    snprintf(kb, sizeof kb, "f8::fused_bottleneck_kernel<%d, %d, %d, %d, %d, false>", na.cd.cin, na.cd.cout, x.W, nd.fb_R, na.cd.cin);
                    st.kernel = kb;
                    break;
                }
                Tensor& s = T[nd.a];
                st.kind = nd.depthwise ? S_DW : S_CONV;
                st.src_t = nd.a;
                int n = 0; consumer_format(s, nd.cd, &n, "finalize");
                st.src_f = nd.stem ? find_form(s, FORM_STEM, 0, 0) : find_form(s, FORM_I8, n, nd.cd.input_signed ? 1 : 0);
                st.relu0 = nd.cd.relu;
                if (nd.depthwise) pack_dw_weights(net, nd, s);
                else {
                    pack_conv_weights(net, nd, s, T[nd.out]);
                    const int M1 = T[nd.out].H * T[nd.out].W;
                    if (!pick_conv_tile(M1 * max_batch, nd.coutP, nd.ck, nd.fused_add >= 0, opt.bk128 != 0, &nd.tile))
                        return fail(F8_ERR_UNSUPPORTED, "finalize: no conv kernel instance for ck=%d coutP=%d", nd.ck, nd.coutP);
                    if (nd.dual >= 0) {       // the dual-GEMM instances: 128x64 / 64x64, or 128x128 where many cout tiles re-read X
                        const int wide = opt.dual_wide;   // measured: 7x7x2048 join 67 -> 52 us; 14x14x1024 and 28x28x512 are slower with the wide tile
                        nd.tile.bk = 64;
                        if (nd.coutP >= wide && nd.tile.bm == 128) nd.tile.bn = 128; else nd.tile.bn = 64;
                    }
                    if (!nd.stem && nd.cd.groups == 1 && nd.cd.kernel == 3 && nd.cd.stride == 1 && nd.cd.pad == 1 && nd.ck == nd.cd.cin && opt.patch3x3 &&
                        !conv3x3_patch_config(nd.cd.cin, s.H, s.W, nd.coutP, &nd.p3_R, &nd.p3_imgs, &nd.p3_bn)) nd.p3_R = 0;
                }
                if (nd.fused_add >= 0) {
                    const Node& ad = ND[nd.fused_add];
                    const int other = (ad.a == nd.out) ? ad.b : ad.a;
                    if (nd.dual >= 0) {
                        Node& g = ND[nd.dual];
                        Tensor& s2 = T[g.a];
                        int n2 = 0; consumer_format(s2, g.cd, &n2, "finalize");
                        st.src2_t = g.a; st.src2_f = find_form(s2, FORM_I8, n2, g.cd.input_signed ? 1 : 0);
                        pack_conv_weights(net, g, s2, T[g.out]);
                    } else {
                        st.res_t = other; st.res_f = find_form(T[other], FORM_I32, 0, 0);
                    }
                    const int dfl = T[nd.out].fl - T[other].fl;     // >0: residual shifts left
                    st.acc_shl = dfl < 0 ? -dfl : 0; st.res_shl = dfl > 0 ? dfl : 0;
                    st.relu1 = ad.relu;
                    out_t = ad.out;
                }
                if (nd.pool >= 0) {                      // ... and the average pool: the step's outputs are the POOLED tensor's forms
                    out_t = ND[nd.pool].out;
                    if (T[out_t].forms.empty()) add_form(T[out_t], FORM_I32, 0, 0);
                }
                Tensor& o = T[out_t];
                if (o.dense_out) { st.dense = true; st.out.t = out_t; }
                else select_outputs(net, out_t, &st.out, &extra);
                const f8_conv_desc& d = nd.cd;
                const double opix = (double)T[nd.out].H * T[nd.out].W;
                st.ops_per_img = 2.0 * opix * d.cout * (d.cin / d.groups) * d.kernel * d.kernel;
                double b = (double)s.H * s.W * (nd.stem ? 4 : s.Cs);            // input once
                if (st.res_t >= 0) b += opix * o.Cs * 4;
                if (nd.dual >= 0) {
                    const Node& g = ND[nd.dual];
                    b += (double)T[g.a].H * T[g.a].W * T[g.a].Cs;
                    st.ops_per_img += 2.0 * opix * g.cd.cout * g.cd.cin;
                }
                if (st.dense) b += (double)d.cout * 4;
                const double outpix = nd.pool >= 0 ? 1.0 : opix;
                if (st.out.f32 >= 0) b += outpix * o.Cs * 4;
                for (int k = 0; k < 2; ++k) if (st.out.f8[k] >= 0) b += outpix * o.Cs;
                st.bytes_per_img = b;
                st.valu_per_img = ((st.res_t >= 0 || nd.dual >= 0) ? 2.0 * opix * o.C : 0.0) + (nd.pool >= 0 ? opix * o.C : 0.0) + 3.0 * outpix * o.Cs * out_forms8(st) + (nd.depthwise ? 9.0 * opix * d.cout / 4.0 : 0.0);   // (a VALU depthwise conv: one v_dot4 per four taps)
                st.bytes_const = nd.depthwise ? (double)s.Cs * 13 : (double)nd.coutP * (nd.ktot + 4);
                if (nd.dual >= 0) st.bytes_const += (double)ND[nd.dual].coutP * (ND[nd.dual].ktot + 4);
                label_conv_step(net, st, nd);
                if (st.dense) {                          // the classifier: logits straight into the caller's buffer (f8_fc.hip)
                    pack_frag_weights(net, nd);
                    st.name = "linear_dense:" + tname(net, nd.out);
                    char kb[64];
                    snprintf(kb, sizeof kb, "f8::fc_dense_kernel<%d>", nd.ck);
                    st.kernel = kb;
                }
                // 1x1 convs (plain, with the residual join, or as the dual GEMM of a stage-opening block) whose weight slice per wave
                // fits the register file: weight-stationary kernel, when a launch gives every workgroup a few pixel tiles to walk
                if (nd.pool >= 0) {
                    pack_frag_weights(net, nd);
                    const size_t colon = st.name.find(':');
                    st.name = std::string(st.res_t >= 0 ? "conv1x1_res+avgpool" : "conv1x1+avgpool") + (colon == std::string::npos ? ":" + tname(net, nd.out) : st.name.substr(colon));
                    char kb[96];
                    snprintf(kb, sizeof kb, "f8::conv1x1_pool_kernel<%d, %s>", nd.ck, st.res_t >= 0 ? "true" : "false");
                    st.kernel = kb;
                } else
                if (opt.wstat && !nd.depthwise && !nd.stem && d.kernel == 1 && d.pad == 0 && d.groups == 1 && !st.dense) {
                    const int k1 = nd.dual >= 0 ? ND[nd.dual].ktot : 0;
                    const bool has_res = nd.dual < 0 && st.res_t >= 0;
                    const bool g_ok = nd.dual < 0 || (ND[nd.dual].cd.kernel == 1 && ND[nd.dual].cd.pad == 0 && ND[nd.dual].cd.groups == 1);
                    if (g_ok && conv1x1_wstat_supported(nd.ck, k1, nd.coutP, has_res)) {
                        const int imgs = opt.whole_batch_launches ? max_batch : std::max(1, max_batch / opt.split);
                        const long tiles = ((long)imgs * (long)opix + 31) / 32;
                        const int ngroups = nd.coutP / (32 * conv1x1_wstat_waves(nd.ck, k1));
                        const int cus = net->num_cu > 0 ? net->num_cu : 256;
                        if (tiles >= (long)opt.wstat_min_tiles * std::max(1, cus / ngroups)) {
                            nd.wstat = true;
                            pack_frag_weights(net, nd);
                            if (nd.dual >= 0) pack_frag_weights(net, ND[nd.dual]);
                            const size_t colon = st.name.find(':');
                            st.name = std::string(nd.dual >= 0 ? "conv1x1_wstat_dual" : has_res ? "conv1x1_wstat_res" : "conv1x1_wstat") +
                                      (colon == std::string::npos ? ":" + tname(net, nd.out) : st.name.substr(colon));
                            char kb[144];
                            bool fast = opt.wstat_fast && (!st.relu0 || (nd.dual < 0 && !has_res && st.out.f32 < 0));      // keep in sync with conv1x1_wstat_fast
                            for (int k = 0; k < 2; ++k) if (st.out.f8[k] >= 0 && T[st.out.t].forms[st.out.f8[k]].n <= 0) fast = false;
                            snprintf(kb, sizeof kb, "f8::conv1x1_wstat_kernel<%d, %d, %d, %s, %s, %d, %s>", nd.ck, k1, conv1x1_wstat_waves(nd.ck, k1), has_res ? "true" : "false",
                                     st.out.f32 >= 0 ? "true" : "false", (st.out.f8[0] >= 0 ? 1 : 0) + (st.out.f8[1] >= 0 ? 1 : 0), fast ? "true" : "false");
                            st.kernel = kb;
                        }
                    }
                }
                // the stride-2 3x3 of a late stage-opening block, int8 outputs only: input patch in LDS, weights straight to registers
                if (nd.pool < 0 && opt.s2wreg && !nd.depthwise && !nd.stem && d.kernel == 3 && d.stride == 2 && d.pad == 1 && d.groups == 1 && nd.fused_add < 0 &&
                    nd.dual < 0 && st.out.f32 < 0 && !st.dense && nd.ck == d.cin && s.H == 2 * T[nd.out].H && s.W == 2 * T[nd.out].W &&
                    conv3x3s2_wreg_supported(nd.ck, T[nd.out].H, T[nd.out].W, nd.coutP)) {
                    nd.s2w = true;
                    pack_frag_weights(net, nd);
                    st.name = "conv3x3s2_wreg:" + tname(net, nd.out);
                    char kb[96];
                    snprintf(kb, sizeof kb, "f8::conv3x3s2_wreg_kernel<%d, %d, %d, %d>", nd.ck, T[nd.out].H, T[nd.out].W, nd.coutP);
                    st.kernel = kb;
                }
                // late, weight-heavy 1x1 convs with int8 outputs only: weights straight to registers (f8_wreg.hip)
                if (nd.pool < 0 && !nd.wstat && opt.wreg && !nd.depthwise && !nd.stem && d.kernel == 1 && d.stride == 1 && d.pad == 0 && d.groups == 1 && nd.fused_add < 0 &&
                    nd.dual < 0 && st.out.f32 < 0 && !st.dense && conv1x1_wreg_supported(nd.ck, nd.coutP)) {
                    nd.wreg = true;
                    pack_frag_weights(net, nd);
                    st.name = "conv1x1s1_wreg:" + tname(net, nd.out);
                    char kb[96];
                    snprintf(kb, sizeof kb, "f8::conv1x1_wreg_kernel<%d, %d>", nd.ck, nd.coutP);
                    st.kernel = kb;
                }
                break;
            }
            case N_ADD: {
                st.kind = S_ADD;
                st.src_t = nd.a; st.src_f = find_form(T[nd.a], FORM_I32, 0, 0);
                st.res_t = nd.b; st.res_f = find_form(T[nd.b], FORM_I32, 0, 0);
                const int dfl = T[nd.a].fl - T[nd.b].fl;
                st.acc_shl = dfl < 0 ? -dfl : 0; st.res_shl = dfl > 0 ? dfl : 0;
                st.relu1 = nd.relu;
                select_outputs(net, nd.out, &st.out, &extra);
                st.name = "add:" + tname(net, nd.out); st.kernel = "f8::add_kernel";
                const double e = (double)T[nd.out].H * T[nd.out].W * T[nd.out].Cs;
                st.bytes_per_img = e * 8 + (st.out.f32 >= 0 ? e * 4 : 0) + (st.out.f8[0] >= 0 ? e : 0) + (st.out.f8[1] >= 0 ? e : 0);
                break;
            }
            case N_MAXPOOL: {
                st.kind = S_MAXPOOL;
                Tensor& s = T[nd.a]; Tensor& o = T[nd.out];
                st.src_t = nd.a;
                const bool i8 = (o.forms.size() == 1 && o.forms[0].kind == FORM_I8);
                st.src_f = i8 ? find_form(s, FORM_I8, o.forms[0].n, o.forms[0].sgn) : find_form(s, FORM_I32, 0, 0);
                if (o.forms.empty()) add_form(o, FORM_I32, 0, 0);
                select_outputs(net, nd.out, &st.out, &extra);
                st.name = std::string(i8 ? "maxpool_i8:" : "maxpool_i32:") + tname(net, nd.out);
                st.kernel = (i8 && (s.Cs & 15) == 0 && nd.ppad < nd.pk) ? "f8::maxpool_i8x16_kernel" : "f8::maxpool_kernel";   // keep in sync with launch_maxpool
                const double ei = (double)s.H * s.W * s.Cs, eo = (double)o.H * o.W * o.Cs;
                st.bytes_per_img = ei * (i8 ? 1 : 4) + (st.out.f32 >= 0 ? eo * 4 : 0) + (st.out.f8[0] >= 0 ? eo : 0) + (st.out.f8[1] >= 0 ? eo : 0);
                break;
            }
            case N_AVGPOOL: {
                st.kind = S_AVGPOOL;
                Tensor& s = T[nd.a]; Tensor& o = T[nd.out];
                st.src_t = nd.a; st.src_f = find_form(s, FORM_I32, 0, 0);
                if (o.forms.empty()) add_form(o, FORM_I32, 0, 0);
                select_outputs(net, nd.out, &st.out, &extra);
                st.name = "avgpool_sum:" + tname(net, nd.out); st.kernel = "f8::avgpool_kernel";
                st.bytes_per_img = (double)s.H * s.W * s.Cs * 4 + (double)o.Cs * 5;
                break;
            }
        }
        net->steps.push_back(st);
        emit_requants(st.out.t, extra);
    }
    if (!T[net->out_t].dense_out) {
        Step st; st.kind = S_OUTPUT; st.node = T[net->out_t].prod;
        st.src_t = net->out_t; st.src_f = find_form(T[net->out_t], FORM_I32, 0, 0);
        st.name = "output:" + tname(net, net->out_t); st.kernel = "f8::output_kernel";
        st.bytes_per_img = (double)T[net->out_t].H * T[net->out_t].W * (T[net->out_t].Cs + T[net->out_t].C) * 4;
        net->steps.push_back(st);
    }

    for (auto& st : net->steps)
        if (st.kind == S_INPUT) {      // keep in sync with launch_input
            const Tensor& o = T[st.out.t];
            bool only_stem = !o.forms.empty();
            for (auto& F : o.forms) only_stem = only_stem && F.kind == FORM_STEM;
            if (only_stem && (o.W & 3) == 0 && o.C <= 4) st.kernel = "f8::input_stem4_kernel";
            // the fused stem launch can read the raw input itself: the input step then launches nothing (run_step)
            if (only_stem && o.C == 3 && net->opt.fuse_input) {
                Step* stem = nullptr; int users = 0;
                for (auto& s2 : net->steps) if (s2.src_t == st.out.t) { ++users; if (s2.kind == S_STEMPOOL || s2.kind == S_HEAD2) stem = &s2; }
                if (stem && users == 1) {
                    st.raw_input = stem->raw_input = true;
                    stem->bytes_per_img += (double)o.C * o.H * o.W * 4 - (double)o.H * o.W * 4;      // int32 planes instead of the NHWC4 copy
                    st.bytes_per_img = 0;
                    st.name = "input(read by the stem launch)";
                }
            }
        }
    return F8_OK;
}

// pass 4
static int layout_arena(f8_net* net, int max_batch) {
    auto& T = net->tensors;
    auto& ND = net->nodes;
    const int nn = (int)ND.size();
    const Options& opt = net->opt;
    const int fuse_blocks = opt.fuse_blocks;
    (void)T; (void)nn; (void)opt; (void)fuse_blocks; (void)max_batch;
    // ---- 4. lifetimes and arena layout (first-fit over a free list; in-place residual update)
    auto touch = [&](int t, int f, int step) {
        if (t < 0 || f < 0) return;
        Form& F = T[t].forms[f];
        if (F.first < 0) F.first = step;
        F.last = std::max(F.last, step);
    };
    for (size_t si = 0; si < net->steps.size(); ++si) {
        Step& st = net->steps[si];
        touch(st.src_t, st.src_f, (int)si);
        touch(st.res_t, st.res_f, (int)si);
        touch(st.src2_t, st.src2_f, (int)si);
        if (st.out.t >= 0 && !st.dense) {
            touch(st.out.t, st.out.f32, (int)si);
            touch(st.out.t, st.out.f8[0], (int)si);
            touch(st.out.t, st.out.f8[1], (int)si);
            if (st.kind == S_INPUT)
                for (size_t f = 0; f < T[st.out.t].forms.size(); ++f) touch(st.out.t, (int)f, (int)si);
        }
    }
    struct Blk { size_t off, size; };
    std::vector<Blk> freel;
    size_t top = 0;
    auto alloc = [&](size_t sz) -> size_t {
        sz = round_up_z(sz, 256);
        size_t best = (size_t)-1; int bi = -1;
        for (size_t i = 0; i < freel.size(); ++i)
            if (freel[i].size >= sz && freel[i].size < best) { best = freel[i].size; bi = (int)i; }
        if (bi >= 0) {
            size_t off = freel[bi].off;
            if (freel[bi].size == sz) freel.erase(freel.begin() + bi);
            else { freel[bi].off += sz; freel[bi].size -= sz; }
            return off;
        }
        size_t off = top; top += sz; return off;
    };
    auto release = [&](size_t off, size_t sz) {
        sz = round_up_z(sz, 256);
        freel.push_back({off, sz});
        std::sort(freel.begin(), freel.end(), [](const Blk& a, const Blk& b) { return a.off < b.off; });
        for (size_t i = 0; i + 1 < freel.size();)
            if (freel[i].off + freel[i].size == freel[i + 1].off) { freel[i].size += freel[i + 1].size; freel.erase(freel.begin() + i + 1); }
            else ++i;
    };
    for (auto& t : T)
        for (auto& F : t.forms) {
            if (F.kind == FORM_I32) { F.bytes_per_img = (size_t)t.H * t.W * t.Cs * 4; F.slack = (size_t)32 * t.Cs * 4; }
            else if (F.kind == FORM_I8) F.bytes_per_img = (size_t)t.H * t.W * t.Cs;
            else F.bytes_per_img = (size_t)F.Hp * F.Wp * 4;
        }
    // the stem image keeps a zero halo that is written once at upload: give it a private region
    for (auto& t : T)
        for (auto& F : t.forms)
            if (F.kind == FORM_STEM) {
                F.off = alloc(F.bytes_per_img * max_batch);
                net->stem_zero_off = F.off; net->stem_zero_bytes = F.bytes_per_img * max_batch;
                net->stem_zero_val = F.sgn ? 0 : 0x80;
            }
    // Chunked execution (for_each_launch) interleaves the steps of a chunk group: chunk c runs steps i..j-1 before chunk c+1
    // does.  A form that dies inside the group is therefore still needed by later chunks while forms born later in the group
    // are already being written by earlier chunks: for the allocator the whole group is ONE step (forms born in it live from
    // its first step, forms dying in it until its last).  Groups are taken as large as any option setting can make them.
    const int ns_all = (int)net->steps.size();
    std::vector<int> gstart(ns_all), gend(ns_all);
    {
        auto chunkable_w = [&](int k) -> int {
            const Step& st = net->steps[k];
            if (st.kind != S_FUSED) return 0;
            const Tensor& x = T[st.src_t];
            return (x.W == x.H && (x.W == 56 || x.W == 28 || x.W == 14)) ? x.W : 0;
        };
        for (int i = 0; i < ns_all;) {
            int j = i + 1;
            const int w = chunkable_w(i);
            if (w) while (j < ns_all && chunkable_w(j) == w) ++j;
            for (int k = i; k < j; ++k) { gstart[k] = i; gend[k] = j - 1; }
            i = j;
        }
    }
    for (int si = 0; si < ns_all; ++si) {
        // forms born at this (super) step, in the order of their producing steps
        if (gstart[si] == si)
            for (int sj = si; sj <= gend[si]; ++sj) {
                Step& st = net->steps[sj];
                for (size_t ti = 0; ti < T.size(); ++ti)
                    for (size_t f = 0; f < T[ti].forms.size(); ++f) {
                        Form& F = T[ti].forms[f];
                        if (F.kind == FORM_STEM || F.first != sj) continue;
                        // in-place residual: out32 of a conv/add step may overwrite a residual operand that dies here
                        // (each thread reads its element before writing it; same geometry, so it also holds per chunk)
                        if (F.kind == FORM_I32 && (int)ti == st.out.t && (int)f == st.out.f32 && st.res_t >= 0 &&
                            (st.kind == S_CONV || st.kind == S_ADD || st.kind == S_FUSED || st.kind == S_IR)) {
                            Form& R = T[st.res_t].forms[st.res_f];
                            if (R.last == sj && R.bytes_per_img == F.bytes_per_img && R.kind == FORM_I32) {
                                F.off = R.off; R.last = -2;   // ownership moves to F
                                continue;
                            }
                        }
                        F.off = alloc(F.bytes_per_img * max_batch + F.slack);
                    }
            }
        // forms dying at this (super) step
        if (gend[si] == si)
            for (size_t ti = 0; ti < T.size(); ++ti)
                for (auto& F : T[ti].forms)
                    if (F.kind != FORM_STEM && F.first >= 0 && F.last >= gstart[si] && F.last <= si) release(F.off, F.bytes_per_img * max_batch + F.slack);
    }
    net->arena_bytes = top;
    for (auto& t : T)
        for (auto& F : t.forms)
            if (F.bytes_per_img * (size_t)max_batch >= (1ull << 31))
                return fail(F8_ERR_UNSUPPORTED, "finalize: a tensor exceeds 2 GiB at max_batch %d (buffer addressing)", max_batch);
    return F8_OK;
}

int f8_net_finalize(f8_net* net, int max_batch) {
    if (!net) return fail(F8_ERR_INVALID, "f8_net_finalize: null net");
    if (net->finalized) return fail(F8_ERR_STATE, "f8_net_finalize: already finalized");
    if (max_batch < 1) return fail(F8_ERR_INVALID, "f8_net_finalize: max_batch < 1");
    if (net->out_t < 0) return fail(F8_ERR_STATE, "f8_net_finalize: no output marked");
    if (net->nodes.empty() || net->nodes[0].kind != N_INPUT) return fail(F8_ERR_STATE, "f8_net_finalize: first node must be the input");
    plan_residual_joins(net, max_batch);            // 1:  a residual add rides in the epilogue of the later of its two producers
    plan_bottleneck_blocks(net, max_batch);         // 1b: bottleneck identity blocks (one launch, or body.0 + body.2 on the 7x7 maps)
    plan_dual_gemm_joins(net, max_batch);           // 1c: body.4 || shortcut of a stage-opening block as one dual GEMM
    plan_stage_opening_blocks(net, max_batch);      // 1d: whole stage-opening blocks (same resolution / stride 2)
    plan_stage_chains(net, max_batch);              // 1f: all consecutive bottleneck blocks of a stage in one launch
    plan_basic_block_chains(net, max_batch);        // 1g: the same for BasicBlocks
    plan_inverted_residuals(net, max_batch);        // 1e: MobileNet-V2 inverted residuals
    plan_mobilenet_v2_head(net, max_batch);         // 1h: MobileNet-V2 head conv + depthwise + 1x1
    plan_last_conv_and_pool(net, max_batch);        // 1i: the last 1x1 conv + the average pool
    plan_tensor_forms(net, max_batch);              // 2:  which forms of each tensor exist in HBM
    int rc = emit_steps(net, max_batch);            // 3:  the launches, packed weights, algorithmic bytes / ops
    if (rc) return rc;
    if ((rc = layout_arena(net, max_batch))) return rc;   // 4: lifetimes, first-fit arena
    net->max_batch = max_batch;
    net->finalized = true;
    return F8_OK;
}

size_t f8_net_describe(const f8_net* net, char* buf, size_t cap) {
    std::string s;
    if (net && net->finalized) {
        char line[512];
        for (size_t i = 0; i < net->steps.size(); ++i) {
            const Step& st = net->steps[i];
            int n8 = (st.out.f8[0] >= 0) + (st.out.f8[1] >= 0);
            snprintf(line, sizeof line, "%3zu %-58s out[i32=%d i8=%d dense=%d] res=%d relu=%d/%d shl=%d/%d\n", i, st.name.c_str(),
                     st.out.f32 >= 0, n8, (int)st.dense, st.res_t >= 0, st.relu0, st.relu1, st.acc_shl, st.res_shl);
            s += line;
        }
        snprintf(line, sizeof line, "arena=%zu B (max_batch %d) weights=%zu B launches=%zu\n", net->arena_bytes, net->max_batch,
                 net->wblob.size(), net->steps.size());
        s += line;
    }
    if (buf && cap) { size_t n = std::min(cap - 1, s.size()); memcpy(buf, s.data(), n); buf[n] = 0; }
    return s.size() + 1;
}
int f8_net_num_launches(const f8_net* net) { return (net && net->finalized) ? (int)net->steps.size() : F8_ERR_STATE; }
size_t f8_net_arena_bytes(const f8_net* net) { return net ? net->arena_bytes : 0; }
size_t f8_net_weight_bytes(const f8_net* net) { return net ? net->wblob.size() : 0; }
int f8_net_output_fraclen(const f8_net* net) { return (net && net->out_t >= 0) ? net->tensors[net->out_t].fl : F8_ERR_STATE; }
size_t f8_net_output_elems(const f8_net* net) {
    if (!net || net->out_t < 0) return 0;
    const Tensor& t = net->tensors[net->out_t];
    return (size_t)t.C * t.H * t.W;
}

int f8_net_launch_info(const f8_net* net, int i, int N, char* name, size_t name_cap, double* alg_bytes, double* alg_ops) {
    if (!net || !net->finalized) return fail(F8_ERR_STATE, "f8_net_launch_info: not finalized");
    if (i < 0 || i >= (int)net->steps.size()) return fail(F8_ERR_INVALID, "f8_net_launch_info: index");
    const Step& st = net->steps[i];
    if (name && name_cap) { snprintf(name, name_cap, "%s", st.name.c_str()); }
    if (alg_bytes) *alg_bytes = st.bytes_per_img * N + st.bytes_const;
    if (alg_ops) *alg_ops = st.ops_per_img * N;
    return F8_OK;
}

int f8_net_launch_valu(const f8_net* net, int i, int N, double* essential_lane_ops) {
    if (!net || !net->finalized) return fail(F8_ERR_STATE, "f8_net_launch_valu: not finalized");
    if (i < 0 || i >= (int)net->steps.size() || !essential_lane_ops) return fail(F8_ERR_INVALID, "f8_net_launch_valu: index / null pointer");
    *essential_lane_ops = net->steps[i].valu_per_img * N;
    return F8_OK;
}

int f8_net_launch_kernel(const f8_net* net, int i, char* buf, size_t cap) {
    if (!net || !net->finalized) return fail(F8_ERR_STATE, "f8_net_launch_kernel: not finalized");
    if (i < 0 || i >= (int)net->steps.size()) return fail(F8_ERR_INVALID, "f8_net_launch_kernel: index");
    if (buf && cap) snprintf(buf, cap, "%s", net->steps[i].kernel.c_str());
    return F8_OK;
}

// Image groups resident at once in a chain launch (grid = groups x tiles per image, one workgroup per CU): a group walks
// ceil(N / groups) images, so of all group counts that need the same number of rounds the SMALLEST is taken — every group then
// has the same number of images (128 images on 18 groups of 14 tiles is 8 rounds for 2 groups and 7 for 16; on 16 groups it is 8
// for all, in the same time, on 224 CUs instead of 252) and the CUs left over run the other batches in flight.
// (taking every resident group instead — uneven rounds, the groups that finish early free their CUs — measured 1-3 % slower, round 4)
static int chain_groups(int N, int max_groups) {
    const int g = std::max(1, std::min(N, max_groups));
    const int rounds = (N + g - 1) / g;
    return (N + rounds - 1) / rounds;
}

// ------------------------------------------------------------------------------ executor
int f8_net_upload(f8_net* net) {
    if (!net || !net->finalized) return fail(F8_ERR_STATE, "f8_net_upload: not finalized");
    if (net->uploaded) return F8_OK;
    hipError_t e;
    const int parts_cap = std::max(net->opt.split, net->opt.arena_copies);     // arena copies: sub-batches of one run, or whole runs in flight (pipelining mode 2)
    if ((e = hipGetDevice(&net->device)) != hipSuccess) return hip_fail(e, "hipGetDevice");
    net->n_copies = parts_cap;
    { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, net->device) == hipSuccess && pr.multiProcessorCount > 0) net->num_cu = pr.multiProcessorCount; }
    net->arena_stride = round_up_z(std::max<size_t>(net->arena_bytes, 256), 4096);
    if ((e = hipMalloc((void**)&net->d_arena, net->arena_stride * parts_cap)) != hipSuccess) return hip_fail(e, "hipMalloc(arena)");
    if ((e = hipMalloc((void**)&net->d_w, std::max<size_t>(net->wblob.size(), 256))) != hipSuccess) return hip_fail(e, "hipMalloc(weights)");
    if (!net->wblob.empty() && (e = hipMemcpy(net->d_w, net->wblob.data(), net->wblob.size(), hipMemcpyHostToDevice)) != hipSuccess)
        return hip_fail(e, "hipMemcpy(weights)");
    if ((e = hipMalloc((void**)&net->d_err, 256)) != hipSuccess) return hip_fail(e, "hipMalloc(error words)");
    if ((e = hipMemset(net->d_err, 0, 256)) != hipSuccess) return hip_fail(e, "hipMemset(error words)");
    for (const Step& st : net->steps)
        if ((st.kind == S_CHAIN || st.kind == S_BCHAIN) && !net->d_chain) {
            size_t xchg = kChainXchgBytes;
            for (const Step& c7 : net->steps)
                if (c7.kind == S_CHAIN) {
                    const Tensor& o7 = net->tensors[net->nodes[net->nodes[c7.node].fused_add].out];   // the stage's map (c7.out.t: the pooled tensor when the pool runs in the launch)
                    if (cchain_supported(o7.C, net->nodes[c7.node].cd.cin, o7.H, o7.W, o7.C, false)) xchg = std::max(xchg, cchain_xchg_bytes());   // (geometry; the last host is an identity block's body.4)
                }
            net->chain_stride = round_up_z(4096 + xchg, 4096);
            if ((e = hipMalloc((void**)&net->d_chain, net->chain_stride * parts_cap)) != hipSuccess) return hip_fail(e, "hipMalloc(chain scratch)");
            if ((e = hipMemset(net->d_chain, 0, net->chain_stride * parts_cap)) != hipSuccess) return hip_fail(e, "hipMemset(chain scratch)");
            // a host-visible mirror of the error word: a chain launch that gives up a halo wait stores its code there as well, and f8_net_run looks at it
            // (a host read: no synchronisation) before it issues anything — a caller that never calls f8_net_check still gets a status (VERDICT r4 #5d)
            if (hipHostMalloc((void**)&net->h_err, 64, hipHostMallocMapped) == hipSuccess) {
                *net->h_err = 0u;
                if (hipHostGetDevicePointer((void**)&net->h_err_dev, net->h_err, 0) != hipSuccess) { (void)hipHostFree(net->h_err); net->h_err = nullptr; net->h_err_dev = nullptr; }
            } else { net->h_err = nullptr; (void)hipGetLastError(); }
        }
    for (int p = 0; p < parts_cap; ++p)
        if (net->stem_zero_bytes && (e = hipMemset(net->d_arena + p * net->arena_stride + net->stem_zero_off, net->stem_zero_val, net->stem_zero_bytes)) != hipSuccess)
            return hip_fail(e, "hipMemset(stem halo)");
    if ((e = hipDeviceSynchronize()) != hipSuccess) return hip_fail(e, "upload sync");
    net->uploaded = true;
    return F8_OK;
}

// Runs one launch for images [n0, n0 + N) of the batch.  Every sub-batch works in its OWN copy of the arena
// (index `part`): the arena packs tensors by lifetime assuming the steps of one batch run in order, so two
// sub-batches that are at different steps at the same time must not share it (a later, larger tensor of the
// sub-batch that is ahead would overlap an earlier tensor the other one is still reading).
static int run_step(f8_net* net, const Step& st, const int32_t* input, void* output, int n0, int N, int part, hipStream_t s) {
    auto& T = net->tensors;
    char* A = net->d_arena + (size_t)part * net->arena_stride;
    auto fp = [&](const Form& F) -> char* { return A + F.off + (size_t)net->chunk_off * F.bytes_per_img; };
    const Node& nd = net->nodes[st.node];
    auto fill_out = [&](int32_t** out32, QuantOut q[2]) {
        *out32 = nullptr; q[0].ptr = q[1].ptr = nullptr; q[0].n = q[1].n = 0; q[0].lo = q[1].lo = 0; q[0].hi = q[1].hi = 0;
        if (st.out.t < 0 || st.dense) return;
        const Tensor& o = T[st.out.t];
        if (st.out.f32 >= 0) *out32 = (int32_t*)fp(o.forms[st.out.f32]);
        for (int k = 0; k < 2; ++k) if (st.out.f8[k] >= 0) set_q(q[k], fp(o.forms[st.out.f8[k]]), o.forms[st.out.f8[k]]);
    };
    hipError_t e = hipSuccess;
    switch (st.kind) {
        case S_INPUT: {
            const Tensor& o = T[st.out.t];
            if (st.raw_input && !(net->in_u8 && net->in_u8_nhwc)) break;       // the stem launch reads the caller's buffer
            InArgs a{}; a.x = input + (size_t)n0 * o.C * o.H * o.W; a.N = N; a.C = o.C; a.H = o.H; a.W = o.W;
            if (net->in_f32) { a.xf = net->in_f32 + (size_t)n0 * o.C * o.H * o.W; a.scale = net->in_scale; a.qlo = net->in_lo; a.qhi = net->in_hi; }
            if (net->in_u8) { a.xu8 = net->in_u8 + (size_t)n0 * o.C * o.H * o.W; a.u8_nhwc = net->in_u8_nhwc; memcpy(a.lut, net->in_lut, sizeof a.lut); }
            for (auto& F : o.forms) {
                if (F.kind == FORM_I8) { if (F.n == 0) { a.out8 = (int8_t*)fp(F); a.Cs8 = o.Cs; if (!F.sgn) a.xor8 = 0x80808080u; } }
                else if (F.kind == FORM_I32) { a.out32 = (int32_t*)fp(F); a.Cs32 = o.Cs; }
                else { a.stem = (int8_t*)fp(F); a.Hp = F.Hp; a.Wp = F.Wp; a.pad = F.pad; if (!F.sgn) a.xor8 = 0x80808080u; }
                // an int32 input that is narrowed to 8 bits without a requant (head format): values outside the format would wrap silently
                if (net->opt.check_input_range && !net->in_f32 && !net->in_u8 && ((F.kind == FORM_I8 && F.n == 0) || F.kind == FORM_STEM)) {
                    a.err = net->d_err; a.chk_lo = F.sgn ? -127 : 0; a.chk_hi = F.sgn ? 127 : 255;
                }
            }
            e = launch_input(a, s);
            break;
        }
        case S_CONV: {
            const Tensor& sT = T[st.src_t]; const Form& sF = sT.forms[st.src_f];
            const Tensor& oT = T[nd.out];
            const f8_conv_desc& d = nd.cd;
            ConvArgs a{};
            a.x = (const int8_t*)fp(sF); a.x_bytes = (uint32_t)(sF.bytes_per_img * N);
            a.w = (const int8_t*)(net->d_w + nd.w_off); a.w_bytes = (uint32_t)((size_t)nd.coutP * nd.ktot);
            a.bias = (const int32_t*)(net->d_w + nd.b_off);
            a.PQ = oT.H * oT.W; a.Q = oT.W; a.M = N * a.PQ;
            make_magic((uint32_t)a.PQ, &a.mPQ, &a.s1PQ, &a.s2PQ);
            make_magic((uint32_t)a.Q, &a.mQ, &a.s1Q, &a.s2Q);
            a.stride = d.stride; a.kh = d.kernel; a.CK = nd.ck; a.ktot = nd.ktot; a.coutP = nd.coutP;
            a.ncc = nd.ncc;
            if (nd.ncc > 0) { a.rowcls = (const uint8_t*)(net->d_w + nd.rc_off); a.colcls = (const uint8_t*)(net->d_w + nd.cc_off); }
            if (nd.stem) {
                a.sN = (int)sF.bytes_per_img; a.sP = d.stride * sF.Wp * 4; a.sQ = d.stride * 4;
                a.origin = 0; a.H = sF.Hp; a.W = sF.Wp; a.pad = 0; a.kw = 1;
                a.tapH = sF.Wp * 4; a.tapW = 0;
            } else {
                a.sN = sT.H * sT.W * sT.Cs; a.sP = d.stride * sT.W * sT.Cs; a.sQ = d.stride * sT.Cs;
                a.origin = -(d.pad * sT.W + d.pad) * sT.Cs; a.H = sT.H; a.W = sT.W; a.pad = d.pad; a.kw = d.kernel;
                a.tapH = sT.W * sT.Cs; a.tapW = sT.Cs;
            }
            a.relu0 = st.relu0; a.deep_nk = net->opt.deep_nk; a.no_fast = net->opt.wstat_fast ? 0 : 1;
            if (st.res_t >= 0) { a.res = (const int32_t*)fp(T[st.res_t].forms[st.res_f]); a.acc_shl = st.acc_shl; a.res_shl = st.res_shl; a.relu1 = st.relu1; }
            if (nd.dual >= 0) {
                const Node& g = net->nodes[nd.dual];
                const Tensor& s2 = T[st.src2_t]; const Form& F2 = s2.forms[st.src2_f];
                a.x2 = (const int8_t*)fp(F2); a.x2_bytes = (uint32_t)(F2.bytes_per_img * N);
                a.w2 = (const int8_t*)(net->d_w + g.w_off); a.w2_bytes = (uint32_t)((size_t)g.coutP * g.ktot);
                a.bias2 = (const int32_t*)(net->d_w + g.b_off);
                a.sN2 = s2.H * s2.W * s2.Cs; a.sP2 = g.cd.stride * s2.W * s2.Cs; a.sQ2 = g.cd.stride * s2.Cs; a.ktot2 = g.ktot;
                a.acc_shl = st.acc_shl; a.res_shl = st.res_shl; a.relu1 = st.relu1;
            }
            fill_out(&a.out32, a.q);
            if (st.dense) {
                a.w = (const int8_t*)(net->d_w + nd.wf_off);
                e = launch_fc_dense(a, (char*)output + (size_t)n0 * oT.C * 4, oT.C, net->out_float, net->d_chain ? (const uint32_t*)(net->d_chain + (size_t)part * net->chain_stride + kChainErrWord * 4) : nullptr, net->epoch, s);
            } else if (nd.pool >= 0) { a.w = (const int8_t*)(net->d_w + nd.wf_off); e = launch_conv1x1_pool(a, s); }
            else if (nd.s2w) { a.w = (const int8_t*)(net->d_w + nd.wf_off); e = launch_conv3x3s2_wreg(a, s); }
            else if (nd.wstat) {
                a.w = (const int8_t*)(net->d_w + nd.wf_off);
                if (nd.dual >= 0) a.w2 = (const int8_t*)(net->d_w + net->nodes[nd.dual].wf_off);
                e = launch_conv1x1_wstat(a, net->num_cu, s);
            } else if (nd.wreg) { a.w = (const int8_t*)(net->d_w + nd.wf_off); e = launch_conv1x1_wreg(a, s); }
            else e = nd.p3_R > 0 ? launch_conv3x3_patch(a, d.cin, s) : launch_conv(a, nd.tile, s);
            break;
        }
        case S_HEAD2: {
            const Node& hh = net->nodes[nd.h2_head]; const Node& hb = net->nodes[nd.h2_dw];
            const Tensor& sT = T[st.src_t]; const Form& sF = sT.forms[st.src_f];
            const Tensor& oT = T[nd.out];
            StemPoolArgs a{};
            a.h2 = 1;
            a.x = (const int8_t*)fp(sF); a.x_bytes = (uint32_t)(sF.bytes_per_img * N);
            a.w = (const int8_t*)(net->d_w + hh.w_off); a.w_bytes = (uint32_t)((size_t)hh.coutP * hh.ktot);
            a.bias = (const int32_t*)(net->d_w + hh.b_off);
            a.wd = (const int8_t*)(net->d_w + hb.w_off); a.bd = (const int32_t*)(net->d_w + hb.cc_off);
            a.w1 = (const int8_t*)(net->d_w + nd.w_off); a.b1 = (const int32_t*)(net->d_w + nd.b_off);
            { int v = 0; consumer_format(T[hb.a], hb.cd, &v, "run"); a.na = v; consumer_format(T[nd.a], nd.cd, &v, "run"); a.nb = v; }
            a.N = N; a.Hp = sF.Hp; a.Wp = sF.Wp; a.org = sF.pad - hh.cd.pad;
            a.Pc = oT.H; a.Qc = oT.W; a.P = oT.H; a.Q = oT.W;
            a.relu0 = 1; a.grid_div = net->opt.stem_grid_div;
            a.acc_ok = conv_acc_bounded(hh) && conv_acc_bounded(hb) && conv_acc_bounded(nd); a.rq_int = !net->opt.requant_float || !a.acc_ok;
            a.rC = sT.C; a.rH = sT.H; a.rW = sT.W; a.xor8 = sF.sgn ? 0u : 0x80808080u;
            a.raw_kind = -1;
            if (st.raw_input && !(net->in_u8 && net->in_u8_nhwc)) {
                const size_t img = (size_t)sT.C * sT.H * sT.W;
                if (net->in_u8) { a.raw_kind = 2; a.xu8 = net->in_u8 + (size_t)n0 * img; memcpy(a.lut, net->in_lut, sizeof a.lut); }
                else if (net->in_f32) { a.raw_kind = 1; a.xf = net->in_f32 + (size_t)n0 * img; a.scale = net->in_scale; a.qlo = net->in_lo; a.qhi = net->in_hi; }
                else {
                    a.raw_kind = 0; a.xi = input + (size_t)n0 * img;
                    if (net->opt.check_input_range) { a.err = net->d_err; a.chk_lo = sF.sgn ? -127 : 0; a.chk_hi = sF.sgn ? 127 : 255; }
                }
            }
            fill_out(&a.out32, a.q);
            e = launch_stem_pool(a, s);
            break;
        }
        case S_STEMPOOL: {
            const Node& pl = net->nodes[nd.sp_pool];
            const Tensor& sT = T[st.src_t]; const Form& sF = sT.forms[st.src_f];
            const Tensor& cT = T[nd.out]; const Tensor& oT = T[pl.out];
            StemPoolArgs a{};
            a.x = (const int8_t*)fp(sF); a.x_bytes = (uint32_t)(sF.bytes_per_img * N);
            a.w = (const int8_t*)(net->d_w + nd.w_off); a.w_bytes = (uint32_t)((size_t)nd.coutP * nd.ktot);
            a.bias = (const int32_t*)(net->d_w + nd.b_off);
            a.N = N; a.Hp = sF.Hp; a.Wp = sF.Wp; a.org = sF.pad - nd.cd.pad;
            a.Pc = cT.H; a.Qc = cT.W; a.P = oT.H; a.Q = oT.W;
            a.relu0 = st.relu0; a.wpc = net->opt.stem_wpc; a.rows = net->opt.stem_rows; a.grid_div = net->opt.stem_grid_div;
            a.acc_ok = conv_acc_bounded(nd); a.rq_int = !net->opt.requant_float;
            a.raw_kind = -1;
            if (st.raw_input && !(net->in_u8 && net->in_u8_nhwc)) {
                const size_t img = (size_t)sT.C * sT.H * sT.W;
                a.rC = sT.C; a.rH = sT.H; a.rW = sT.W; a.xor8 = sF.sgn ? 0u : 0x80808080u;
                if (net->in_u8) { a.raw_kind = 2; a.xu8 = net->in_u8 + (size_t)n0 * img; memcpy(a.lut, net->in_lut, sizeof a.lut); }
                else if (net->in_f32) { a.raw_kind = 1; a.xf = net->in_f32 + (size_t)n0 * img; a.scale = net->in_scale; a.qlo = net->in_lo; a.qhi = net->in_hi; }
                else {
                    a.raw_kind = 0; a.xi = input + (size_t)n0 * img;
                    if (net->opt.check_input_range) { a.err = net->d_err; a.chk_lo = sF.sgn ? -127 : 0; a.chk_hi = sF.sgn ? 127 : 255; }
                }
            }
            fill_out(&a.out32, a.q);
            e = launch_stem_pool(a, s);
            break;
        }
        case S_FUSED: {
            if (nd.fbd_a >= 0) {      // stage-opening block: nd = shortcut conv, nd.dual = body.4
                const Node& na = net->nodes[nd.fbd_a]; const Node& nb = net->nodes[nd.fbd_b]; const Node& ng = net->nodes[nd.dual];
                const Tensor& x = T[st.src_t]; const Form& xF = x.forms[st.src_f];
                FusedArgs a{};
                a.x8 = (const int8_t*)fp(xF); a.x_bytes = (uint32_t)(xF.bytes_per_img * N);
                a.w0 = (const int8_t*)(net->d_w + na.w_off); a.w0_bytes = (uint32_t)((size_t)na.coutP * na.ktot);
                a.w2 = (const int8_t*)(net->d_w + nb.w_off); a.w2_bytes = (uint32_t)((size_t)nb.coutP * nb.ktot);
                a.w4 = (const int8_t*)(net->d_w + ng.w_off); a.w4_bytes = (uint32_t)((size_t)ng.coutP * ng.ktot);
                a.wsc = (const int8_t*)(net->d_w + nd.w_off); a.wsc_bytes = (uint32_t)((size_t)nd.coutP * nd.ktot);
                a.b0 = (const int32_t*)(net->d_w + na.b_off); a.b2 = (const int32_t*)(net->d_w + nb.b_off);
                a.b4 = (const int32_t*)(net->d_w + ng.b_off); a.bsc = (const int32_t*)(net->d_w + nd.b_off);
                a.N = N; a.H = x.H; a.W = x.W; a.C = na.cd.cin; a.MID = na.cd.cout; a.COUT = nd.cd.cout; a.R = nd.fb_R;
                a.tiles_per_img = nd.fbd_s2 ? (x.H / 2) / nd.fb_R : (x.H + nd.fb_R - 1) / nd.fb_R;
                a.stride2 = nd.fbd_s2 ? 1 : 0; a.stg = net->opt.opener_stg;
                auto fmt = [&](const Node& cons, const Tensor& src, int32_t* n, int32_t* lo, int32_t* hi, uint32_t* x_or) {
                    int nn = 0; consumer_format(src, cons.cd, &nn, "run");
                    *n = nn; *lo = cons.cd.input_signed ? -127 : 0; *hi = cons.cd.input_signed ? 127 : 255;
                    *x_or = cons.cd.input_signed ? 0u : 0x80808080u;
                };
                fmt(nb, T[nb.a], &a.n1, &a.lo1, &a.hi1, &a.xor1);
                fmt(ng, T[ng.a], &a.n2, &a.lo2, &a.hi2, &a.xor2);
                a.relu_a = na.cd.relu; a.relu_b = nb.cd.relu;
                a.acc_shl = st.acc_shl; a.res_shl = st.res_shl; a.relu1 = st.relu1;
                fill_out(&a.out32, a.q);
                e = nd.fbd_s2 ? launch_fused_opener(a, s) : launch_fused_bottleneck(a, s);
                break;
            }
            const Node& na = net->nodes[nd.fb_a]; const Node& nb = net->nodes[nd.fb_b];
            const Tensor& x = T[st.src_t]; const Form& xF = x.forms[st.src_f];
            FusedArgs a{};
            a.x8 = (const int8_t*)fp(xF); a.x_bytes = (uint32_t)(xF.bytes_per_img * N);
            a.xr = (const int32_t*)fp(T[st.res_t].forms[st.res_f]);
            a.w0 = (const int8_t*)(net->d_w + na.w_off); a.w0_bytes = (uint32_t)((size_t)na.coutP * na.ktot);
            a.w2 = (const int8_t*)(net->d_w + nb.w_off); a.w2_bytes = (uint32_t)((size_t)nb.coutP * nb.ktot);
            a.w4 = (const int8_t*)(net->d_w + nd.w_off); a.w4_bytes = (uint32_t)((size_t)nd.coutP * nd.ktot);
            a.b0 = (const int32_t*)(net->d_w + na.b_off); a.b2 = (const int32_t*)(net->d_w + nb.b_off); a.b4 = (const int32_t*)(net->d_w + nd.b_off);
            a.N = N; a.H = x.H; a.W = x.W; a.C = na.cd.cin; a.MID = na.cd.cout; a.R = nd.fb_R;
            a.tiles_per_img = (x.H + nd.fb_R - 1) / nd.fb_R;
            auto fmt = [&](const Node& cons, const Tensor& src, int32_t* n, int32_t* lo, int32_t* hi, uint32_t* x_or) {
                int nn = 0; consumer_format(src, cons.cd, &nn, "run");
                *n = nn; *lo = cons.cd.input_signed ? -127 : 0; *hi = cons.cd.input_signed ? 127 : 255;
                *x_or = cons.cd.input_signed ? 0u : 0x80808080u;
            };
            fmt(nb, T[nb.a], &a.n1, &a.lo1, &a.hi1, &a.xor1);
            fmt(nd, T[nd.a], &a.n2, &a.lo2, &a.hi2, &a.xor2);
            a.relu_a = na.cd.relu; a.relu_b = nb.cd.relu;
            a.acc_shl = st.acc_shl; a.res_shl = st.res_shl; a.relu1 = st.relu1;
            fill_out(&a.out32, a.q);
            e = launch_fused_bottleneck(a, s);
            break;
        }
        case S_CHAIN: {
            const std::vector<int>& ch = nd.chain;
            ChainArgs a{};
            a.nblk = (int)ch.size();
            auto fmt = [&](const Node& cons, const Tensor& src, int32_t* n, int32_t* lo, int32_t* hi, uint32_t* x_or) {
                int nn = 0; consumer_format(src, cons.cd, &nn, "run");
                *n = nn; *lo = cons.cd.input_signed ? -127 : 0; *hi = cons.cd.input_signed ? 127 : 255;
                *x_or = cons.cd.input_signed ? 0u : 0x80808080u;
            };
            for (int k = 0; k < a.nblk; ++k) {
                const Node& hh = net->nodes[ch[k]];
                ChainBlk& B = a.blk[k];
                if (hh.tail) {                                  // only the join of a stride-2 opening block: shortcut (hh) + body.4 (its dual)
                    const Node& g4 = net->nodes[hh.dual];
                    B.w4 = (const int8_t*)(net->d_w + g4.wf_off); B.b4 = (const int32_t*)(net->d_w + g4.b_off);
                    B.wsc = (const int8_t*)(net->d_w + hh.wf_off); B.bsc = (const int32_t*)(net->d_w + hh.b_off);
                    B.nq = B.n1 = B.n2 = 1; B.hiq = B.hi1 = B.hi2 = 255; B.xorq = B.xor1 = B.xor2 = 0x80808080u;     // (unused: no body.0 / body.2 here)
                    B.relu_a = B.relu_b = 1; B.relu1 = net->nodes[hh.fused_add].relu;
                    a.acc_ok = 1; a.stream_ok = stream_bounded(net, net->nodes[hh.fused_add].out) ? 1 : 0; a.rq_int = !net->opt.requant_float;
                    const int dfl = T[hh.out].fl - T[g4.out].fl;             // (shortcut << acc_shl) + (body.4 << res_shl)
                    B.acc_shl = dfl < 0 ? -dfl : 0; B.res_shl = dfl > 0 ? dfl : 0;
                    continue;
                }
                const bool hds = hh.fbd_a >= 0;
                const Node& na = net->nodes[hds ? hh.fbd_a : hh.fb_a]; const Node& nb = net->nodes[hds ? hh.fbd_b : hh.fb_b];
                const Node& ng = hds ? net->nodes[hh.dual] : hh;
                B.w0 = (const int8_t*)(net->d_w + na.wf_off); B.w2 = (const int8_t*)(net->d_w + nb.wf_off); B.w4 = (const int8_t*)(net->d_w + ng.wf_off);
                B.b0 = (const int32_t*)(net->d_w + na.b_off); B.b2 = (const int32_t*)(net->d_w + nb.b_off); B.b4 = (const int32_t*)(net->d_w + ng.b_off);
                if (hds) { B.wsc = (const int8_t*)(net->d_w + hh.wf_off); B.bsc = (const int32_t*)(net->d_w + hh.b_off); }
                const Tensor& xin = T[na.a];
                fmt(na, xin, &B.nq, &B.loq, &B.hiq, &B.xorq);
                fmt(nb, T[nb.a], &B.n1, &B.lo1, &B.hi1, &B.xor1);
                fmt(ng, T[ng.a], &B.n2, &B.lo2, &B.hi2, &B.xor2);
                B.relu_a = na.cd.relu; B.relu_b = nb.cd.relu; B.relu1 = net->nodes[hh.fused_add].relu;
                if (k == 0) { a.acc_ok = 1; a.rq_int = !net->opt.requant_float; a.stream_ok = (hds || stream_bounded(net, na.a)) ? 1 : 0; }   // identity first block: the stream it reads
                a.acc_ok = a.acc_ok && conv_acc_bounded(na) && conv_acc_bounded(nb);
                a.stream_ok = a.stream_ok && stream_bounded(net, net->nodes[hh.fused_add].out);
                // identity: (body.4 << acc_shl) + (block input << res_shl); opening block: (shortcut << acc_shl) + (body.4 << res_shl)
                const int dfl = T[hh.out].fl - (hds ? T[ng.out].fl : xin.fl);
                B.acc_shl = dfl < 0 ? -dfl : 0; B.res_shl = dfl > 0 ? dfl : 0;
            }
            const Node& hf = net->nodes[ch[0]];
            const bool tail = hf.tail, ds = hf.fbd_a >= 0 || tail;
            const Tensor& x = T[st.src_t]; const Form& xF = x.forms[st.src_f];
            if (ds) a.x8in = (const int8_t*)fp(xF); else a.xr = (const int32_t*)fp(xF);
            if (tail) { a.tail = 1; a.m2in = (const int8_t*)fp(T[st.src2_t].forms[st.src2_f]); }
            const Node& a0 = tail ? net->nodes[hf.dual] : net->nodes[ds ? hf.fbd_a : hf.fb_a];
            const Tensor& oT = T[net->nodes[nd.fused_add].out];      // the stage's output map (st.out.t: the pooled tensor when the pool runs in the launch)
            const int C = oT.C, MID = tail ? a0.cd.cin : a0.cd.cout;
            a.pool = nd.pool >= 0 ? 1 : 0;
            int wg_per_cu = 1;
            chain_shape(C, MID, oT.H, oT.W, tail ? hf.cd.cin : a0.cd.cin, tail, &a.R, &wg_per_cu);
            const int tiles = (oT.H + a.R - 1) / a.R;
            const int slots = (net->num_cu > 0 ? std::min(net->num_cu, 256) : 256) * wg_per_cu;
            // every workgroup of a chain launch must be resident: a device with fewer slots than one image has tiles cannot run it
            if (slots < tiles)
                return fail(F8_ERR_STATE, "f8_net_run: a stage-chain launch needs %d co-resident workgroups per image, the device has %d compute units (plan with fuse_chain = 0 / fuse_bchain = 0)", tiles, net->num_cu);
            a.N = N; a.NG = chain_groups(N, slots / tiles);
            if ((!ds || tail) && cchain_supported(C, MID, oT.H, oT.W, tail ? hf.cd.cin : a0.cd.cin, tail)) {      // the 7x7 stage: clusters of eight workgroups, four images per cluster and round
                a.NG = cchain_clusters(N, slots);
                if (a.NG < 1) return fail(F8_ERR_STATE, "f8_net_run: the 7x7 stage-chain launch needs 8 co-resident workgroups, the device has %d compute units (plan with fuse_chain7 = 0)", net->num_cu);
            }
            fill_out(&a.out32, a.q);
            if (!net->d_chain) return fail(F8_ERR_STATE, "f8_net_run: chain scratch missing");
            a.sync = (uint32_t*)(net->d_chain + (size_t)part * net->chain_stride);
            a.err = a.sync + kChainErrWord; a.err_host = net->h_err_dev; a.epoch = net->epoch;
            a.xchg = (int8_t*)(net->d_chain + (size_t)part * net->chain_stride + 4096);
            a.timeout_ticks = (uint32_t)std::min<long long>((long long)net->opt.chain_timeout_ms * 100000ll, 0x7fffffffll);
            char kb[160] = "";
            e = launch_chain(a, C, MID, oT.H, oT.W, tail ? hf.cd.cin : a0.cd.cin, s, kb, sizeof kb);
            if (kb[0] && st.kernel != kb) st.kernel = kb;
            break;
        }
        case S_BCHAIN: {
            const std::vector<int>& ch = nd.bchain;
            BChainArgs a{};
            a.nblk = (int)ch.size();
            auto fmt = [&](const Node& cons, const Tensor& src, int32_t* n, int32_t* lo, int32_t* hi, uint32_t* x_or) {
                int nn = 0; consumer_format(src, cons.cd, &nn, "run");
                *n = nn; *lo = cons.cd.input_signed ? -127 : 0; *hi = cons.cd.input_signed ? 127 : 255;
                *x_or = cons.cd.input_signed ? 0u : 0x80808080u;
            };
            const bool ds = net->nodes[ch[0]].bds_a >= 0;
            for (int k = 0; k < a.nblk; ++k) {
                const Node& hk = net->nodes[ch[k]];
                const bool hds = hk.bds_a >= 0;                 // opening block: hk = its shortcut conv
                const Node& c2 = hds ? net->nodes[hk.bds_b] : hk; const Node& c1 = net->nodes[hds ? hk.bds_a : hk.bb_a];
                BChainBlk& B = a.blk[k];
                B.wa = (const int8_t*)(net->d_w + c1.wf_off); B.wb = (const int8_t*)(net->d_w + c2.wf_off);
                B.ba = (const int32_t*)(net->d_w + c1.b_off); B.bb = (const int32_t*)(net->d_w + c2.b_off);
                const Tensor& xin = T[c1.a];
                fmt(c1, xin, &B.nq, &B.loq, &B.hiq, &B.xorq);
                fmt(c2, T[c2.a], &B.n1, &B.lo1, &B.hi1, &B.xor1);
                B.relu_a = c1.cd.relu; B.relu1 = net->nodes[hk.fused_add].relu;
                if (k == 0) { a.acc_ok = 1; a.rq_int = !net->opt.requant_float; a.stream_ok = (hds || stream_bounded(net, c1.a)) ? 1 : 0; }   // identity first block: the stream it reads
                a.acc_ok = a.acc_ok && conv_acc_bounded(c1);
                a.stream_ok = a.stream_ok && stream_bounded(net, net->nodes[hk.fused_add].out);
                // identity: (second conv << acc_shl) + (block input << res_shl); opening block: (second conv << acc_shl) + (shortcut << res_shl)
                const int dfl = T[c2.out].fl - (hds ? T[hk.out].fl : xin.fl);
                B.acc_shl = dfl < 0 ? -dfl : 0; B.res_shl = dfl > 0 ? dfl : 0;
                if (hds) { a.wsc = (const int8_t*)(net->d_w + hk.wf_off); a.bsc = (const int32_t*)(net->d_w + hk.b_off); }
            }
            const Tensor& xs = T[st.src_t];
            if (ds) { a.x8in = (const int8_t*)fp(xs.forms[st.src_f]); a.x8sc = (const int8_t*)fp(xs.forms[st.res_f]); }
            else a.xr = (const int32_t*)fp(xs.forms[st.src_f]);
            const Tensor& x = T[st.out.t];
            const int tiles = bchain_tiles_per_img(x.C, x.H, x.W);
            // every workgroup of a chain launch must be resident (one per CU): a device with fewer CUs than one image has tiles cannot run it
            if ((net->num_cu > 0 ? std::min(net->num_cu, 256) : 256) < tiles)
                return fail(F8_ERR_STATE, "f8_net_run: a stage-chain launch needs %d co-resident workgroups per image, the device has %d compute units (plan with fuse_chain = 0 / fuse_bchain = 0)", tiles, net->num_cu);
            a.N = N; a.NG = chain_groups(N, (net->num_cu > 0 ? std::min(net->num_cu, 256) : 256) / tiles);
            fill_out(&a.out32, a.q);
            if (!net->d_chain) return fail(F8_ERR_STATE, "f8_net_run: chain scratch missing");
            a.sync = (uint32_t*)(net->d_chain + (size_t)part * net->chain_stride);
            a.err = a.sync + kChainErrWord; a.err_host = net->h_err_dev; a.epoch = net->epoch;
            a.xchg = (int8_t*)(net->d_chain + (size_t)part * net->chain_stride + 4096);
            a.timeout_ticks = (uint32_t)std::min<long long>((long long)net->opt.chain_timeout_ms * 100000ll, 0x7fffffffll);
            char kb[160] = "";
            e = launch_bchain(a, x.C, x.H, x.W, s, kb, sizeof kb);
            if (kb[0] && st.kernel != kb) st.kernel = kb;
            break;
        }
        case S_P12: {
            const Node& na = net->nodes[nd.p12_a];
            const Tensor& x = T[st.src_t]; const Form& xF = x.forms[st.src_f];
            if (nd.p12_s2) {                                    // body.0 + body.2 of a stride-2 opening block on f8_opener.hip (P12); q[0] = mid2
                FusedArgs a{};
                a.x8 = (const int8_t*)fp(xF); a.x_bytes = (uint32_t)(xF.bytes_per_img * N);
                a.w0 = (const int8_t*)(net->d_w + na.w_off); a.w0_bytes = (uint32_t)((size_t)na.coutP * na.ktot);
                a.w2 = (const int8_t*)(net->d_w + nd.w_off); a.w2_bytes = (uint32_t)((size_t)nd.coutP * nd.ktot);
                a.b0 = (const int32_t*)(net->d_w + na.b_off); a.b2 = (const int32_t*)(net->d_w + nd.b_off);
                a.N = N; a.H = x.H; a.W = x.W; a.C = na.cd.cin; a.MID = na.cd.cout; a.COUT = 4 * na.cd.cout; a.R = nd.fb_R;
                a.tiles_per_img = (x.H / 2 + nd.fb_R - 1) / nd.fb_R;
                { int nn = 0; consumer_format(T[nd.a], nd.cd, &nn, "run"); a.n1 = nn; a.lo1 = nd.cd.input_signed ? -127 : 0; a.hi1 = nd.cd.input_signed ? 127 : 255;
                  a.xor1 = nd.cd.input_signed ? 0u : 0x80808080u; }
                a.relu_a = na.cd.relu; a.relu_b = nd.cd.relu;
                fill_out(&a.out32, a.q);
                a.n2 = a.q[0].n; a.lo2 = a.q[0].lo; a.hi2 = a.q[0].hi; a.xor2 = a.q[0].bias_xor;      // mid2's one form = body.4's input format
                a.stride2 = 1; a.p12only = 1; a.acc_ok = conv_acc_bounded(na) && conv_acc_bounded(nd); a.rq_int = !net->opt.requant_float;
                e = launch_fused_opener(a, s);
                break;
            }
            FusedArgs a{};
            a.x8 = (const int8_t*)fp(xF); a.x_bytes = (uint32_t)(xF.bytes_per_img * N);
            a.w0 = (const int8_t*)(net->d_w + na.wf_off); a.w0_bytes = (uint32_t)((size_t)na.coutP * na.ktot);     // fragment order
            a.w2 = (const int8_t*)(net->d_w + nd.wf_off); a.w2_bytes = (uint32_t)((size_t)nd.coutP * nd.ktot);
            a.b0 = (const int32_t*)(net->d_w + na.b_off); a.b2 = (const int32_t*)(net->d_w + nd.b_off);
            a.N = N; a.H = x.H; a.W = x.W; a.C = na.cd.cin; a.MID = na.cd.cout;
            { int nn = 0; consumer_format(T[nd.a], nd.cd, &nn, "run"); a.n1 = nn; a.lo1 = nd.cd.input_signed ? -127 : 0; a.hi1 = nd.cd.input_signed ? 127 : 255;
              a.xor1 = nd.cd.input_signed ? 0u : 0x80808080u; }
            a.relu_a = na.cd.relu; a.relu_b = nd.cd.relu;
            fill_out(&a.out32, a.q);
            e = launch_fused_p12(a, s);
            break;
        }
        case S_IR: {
            const Node& na = net->nodes[nd.ir_a]; const Node& nb = net->nodes[nd.ir_b];
            const Tensor& x = T[st.src_t]; const Form& xF = x.forms[st.src_f];
            const Tensor& oT = T[nd.out];
            IRArgs a{};
            a.x8 = (const int8_t*)fp(xF);
            if (st.res_t >= 0) a.xr = (const int32_t*)fp(T[st.res_t].forms[st.res_f]);
            a.w0 = (const int8_t*)(net->d_w + na.w_off); a.b0 = (const int32_t*)(net->d_w + na.b_off);
            a.wd4 = (const int8_t*)(net->d_w + nb.rc_off); a.bd4 = (const int32_t*)(net->d_w + nb.cc_off);
            a.w4 = (const int8_t*)(net->d_w + nd.w_off); a.b4 = (const int32_t*)(net->d_w + nd.b_off);
            a.N = N; a.H = x.H; a.W = x.W; a.Ho = oT.H; a.Wo = oT.W; a.stride = nb.cd.stride; a.R = nd.ir_R; a.G = nd.ir_G;
            a.tiles_per_img = oT.H / nd.ir_R; a.E32 = na.coutP;
            auto fmt = [&](const Node& cons, const Tensor& src, int32_t* n, int32_t* lo, int32_t* hi, uint32_t* x_or) {
                int nn = 0; consumer_format(src, cons.cd, &nn, "run");
                *n = nn; *lo = cons.cd.input_signed ? -127 : 0; *hi = cons.cd.input_signed ? 127 : 255;
                *x_or = cons.cd.input_signed ? 0u : 0x80808080u;
            };
            fmt(nb, T[nb.a], &a.n1, &a.lo1, &a.hi1, &a.xor1);
            fmt(nd, T[nd.a], &a.n2, &a.lo2, &a.hi2, &a.xor2);
            a.relu_a = na.cd.relu; a.relu_b = nb.cd.relu; a.relu0 = st.relu0;
            a.acc_ok = conv_acc_bounded(na) && conv_acc_bounded(nb); a.rq_int = !net->opt.requant_float;
            a.acc_shl = st.acc_shl; a.res_shl = st.res_shl; a.relu1 = st.relu1;
            make_magic((uint32_t)x.W, &a.mW, &a.s1W, &a.s2W);
            make_magic((uint32_t)(x.H * x.W), &a.mHW, &a.s1HW, &a.s2HW);
            make_magic((uint32_t)oT.W, &a.mWo, &a.s1Wo, &a.s2Wo);
            make_magic((uint32_t)(nd.ir_R * oT.W), &a.mRWo, &a.s1RWo, &a.s2RWo);
            fill_out(&a.out32, a.q);
            e = launch_fused_ir(a, x.Cs, nd.coutP, s);
            break;
        }
        case S_DW: {
            const Tensor& sT = T[st.src_t]; const Form& sF = sT.forms[st.src_f];
            const Tensor& oT = T[nd.out];
            DwArgs a{};
            a.x = (const int8_t*)fp(sF); a.w = (const int8_t*)(net->d_w + nd.w_off); a.bias = (const int32_t*)(net->d_w + nd.b_off);
            a.w4 = (const int8_t*)(net->d_w + nd.rc_off); a.bias4 = (const int32_t*)(net->d_w + nd.cc_off);
            a.N = N; a.H = sT.H; a.W = sT.W; a.P = oT.H; a.Q = oT.W; a.Cs = sT.Cs; a.stride = nd.cd.stride; a.pad = nd.cd.pad;
            a.in_signed = nd.cd.input_signed; a.relu0 = st.relu0; a.use_dot4 = net->opt.dw_dot4; a.use_mma = net->opt.dw_mma; a.acc_ok = conv_acc_bounded(nd); a.rq_int = !net->opt.requant_float;
            fill_out(&a.out32, a.q);
            e = launch_dwconv(a, s);
            break;
        }
        case S_ADD: case S_REQUANT: {
            const Tensor& sT = T[st.src_t];
            AddArgs a{};
            a.a = (const int32_t*)fp(sT.forms[st.src_f]);
            a.b = st.kind == S_ADD ? (const int32_t*)fp(T[st.res_t].forms[st.res_f]) : nullptr;
            a.M = N * sT.H * sT.W; a.Cs = sT.Cs; a.a_shl = st.acc_shl; a.b_shl = st.res_shl; a.relu = st.relu1;
            fill_out(&a.out32, a.q);
            e = launch_add(a, s);
            break;
        }
        case S_MAXPOOL: {
            const Tensor& sT = T[st.src_t]; const Form& sF = sT.forms[st.src_f];
            const Tensor& oT = T[nd.out];
            PoolArgs a{};
            a.x = fp(sF); a.in_is_i8 = sF.kind == FORM_I8; a.in_signed = sF.sgn;
            if (a.in_is_i8) { a.q[0].bias_xor = 0; }
            a.N = N; a.H = sT.H; a.W = sT.W; a.P = oT.H; a.Q = oT.W; a.Cs = sT.Cs; a.k = nd.pk; a.stride = nd.pstride; a.pad = nd.ppad;
            fill_out(&a.out32, a.q);
            e = launch_maxpool(a, s);
            break;
        }
        case S_AVGPOOL: {
            const Tensor& sT = T[st.src_t];
            AvgArgs a{};
            a.x = (const int32_t*)fp(sT.forms[st.src_f]); a.N = N; a.HW = sT.H * sT.W; a.Cs = sT.Cs;
            fill_out(&a.out32, a.q);
            e = launch_avgpool(a, s);
            break;
        }
        case S_OUTPUT: {
            const Tensor& sT = T[st.src_t];
            OutArgs a{};
            a.x = (const int32_t*)fp(sT.forms[st.src_f]); a.N = N; a.C = sT.C; a.HW = sT.H * sT.W; a.Cs = sT.Cs;
            a.out = (char*)output + (size_t)n0 * sT.C * sT.H * sT.W * 4; a.as_float = net->out_float;
            a.err = net->d_chain ? (const uint32_t*)(net->d_chain + (size_t)part * net->chain_stride + kChainErrWord * 4) : nullptr; a.epoch = net->epoch;
            e = launch_output(a, s);
            break;
        }
    }
    if (e != hipSuccess) return hip_fail(e, st.name.c_str());
    return F8_OK;
}

static int split_batch(const f8_net* net, int N, int cut[5]);

// Measured tile choice: every implicit-GEMM conv step is timed on the device with each tile that has a kernel
// instance (HIP events, best of a few repetitions, on garbage activations: integer kernels are data-independent in
// time) and keeps the fastest.  Results are bit-identical for every tile; only the plan description changes.
int f8_net_autotune(f8_net* net, int N, void* stream) {
    if (!net || !net->finalized) return fail(F8_ERR_STATE, "f8_net_autotune: not finalized");
    if (N < 1 || N > net->max_batch) return fail(F8_ERR_INVALID, "f8_net_autotune: N=%d outside [1,%d]", N, net->max_batch);
    int rc = f8_net_upload(net);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    int cut[5];
    (void)split_batch(net, N, cut);
    const int n_launch = cut[1] - cut[0];                      // images of one sub-batch launch
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(F8_ERR_HIP, "f8_net_autotune: events");
    static const int cand_bm[4] = {128, 128, 64, 64}, cand_bn[4] = {128, 64, 128, 64};
    int changed = 0;
    for (auto& st : net->steps) {
        if (st.kind != S_CONV) continue;
        Node& nd = net->nodes[st.node];
        if (nd.p3_R > 0 || nd.wreg || nd.wstat || nd.s2w || st.dense) continue;
        const ConvTile keep = nd.tile;
        ConvTile best = keep; float best_ms = 1e30f;
        for (int c = 0; c < 4; ++c) {
            ConvTile t = keep; t.bm = cand_bm[c]; t.bn = cand_bn[c];
            if (t.bn > 64 && nd.coutP <= 64) continue;
            if (nd.dual >= 0 && !(t.bn == 64 || (t.bn == 128 && t.bm == 128))) continue;     // dual-GEMM instances
            if (t.bm == 64 && t.bn == 32) continue;
            nd.tile = t;
            if (run_step(net, st, nullptr, nullptr, 0, n_launch, 0, s) != F8_OK) { (void)hipGetLastError(); continue; }   // no instance
            float ms_min = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0, s);
                for (int k = 0; k < 4; ++k) (void)run_step(net, st, nullptr, nullptr, 0, n_launch, 0, s);
                (void)hipEventRecord(e1, s);
                if (hipEventSynchronize(e1) != hipSuccess) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return fail(F8_ERR_HIP, "f8_net_autotune: sync"); }
                float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < ms_min) ms_min = ms;
            }
            const float bias = (t.bm == keep.bm && t.bn == keep.bn) ? 0.97f : 1.0f;   // switch only for a clear (> 3 %) win
            if (ms_min * bias < best_ms) { best_ms = ms_min * bias; best = t; }
        }
        nd.tile = best;
        if (best.bm != keep.bm || best.bn != keep.bn) { ++changed; label_conv_step(net, st, nd); }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return changed;
}

// Cuts the batch into up to F8_SPLIT (default 2, max 4) sub-batches; every I32T form needs each cut at a
// multiple of 32 pixels.  Returns the number of parts and their starts in cut[0..parts].
static int split_batch(const f8_net* net, int N, int cut[5]) {
    const int want = std::min(net->opt.split, net->n_copies > 0 ? net->n_copies : net->opt.split);   // never more parts than arena copies
    cut[0] = 0; cut[1] = N;
    if (want < 2 || N < 2) return 1;
    const int gran = 1;
    int parts = want;
    while (parts > 1 && (N / parts) / gran * gran == 0) --parts;
    if (parts < 2) return 1;
    const int per = (N / parts) / gran * gran;
    for (int p = 0; p < parts; ++p) cut[p] = p * per;
    cut[parts] = N;
    return parts;
}

// Chunked execution.  The fused bottleneck blocks of a stage run in chunks of images, block after block per chunk: a chunk's
// int32 stream (78 MB for 24 images at 56x56, 103 MB for 64 images at 28x28) is still in the memory-side cache when the next
// block reads it.  Measured, 56x56 identity blocks alone: 3.12 us per image in one 128-image launch, 2.66 us in 32-image
// launches.  Whole net at bs 128, same box: 56x56 chunks of 16 / 24 / 32 / 40 / 48 images = +1.0 / +2.1 / +1.4 / -1.4 / -0.7 %;
// then 28x28 chunks of 32 / 48 / 64 / 80 / 96 = +1.7 / +2.5 / +2.9 / +2.8 / +1.1 %; the 14x14 tensors (103 MB per batch) need
// none.  Applies in every schedule (a sub-batch of 64 images still runs its 56x56 blocks in chunks: non-pipelined run
// 73.4k -> 75.1k img/s).
// Defaults are DERIVED from two numbers, not hand-fitted per net:
//   cache : the int32 stream of a chunk (bytes per image of the wider of the block's int32-sized input and output maps: 3.2 MB
//           at 56x56x256, 1.6 MB at 28x28x512, 0.8 MB at 14x14x1024) should fit Options::chunk_budget_mb (96 MiB = 3/8 of the
//           256 MiB memory-side cache);
//   fill  : a chunk launch must still fill the chip: >= 85 % of (CUs x resident workgroups per CU) workgroups.
// chunk = max(cache, fill): 56x56 -> 30 images (cache), 28x28 -> 64 (fill: 7 tiles per image, 2 workgroups per CU),
// 14x14 -> 128 (fill: 2 tiles per image, 1 per CU), which is what sweeps over explicit sizes had found (24-32 / 64 / 128).
// chunk56 / chunk28 / chunk14 override (0 = whole batch).  Every chunk offset must keep the I32T blocks of the group's tensors
// aligned (chunk * H*W a multiple of 32 pixels, for the source map AND a stage-opening block's half-resolution output):
// chunks are multiples of 2 / 8 / 32 images.
static int chunk_gran(int W) { return W == 56 ? 2 : (W == 28 ? 8 : 32); }
static int step_chunk(const f8_net* net, int i) {        // nominal images per chunk of step i (0 = not chunked)
    const Step& st = net->steps[i];
    if (st.kind != S_FUSED) return 0;                    // (chunking the stage-opening convs as well: -3 %, more launches than locality)
    const Node& nd = net->nodes[st.node];
    const Options& o = net->opt;
    if (nd.fbd_a >= 0 && !(nd.fbd_s2 ? o.chunk_opener : o.chunk_ds)) return 0;
    const Tensor& x = net->tensors[st.src_t];
    if (x.W != x.H || !(x.W == 56 || x.W == 28 || x.W == 14)) return 0;
    int chunk = x.W == 56 ? o.chunk56 : (x.W == 28 ? o.chunk28 : o.chunk14);
    const int gran = chunk_gran(x.W);
    if (chunk < 0) {
        size_t per_img = (size_t)x.H * x.W * x.Cs * 4;
        if (st.out.t >= 0) { const Tensor& y = net->tensors[st.out.t]; per_img = std::max(per_img, (size_t)y.H * y.W * y.Cs * 4); }
        const int cache = (int)std::min<size_t>(((size_t)o.chunk_budget_mb << 20) / std::max<size_t>(per_img, 1), 1 << 20) / gran * gran;
        const int mid = net->nodes[nd.fbd_a >= 0 ? nd.fbd_a : nd.fb_a].cd.cout;
        const int tiles = std::max(1, (nd.fbd_s2 ? x.H / 2 : x.H) / std::max(1, nd.fb_R));        // workgroups per image
        const int slots = (net->num_cu > 0 ? net->num_cu : 256) * ((nd.fbd_s2 || mid > 128) ? 1 : 2);
        const int fill = ((slots * 85 / 100 + tiles - 1) / tiles + gran - 1) / gran * gran;
        chunk = std::max(cache, fill);
    }
    chunk = chunk / gran * gran;
    return chunk;
}
// f(step index, first image of the chunk, images) for every kernel launch of a run over N images, in launch order.
// A group = consecutive chunkable steps at one resolution with the same nominal chunk; the batch is cut into equal chunks
// (no tiny tail), and not at all when it exceeds the nominal chunk by less than a quarter.
extern "C++" template <class F>
static int for_each_launch(const f8_net* net, int N, F&& f) {
    const int ns = (int)net->steps.size();
    for (int i = 0; i < ns;) {
        const int nominal = step_chunk(net, i);
        const int W = nominal > 0 ? net->tensors[net->steps[i].src_t].W : 0;
        int k = nominal > 0 ? (N + nominal - 1) / nominal : 1;
        if (nominal > 0 && (long)N * 4 <= (long)nominal * 5) k = 1;
        if (k > 1) {
            const int gran = chunk_gran(W);
            const int chunk = ((N + k - 1) / k + gran - 1) / gran * gran;
            int j = i;
            while (j < ns && step_chunk(net, j) == nominal && net->tensors[net->steps[j].src_t].W == W) ++j;
            for (int c0 = 0; c0 < N; c0 += chunk)
                for (int kk = i; kk < j; ++kk) { const int rc = f(kk, c0, std::min(chunk, N - c0)); if (rc) return rc; }
            i = j;
        } else {
            const int rc = f(i, 0, N);
            if (rc) return rc;
            ++i;
        }
    }
    return F8_OK;
}
// all launches of one run for images [n0, n0 + N) of the network input, on stream s, in arena copy `part`
static int run_steps(f8_net* net, const int32_t* input, void* output, int n0, int N, int part, hipStream_t s) {
    const int rc = for_each_launch(net, N, [&](int k, int c0, int cn) {
        net->chunk_off = c0;
        return run_step(net, net->steps[k], input, output, n0, cn, part, s);
    });
    net->chunk_off = 0;
    return rc;
}

static int run_common(f8_net* net, const int32_t* input, void* output, int N, void* stream, float* ms, int cap) {
    // the one-shot event is consumed by THIS call whatever happens next: a run that fails validation must not leave it armed for a
    // later, unrelated run (by then the caller's event may be gone)
    hipEvent_t in_ready = nullptr;
    if (net) { in_ready = net->input_ready; net->input_ready = nullptr; }
    if (!net || !net->finalized) return fail(F8_ERR_STATE, "f8_net_run: not finalized");
    if (N < 1 || N > net->max_batch) return fail(F8_ERR_INVALID, "f8_net_run: N=%d outside [1,%d]", N, net->max_batch);
    if (!input || !output) return fail(F8_ERR_INVALID, "f8_net_run: null pointer");
    int rc = f8_net_upload(net);
    if (rc) return rc;
    if (net->opt.check_device) {       // the arena, the weights and the internal streams belong to the device of the upload
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev != net->device)
            return fail(F8_ERR_STATE, "f8_net_run: current device %d, but this net lives on device %d (hipSetDevice before the call; one handle per device)", dev, net->device);
    }
    hipStream_t s = (hipStream_t)stream;
    // this run's tag for the chain error words: a word an EARLIER run left behind (a transient halo time-out nobody collected with
    // f8_net_check) neither cuts this run's waits short nor poisons its logits; it stays where it is for f8_net_check to report.
    // (A replayed hipGraph carries the tag of its capture: there the word is sticky until f8_net_check, as it was before round 5.)
    if (net->h_err) {
        const uint32_t hw = *(volatile uint32_t*)net->h_err;
        if (hw) return fail(F8_ERR_HIP, "f8_net_run: a stage-chain launch of an earlier run gave up waiting for a neighbouring tile (code 0x%x of run tag %u): the logits of that run are "
                                        "poisoned (NaN / INT32_MIN); call f8_net_check to collect the error and re-arm the handle", hw & 0xffu, hw >> 8);
    }
    net->epoch = net->epoch % 0xffffffu + 1u;
    if (in_ready) {                    // the producer of this run's input (f8_net_set_input_ready); every schedule forks from / runs on `s`
        (void)hipStreamWaitEvent(s, in_ready, 0);
    }
    const int ns = (int)net->steps.size();
    int cut[5];
    int parts = split_batch(net, N, cut);
    if (ms && net->pipelined == 2) { parts = 1; cut[0] = 0; cut[1] = N; }   // time the launches the alternating mode issues
    if (ms) {
        // profiled: the parts back to back on the caller's stream, one event after every launch; a step's time is the sum
        // over its launches (parts x chunks)
        if (cap < ns) return fail(F8_ERR_INVALID, "f8_net_run_profiled: ms capacity %d < %d launches", cap, ns);
        std::vector<int> owner;                          // step index of every launch, in order
        for (int p = 0; p < parts; ++p) {
            owner.push_back(-1);                         // start marker of the part
            (void)for_each_launch(net, cut[p + 1] - cut[p], [&](int k, int, int) { owner.push_back(k); return 0; });
        }
        const int ne = (int)owner.size();
        if (net->n_events < ne) {
            if (net->events) { for (int i = 0; i < net->n_events; ++i) (void)hipEventDestroy(net->events[i]); delete[] net->events; }
            net->events = new hipEvent_t[ne]; net->n_events = ne;
            for (int i = 0; i < ne; ++i) { hipError_t e = hipEventCreate(&net->events[i]); if (e != hipSuccess) return hip_fail(e, "hipEventCreate"); }
        }
        int pos = 0;
        for (int p = 0; p < parts; ++p) {
            (void)hipEventRecord(net->events[pos++], s);
            rc = for_each_launch(net, cut[p + 1] - cut[p], [&](int k, int c0, int cn) {
                net->chunk_off = c0;
                const int r = run_step(net, net->steps[k], input, output, cut[p], cn, p, s);
                (void)hipEventRecord(net->events[pos++], s);
                return r;
            });
            net->chunk_off = 0;
            if (rc) return rc;
        }
        hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) return hip_fail(e, "f8_net_run_profiled: sync");
        for (int i = 0; i < ns; ++i) ms[i] = 0.f;
        for (int q = 1; q < ne; ++q) {
            if (owner[q] < 0) continue;
            float t = 0.f;
            (void)hipEventElapsedTime(&t, net->events[q - 1], net->events[q]);
            ms[owner[q]] += t;
        }
        // a step that launched nothing (the input step when the stem launch reads the caller's buffer) has no duration: the two
        // events around it are ~5 us apart on their own
        for (int i = 0; i < ns; ++i)
            if (net->steps[i].kind == S_INPUT && net->steps[i].raw_input && !(net->in_u8 && net->in_u8_nhwc)) ms[i] = 0.f;
        return F8_OK;
    }
    auto ensure_aux = [&]() -> int {
        if (net->aux[0] && !(net->opt.graph && net->aux_shared)) return F8_OK;
        if (net->aux[0]) {                               // `graph` was switched on after the pooled streams were taken: streams of its own
            for (int k = 0; k < 4; ++k) { (void)hipStreamSynchronize(net->aux[k]); net->aux[k] = nullptr; }
            net->aux_shared = false;
        }
        // The internal streams are ONE set per device, shared by every handle of the process.  HIP maps streams onto a few hardware
        // queues (GPU_MAX_HW_QUEUES, default 4) in creation order: the first handle's streams got a queue each, a later handle's
        // shared queues with them and its concurrent sub-batches / batches in flight serialised — measured, ResNet-18 with three
        // batches in flight: the handle created first 330 k img/s, every later one 308 k (all 332 k with GPU_MAX_HW_QUEUES=16;
        // tools/prof_intmodel.py).  Work of two handles on one stream is ordered, which costs nothing they would not contend for anyway.
        // (hipGraph capture needs streams of its own: a capture must not see another handle's launches.)
        static std::mutex pool_mu;
        static hipStream_t pool[64][4] = {};
        const int dv = net->device;
        if (net->opt.shared_streams && !net->opt.graph && dv >= 0 && dv < 64) {
            std::lock_guard<std::mutex> lk(pool_mu);
            for (int k = 0; k < 4; ++k) {
                if (!pool[dv][k]) {
                    hipError_t e = hipStreamCreateWithFlags(&pool[dv][k], hipStreamNonBlocking);
                    if (e != hipSuccess) { pool[dv][k] = nullptr; return hip_fail(e, "hipStreamCreate"); }
                }
            }
            for (int k = 0; k < 4; ++k) net->aux[k] = pool[dv][k];
            net->aux_shared = true;
        } else
        for (int k = 0; k < 4; ++k) {
            hipError_t e = hipStreamCreateWithFlags(&net->aux[k], hipStreamNonBlocking);
            if (e != hipSuccess) return hip_fail(e, "hipStreamCreate");
        }
        for (int k = 0; k < 5; ++k) {
            if (net->aux_ev[k]) continue;
            hipError_t e = hipEventCreateWithFlags(&net->aux_ev[k], hipEventDisableTiming);
            if (e != hipSuccess) return hip_fail(e, "hipEventCreate");
        }
        return F8_OK;
    };
    const int arena_copies = net->n_copies;
    const int use_streams = net->opt.split_streams;
    if (net->pipelined == 2 && arena_copies >= 2 && !use_streams)     // rocprofv3 runs: the same launches, alone on the caller's stream
        return run_steps(net, input, output, 0, N, 0, s);
    if (net->pipelined == 2 && arena_copies >= 2) {
        // alternating whole batches: run i executes UNSPLIT on internal stream / arena copy i % 2, so that two consecutive
        // runs are in flight together — the same occupancy as two concurrent sub-batches, but every launch covers the whole
        // batch (twice the workgroups per launch: at 128 images the latency-bound launches of the late stages fill the chip).
        // Fork: as in the lagged mode the stream waits for `s` as of the PREVIOUS run's entry (and, by stream order, for
        // run i-2 on the same arena copy); join: `s` waits for this run.
        // Options::pipeline_depth D (2..4, at most the arena copies): D runs in flight; the fork then lags D-1 entries.
        if ((rc = ensure_aux())) return rc;
        if (!net->start_ev[0])
            for (int k = 0; k < 4; ++k) (void)hipEventCreateWithFlags(&net->start_ev[k], hipEventDisableTiming);
        const int D = std::max(2, std::min(net->opt.pipeline_depth, arena_copies));
        if (net->prev_stream != s) net->pipe_count = 0;
        const int cur = net->start_idx, slot = net->alt_idx % D;
        (void)hipEventRecord(net->start_ev[cur], s);
        const int lagn = std::min(net->pipe_count, D - 1);
        hipEvent_t dep = net->start_ev[(cur - lagn + 4) & 3];
        (void)hipStreamWaitEvent(net->aux[slot], dep, 0);
        if (in_ready) (void)hipStreamWaitEvent(net->aux[slot], in_ready, 0);   // the lagged dependency does not cover this run's input
        net->start_idx = (cur + 1) & 3; ++net->pipe_count; net->prev_stream = s; net->alt_idx = (slot + 1) % D;
        if ((rc = run_steps(net, input, output, 0, N, slot, net->aux[slot]))) return rc;
        (void)hipEventRecord(net->aux_ev[1 + slot], net->aux[slot]);
        (void)hipStreamWaitEvent(s, net->aux_ev[1 + slot], 0);
        return F8_OK;
    }
    if (parts == 1) return run_steps(net, input, output, 0, N, 0, s);
    // F8_SPLIT_STREAMS=0: same launches, serialised on the caller's stream (used for rocprofv3 runs so
    // that per-kernel durations are not inflated by the overlap of the two sub-batches)
    if (!use_streams) {
        for (int p = 0; p < parts; ++p)
            if ((rc = run_steps(net, input, output, cut[p], cut[p + 1] - cut[p], p, s))) return rc;
        return F8_OK;
    }
    // independent sub-batches on internal streams: while one is in a layer's tail / epilogue phase the
    // others keep the CUs busy.  Fork from and join to the caller's stream with events (no host sync).
    if ((rc = ensure_aux())) return rc;
    // F8_GRAPH=1: the second call with the same (input, output, N, stream) captures the launches below into a hipGraph
    // (the aux streams join the capture through the fork event); later calls replay it with one hipGraphLaunch.
    // The legacy null stream cannot be captured: the graph then lives on an internal stream fenced by events.
    const int use_graph = net->opt.graph;
    bool capturing = false;
    hipStream_t user_s = s;
    auto graph_replay = [&]() -> int {
        hipStream_t gs = user_s ? user_s : net->aux[3];
        if (gs != user_s) { (void)hipEventRecord(net->aux_ev[0], user_s); (void)hipStreamWaitEvent(gs, net->aux_ev[0], 0); }
        hipError_t e = hipGraphLaunch(net->g_exec, gs);
        if (e != hipSuccess) return hip_fail(e, "hipGraphLaunch");
        if (gs != user_s) { (void)hipEventRecord(net->aux_ev[4], gs); (void)hipStreamWaitEvent(user_s, net->aux_ev[4], 0); }
        return F8_OK;
    };
    if (use_graph && parts <= 3 && !net->in_f32 && !net->in_u8) {
        const bool same = net->g_in == input && net->g_out == output && net->g_N == N && net->g_stream == user_s;
        if (same && net->g_exec) return graph_replay();
        if (!same) {
            if (net->g_exec) { (void)hipGraphExecDestroy(net->g_exec); net->g_exec = nullptr; }
            net->g_in = input; net->g_out = output; net->g_N = N; net->g_stream = user_s; net->g_warm = 0;
        }
        if (net->g_warm++ >= 1) {      // first call with a new key runs eagerly (one-time kernel attribute calls happen there)
            s = user_s ? user_s : net->aux[3];
            hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
            if (e != hipSuccess) return hip_fail(e, "hipStreamBeginCapture");
            capturing = true;
        }
    }
    if (net->pipelined && !capturing) {
        // lagged fork: the sub-batch streams order themselves behind the previous run on the same arena copy by stream
        // order; towards the caller they only wait for the state of `s` at the previous run's entry (header contract)
        if (!net->start_ev[0])
            for (int k = 0; k < 4; ++k) (void)hipEventCreateWithFlags(&net->start_ev[k], hipEventDisableTiming);
        if (net->prev_stream != s) net->pipe_count = 0;
        const int cur = net->start_idx;
        (void)hipEventRecord(net->start_ev[cur], s);
        hipEvent_t dep = net->start_ev[(cur - std::min(net->pipe_count, 1) + 4) & 3];
        for (int k = 0; k < parts; ++k) { (void)hipStreamWaitEvent(net->aux[k], dep, 0); if (in_ready) (void)hipStreamWaitEvent(net->aux[k], in_ready, 0); }
        net->start_idx = (cur + 1) & 3; ++net->pipe_count; net->prev_stream = s;
    } else {
        (void)hipEventRecord(net->aux_ev[0], s);
        for (int k = 0; k < parts; ++k) (void)hipStreamWaitEvent(net->aux[k], net->aux_ev[0], 0);
        net->pipe_count = 0;
    }
    // optional stagger: sub-batch p starts only after sub-batch p-1 has finished its first `lag` launches, so that the
    // streams do not march through the memory-bound and the latency-bound layers in lock step
    const int lag_env = net->opt.stagger, lag_pipe = net->opt.stagger_pipelined;
    const int lag = lag_env >= 0 ? lag_env : (net->pipelined ? lag_pipe : 2);   // measured: lag 0/1/2/4/8 = 56.06/56.37/56.65/56.34/55.1 k img/s
    // the launches of every part (chunked where a part is larger than a chunk), submitted round-robin so that no stream's
    // queue starts late
    struct Launch { int k, c0, cn; };
    std::vector<Launch> plan[4];
    size_t longest = 0;
    for (int p = 0; p < parts; ++p) {
        (void)for_each_launch(net, cut[p + 1] - cut[p], [&](int k, int c0, int cn) { plan[p].push_back({k, c0, cn}); return 0; });
        longest = std::max(longest, plan[p].size());
    }
    for (size_t t = 0; t < longest; ++t)
        for (int p = 0; p < parts; ++p) {
            if (t >= plan[p].size()) continue;
            const Launch& L = plan[p][t];
            net->chunk_off = L.c0;
            rc = run_step(net, net->steps[L.k], input, output, cut[p], L.cn, p, net->aux[p]);
            net->chunk_off = 0;
            if (rc) {
                if (capturing) { hipGraph_t g = nullptr; (void)hipStreamEndCapture(s, &g); if (g) (void)hipGraphDestroy(g); }
                return rc;
            }
            if (lag > 0 && (int)t == lag - 1 && p + 1 < parts) {
                if (!net->lag_ev[p]) (void)hipEventCreateWithFlags(&net->lag_ev[p], hipEventDisableTiming);
                (void)hipEventRecord(net->lag_ev[p], net->aux[p]);
                (void)hipStreamWaitEvent(net->aux[p + 1], net->lag_ev[p], 0);
            }
        }
    for (int k = 0; k < parts; ++k) {
        (void)hipEventRecord(net->aux_ev[1 + k], net->aux[k]);
        (void)hipStreamWaitEvent(s, net->aux_ev[1 + k], 0);
    }
    if (capturing) {
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(s, &g);
        if (e != hipSuccess) return hip_fail(e, "hipStreamEndCapture");
        e = hipGraphInstantiate(&net->g_exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) { net->g_exec = nullptr; return hip_fail(e, "hipGraphInstantiate"); }
        return graph_replay();
    }
    return F8_OK;
}

int f8_net_num_parts(const f8_net* net, int N) {
    if (!net || !net->finalized) return fail(F8_ERR_STATE, "f8_net_num_parts: not finalized");
    int cut[5];
    const int parts = split_batch(net, N, cut);
    return (net->pipelined == 2 && parts >= 2) ? 1 : parts;     // alternating whole batches: one launch set per run
}

int f8_net_step_launches(const f8_net* net, int i, int N) {
    if (!net || !net->finalized) return fail(F8_ERR_STATE, "f8_net_step_launches: not finalized");
    if (i < 0 || i >= (int)net->steps.size()) return fail(F8_ERR_INVALID, "f8_net_step_launches: launch index out of range");
    int cut[5];
    int parts = split_batch(net, N, cut);
    if (net->pipelined == 2 && parts >= 2) { parts = 1; cut[0] = 0; cut[1] = N; }
    int n = 0;
    for (int p = 0; p < parts; ++p)
        (void)for_each_launch(net, cut[p + 1] - cut[p], [&](int k, int, int) { n += (k == i); return 0; });
    return n;
}

int f8_net_run(f8_net* net, const int32_t* input, void* output, int N, void* stream) {
    return run_common(net, input, output, N, stream, nullptr, 0);
}
int f8_net_run_f32(f8_net* net, const float* images, int normalize, void* output, int N, void* stream) {
    if (!net || !net->finalized) return fail(F8_ERR_STATE, "f8_net_run_f32: not finalized");
    if (!images) return fail(F8_ERR_INVALID, "f8_net_run_f32: null pointer");
    // the consumer format of the network input: fraclen of the input tensor, signedness of the conv(s) reading it
    const Tensor& in = net->tensors[net->nodes[0].out];
    int sgn = -1;
    for (int c : in.consumers) {
        const Node& nd = net->nodes[c];
        if (nd.kind != N_CONV && nd.kind != N_LINEAR) return fail(F8_ERR_UNSUPPORTED, "f8_net_run_f32: the input must feed convolutions");
        if (sgn >= 0 && sgn != (nd.cd.input_signed ? 1 : 0)) return fail(F8_ERR_UNSUPPORTED, "f8_net_run_f32: consumers disagree on signedness");
        sgn = nd.cd.input_signed ? 1 : 0;
    }
    if (sgn < 0) return fail(F8_ERR_UNSUPPORTED, "f8_net_run_f32: the input has no consumer");
    if (normalize) {
        if (in.fl < 0 || in.fl > (sgn ? 7 : 8)) return fail(F8_ERR_INVALID, "f8_net_run_f32: input fraclen %d outside [0,%d]", in.fl, sgn ? 7 : 8);
        net->in_scale = (float)(1 << in.fl); net->in_lo = sgn ? -127 : 0; net->in_hi = sgn ? 127 : 255;
    } else {
        if (in.fl != 8 || sgn) return fail(F8_ERR_INVALID, "f8_net_run_f32: normalize == 0 needs an unsigned input at fraclen 8 (fix_train.py:689-692), net has fl %d %s",
                                           in.fl, sgn ? "signed" : "unsigned");
        net->in_scale = 255.f; net->in_lo = 0; net->in_hi = 255;      // images are in [0,1] (asserted >= 0 by the reference); 8-bit storage
    }
    net->in_f32 = images;
    const int rc = run_common(net, (const int32_t*)images, output, N, stream, nullptr, 0);
    net->in_f32 = nullptr;
    return rc;
}
// sign of the network input's consumers (0 unsigned, 1 signed, < 0 status)
static int input_sign(const f8_net* net, const char* who) {
    const Tensor& in = net->tensors[net->nodes[0].out];
    int sgn = -1;
    for (int c : in.consumers) {
        const Node& nd = net->nodes[c];
        if (nd.kind != N_CONV && nd.kind != N_LINEAR) return fail(F8_ERR_UNSUPPORTED, "%s: the input must feed convolutions", who);
        if (sgn >= 0 && sgn != (nd.cd.input_signed ? 1 : 0)) return fail(F8_ERR_UNSUPPORTED, "%s: consumers disagree on signedness", who);
        sgn = nd.cd.input_signed ? 1 : 0;
    }
    if (sgn < 0) return fail(F8_ERR_UNSUPPORTED, "%s: the input has no consumer", who);
    return sgn;
}

int f8_net_run_u8(f8_net* net, const uint8_t* images, int nhwc, int normalize, const float* mean, const float* stdv, void* output, int N, void* stream) {
    if (!net || !net->finalized) return fail(F8_ERR_STATE, "f8_net_run_u8: not finalized");
    if (!images) return fail(F8_ERR_INVALID, "f8_net_run_u8: null pointer");
    const Tensor& in = net->tensors[net->nodes[0].out];
    if (in.C > 3 && normalize) return fail(F8_ERR_UNSUPPORTED, "f8_net_run_u8: mean / std are given for 3 channels");
    const int sgn = input_sign(net, "f8_net_run_u8");
    if (sgn < 0) return sgn;
    // The decoder-side pipeline of the reference, per pixel value k of channel c, in the float32 operations torch executes
    // (fix_train.py:299-329 transforms.ToTensor / Normalize, then :683-692):
    //   t = float(k) / 255                        ToTensor
    //   normalize == 0:  x_int = round_half_even(255 * t)                  (== k; fraclen 8)
    //   normalize != 0:  t = (t - mean[c]) / std[c];  x_int = clamp(round_half_even(t * 2^fl), +-127 or [0,255])   (fix_quant)
    // A uint8 has 256 values: the whole pipeline is a 3 x 256 table built here on the host (IEEE single precision, one
    // rounding per operation as in torch) and looked up inside the input kernel.
    if (normalize) {
        if (!mean || !stdv) return fail(F8_ERR_INVALID, "f8_net_run_u8: normalize needs mean and std");
        if (in.fl < 0 || in.fl > (sgn ? 7 : 8)) return fail(F8_ERR_INVALID, "f8_net_run_u8: input fraclen %d outside [0,%d]", in.fl, sgn ? 7 : 8);
        for (int c = 0; c < 3; ++c) if (!(stdv[c] != 0.f)) return fail(F8_ERR_INVALID, "f8_net_run_u8: std[%d] is zero", c);
    } else if (in.fl != 8 || sgn) {
        return fail(F8_ERR_INVALID, "f8_net_run_u8: normalize == 0 needs an unsigned input at fraclen 8 (fix_train.py:689-692), net has fl %d %s", in.fl, sgn ? "signed" : "unsigned");
    }
    const float scale = (float)(1 << (normalize ? in.fl : 0));
    const float lo = sgn ? -127.f : 0.f, hi = sgn ? 127.f : 255.f;
    for (int c = 0; c < 3; ++c)
        for (int k = 0; k < 256; ++k) {
            volatile float t = (float)k / 255.0f;              // volatile: one rounding per operation, no contraction / excess precision
            int v;
            if (!normalize) { volatile float r = 255.0f * t; v = (int)__builtin_rintf(r); }
            else {
                volatile float d = t - mean[c];
                volatile float q = d / stdv[c];
                volatile float r = q * scale;
                float rr = __builtin_rintf(r);
                rr = rr < lo ? lo : (rr > hi ? hi : rr);
                v = (int)rr;
            }
            net->in_lut[c * 256 + k] = (int16_t)v;
        }
    net->in_u8 = images; net->in_u8_nhwc = nhwc ? 1 : 0;
    const int rc = run_common(net, (const int32_t*)images, output, N, stream, nullptr, 0);
    net->in_u8 = nullptr;
    return rc;
}

int f8_net_run_profiled(f8_net* net, const int32_t* input, void* output, int N, void* stream, float* ms, int cap) {
    if (!ms) return fail(F8_ERR_INVALID, "f8_net_run_profiled: null ms");
    return run_common(net, input, output, N, stream, ms, cap);
}

}  // extern "C"
