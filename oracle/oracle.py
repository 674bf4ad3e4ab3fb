"""CPU oracle: restatement of the reference's `int_op_only` forward (numpy + oracle/f8_oracle.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, `__graft_entry__.smoke()` and bench.py's
`cpu_baseline` leg, never by the product package `f8net_amd`.

Every tensor is an int32 NCHW numpy array, as in the reference (all int32 on CPU,
/root/reference/fix_train.py:933).  The `output_fraclen` Python attribute the reference hangs on
its tensors (fix_resnet.py:37,48,54,70,76) travels here as an explicit integer.

Pinned against the imported reference by tests/test_oracle_golden.py (fixtures written by
oracle/gen_golden.py in the build container).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, 'libf8oracle.so')
    src = os.path.join(_HERE, 'f8_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B', 'libf8oracle.so'])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.f8o_requant.restype = ctypes.c_int
        _LIB.f8o_conv2d.restype = ctypes.c_int
        _LIB.f8o_add_align.restype = ctypes.c_int
        _LIB.f8o_avgpool_sum.restype = ctypes.c_int
        _LIB.f8o_num_threads.restype = ctypes.c_int
    return _LIB


def num_threads() -> int:
    return lib().f8o_num_threads()


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _i32(a):
    a = np.ascontiguousarray(a)
    if a.dtype != np.int32:
        a = a.astype(np.int32)
    return a


# ---------------------------------------------------------------- op level

def requant(x, dst_fl: int, src_fl: int, signed: bool):
    """int_op_only_fix_quant(x, 8, dst_fl, src_fl, signed), fix_quant_ops.py:90-114."""
    x = _i32(x)
    y = np.empty_like(x)
    rc = lib().f8o_requant(_p(x), _p(y), ctypes.c_size_t(x.size), int(src_fl), int(dst_fl), int(bool(signed)))
    if rc != 0:
        raise AssertionError(f'requant: arguments rejected (src_fl={src_fl}, dst_fl={dst_fl}, signed={signed})')
    return y


def requant_py(x, dst_fl: int, src_fl: int, signed: bool):
    """Second, independent statement of the same function with Python big ints (small inputs only)."""
    out = []
    n = src_fl - dst_fl
    for v in np.asarray(x).reshape(-1).tolist():
        if n > 0:
            h = 1 << (n - 1)
            r = _wrap(v + h)
            q = _wrap((r >> (n + 1)) << 1) if (v % (1 << n)) == h else (r >> n)
        else:
            q = _wrap(v << (-n))
        lo, hi = (-127, 127) if signed else (0, 255)
        out.append(min(max(q, lo), hi))
    return np.array(out, dtype=np.int32).reshape(np.asarray(x).shape)


def _wrap(v: int) -> int:
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v >= (1 << 31) else v


def relu(x):
    return np.maximum(x, 0).astype(np.int32)


def conv2d(x, w, b, stride: int, pad: int, groups: int = 1):
    x, w = _i32(x), _i32(w)
    b = None if b is None else _i32(b)
    N, C, H, W = x.shape
    K, cg, kh, kw = w.shape
    assert cg * groups == C, (x.shape, w.shape, groups)
    P = (H + 2 * pad - kh) // stride + 1
    Q = (W + 2 * pad - kw) // stride + 1
    y = np.empty((N, K, P, Q), dtype=np.int32)
    rc = lib().f8o_conv2d(_p(x), _p(w), None if b is None else _p(b), _p(y),
                          N, C, H, W, K, kh, kw, stride, pad, groups)
    if rc != 0:
        raise AssertionError('conv2d: unsupported groups')
    return y


def linear(x, w, b):
    x, w = _i32(x), _i32(w)
    b = None if b is None else _i32(b)
    N, C = x.shape
    K = w.shape[0]
    y = np.empty((N, K), dtype=np.int32)
    lib().f8o_linear(_p(x), _p(w), None if b is None else _p(b), _p(y), N, C, K)
    return y


def add_align(res, x, res_fl: int, x_fl: int):
    """fix_resnet.py:40-54.  Returns (sum, output_fraclen)."""
    res = _i32(res).copy()
    x = _i32(x)
    assert res.shape == x.shape
    fl = lib().f8o_add_align(_p(res), _p(x), ctypes.c_size_t(res.size), int(res_fl), int(x_fl))
    if fl < 0:
        raise AssertionError('add_align: shift too large')
    return res, fl


def maxpool(x, k: int = 3, stride: int = 2, pad: int = 1):
    x = _i32(x)
    N, C, H, W = x.shape
    P = (H + 2 * pad - k) // stride + 1
    Q = (W + 2 * pad - k) // stride + 1
    y = np.empty((N, C, P, Q), dtype=np.int32)
    lib().f8o_maxpool(_p(x), _p(y), N, C, H, W, k, stride, pad)
    return y


def avgpool_sum(x):
    """FXQAvgPool2d int branch, fix_quant_ops.py:127-134.  Returns ([N,C] int32, fraclen increment 6
    for the 7x7 pool the nets use: shiftnum = round(log2(49)), :121-122)."""
    x = _i32(x)
    N, C, H, W = x.shape
    y = np.empty((N, C), dtype=np.int32)
    rc = lib().f8o_avgpool_sum(_p(x), _p(y), N, C, H, W)
    if rc != 0:
        raise AssertionError('avgpool_sum: reference assert (res <= 2^32-1) would fire')
    return y


AVGPOOL_SHIFT = 6   # FXQAvgPool2d(7).shiftnum


# ---------------------------------------------------------------- model level

def _conv_layer(spec_conv, params, x, x_fl, quant_input=True, tap=None):
    """requant -> conv -> tag fraclen, the inner step of IntBlock.forward (fix_resnet.py:29-37)."""
    k = spec_conv.key
    in_fl = int(params[k + '.input_fraclen'].reshape(-1)[0])
    w_fl = int(params[k + '.weight_fraclen'].reshape(-1)[0])
    if quant_input:
        x = requant(x, in_fl, x_fl, spec_conv.signed_in)
    y = conv2d(x, params[k + '.weight'], params[k + '.bias'], spec_conv.stride, spec_conv.pad,
               spec_conv.groups)
    if tap is not None:
        tap(k, y, in_fl + w_fl)          # conv output before the in-place ReLU
    if spec_conv.relu:
        y = relu(y)
    return y, in_fl + w_fl


def block_forward(bspec, params, x, x_fl, tap=None):
    """IntBlock.forward int branch: fix_resnet.py:26-77, fix_mobilenet_v2.py:20-48,
    fix_mobilenet_v1.py:25-38."""
    res, res_fl = x, x_fl
    for c in bspec.body:
        res, res_fl = _conv_layer(c, params, res, res_fl, tap=tap)
    if bspec.shortcut is not None:
        sx, sx_fl = _conv_layer(bspec.shortcut, params, x, x_fl, tap=tap)
        res, res_fl = add_align(res, sx, res_fl, sx_fl)
    elif bspec.residual:
        res, res_fl = add_align(res, x, res_fl, x_fl)
    if bspec.post_relu:
        res = relu(res)
    if tap is not None:
        tap(bspec.name, res, res_fl)
    return res, res_fl


def net_forward(spec, params, x, x_fl=None, tap=None):
    """IntModel.forward int branch: fix_resnet.py:354-383, fix_mobilenet_v2.py:209-241,
    fix_mobilenet_v1.py:122-147.  x: int32 [N,3,H,W] as produced by forward_loss
    (fix_train.py:683-692).  Returns float32 logits [N, classes] (`.float()` of int32, :383)."""
    x = _i32(x)
    # head: no requant of the input (caller already produced head-format integers)
    t, fl = _conv_layer(spec.head, params, x, x_fl, quant_input=False, tap=tap)
    if spec.head_maxpool:
        # float MaxPool detour (fix_resnet.py:358-359) is exact only below 2^24
        assert np.abs(t).max() < (1 << 24)
        t = maxpool(t, 3, 2, 1)
        if tap is not None:
            tap('head.maxpool', t, fl)
    for b in spec.blocks:
        t, fl = block_forward(b, params, t, fl, tap=tap)
    if spec.tail is not None:
        t, fl = _conv_layer(spec.tail, params, t, fl, tap=tap)
    t = avgpool_sum(t)
    fl += AVGPOOL_SHIFT
    assert fl <= 32   # fix_quant_ops.py:129
    k = spec.fc_key
    in_fl = int(params[k + '.input_fraclen'].reshape(-1)[0])
    t = requant(t, in_fl, fl, spec.fc_signed_in)
    if tap is not None:
        tap('fc.in', t, in_fl)
    logits = linear(t, params[k + '.weight'], params[k + '.bias'])
    if tap is not None:
        tap(k, logits, in_fl + int(params[k + '.weight_fraclen'].reshape(-1)[0]))
    return logits.astype(np.float32)


def quantize_input_u8(img01):
    """fix_train.py:689-692: (255*x).round_().int(), output_fraclen 8 (round half to even)."""
    return np.rint(255.0 * np.asarray(img01, dtype=np.float32)).astype(np.int32), 8


def quantize_input_pixels(u8, normalize=False, mean=None, std=None, head_in_fl=None, signed=True):
    """The decoder-side pipeline on uint8 pixels NCHW: transforms.ToTensor (`to(float32).div(255)`) [+ transforms.Normalize
    (`sub_(mean).div_(std)`)], fix_train.py:299-329, then the input quantisation of forward_loss, fix_train.py:683-692 — each
    statement in float32, one rounding per operation."""
    t = np.asarray(u8).astype(np.float32) / np.float32(255.0)
    if not normalize:
        return np.rint(np.float32(255.0) * t).astype(np.int32), 8
    m = np.asarray(mean, dtype=np.float32).reshape(1, -1, 1, 1)
    sd = np.asarray(std, dtype=np.float32).reshape(1, -1, 1, 1)
    t = (t - m) / sd
    v = np.rint(t * np.float32(2.0 ** head_in_fl))
    return (np.clip(v, -127, 127) if signed else np.clip(v, 0, 255)).astype(np.int32), head_in_fl


def topk_correct(logits, target, topk=(1, 5)):
    """fix_train.py:697-704: rows of per-sample flags `target in top-k`.  Equal logits rank by lower class index (what a
    stable descending sort gives; torch.topk leaves the order of ties unspecified, tie-free inputs agree exactly)."""
    logits = np.asarray(logits, dtype=np.float32)
    target = np.asarray(target).reshape(-1)
    lt = logits[np.arange(logits.shape[0]), target][:, None]
    idx = np.arange(logits.shape[1])[None, :]
    rank = ((logits > lt) | ((logits == lt) & (idx < target[:, None]))).sum(1)
    return np.stack([(rank < k).astype(np.float32) for k in topk], 0)


def quantize_input_normalized(x, head_in_fl: int):
    """fix_train.py:683-687 via fix_quant (fix_quant_ops.py:64-87): round(x*2^fl) clamp +-127, fl."""
    v = np.rint(np.asarray(x, dtype=np.float32) * np.float32(2.0 ** head_in_fl))
    return np.clip(v, -127, 127).astype(np.int32), head_in_fl


def graph_forward(ig, x, input_fraclen=None):
    """Interpreter for an imported ONNX program (f8net_amd.onnx_import.IntGraph) over the op-level functions above:
    the same walk IntModel.forward does, driven by the graph instead of the topology table."""
    V, layer = ig.solve_fraclens(input_fraclen)
    t = {}
    for i, o in enumerate(ig.ops):
        if o.kind == 'input':
            t[i] = _i32(x)
        elif o.kind in ('conv', 'linear'):
            s = t[o.src]
            if o.shift is not None:
                s = requant(s, layer[i][0], V[o.src], o.signed)
            if o.kind == 'conv':
                y = conv2d(s, o.weight, o.bias, o.stride, o.pad, o.groups)
            else:
                y = linear(s.reshape(s.shape[0], -1), o.weight, o.bias)
            t[i] = relu(y) if o.relu else y
        elif o.kind == 'add':
            y, fl = add_align(t[o.src], t[o.src2], V[o.src], V[o.src2])
            assert fl == V[i]
            t[i] = relu(y) if o.relu else y
        elif o.kind == 'maxpool':
            t[i] = maxpool(t[o.src], o.kernel, o.stride, o.pad)
        elif o.kind == 'avgpool':
            t[i] = avgpool_sum(t[o.src])
    y = t[ig.output]
    return y.astype(np.float32) if ig.output_float else y
