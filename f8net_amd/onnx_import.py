"""Importer for the reference's `int_op_only_model.onnx`.

`torch.onnx.export` of the reference IntModel (/root/reference/myutils/export.py:4-31, fix_train.py:948-954) traces
the integer forward into opset-11 primitives: the fraction lengths are Python ints at trace time, so what the file holds
is the *arithmetic* — every `int_op_only_fix_quant` (fix_quant_ops.py:90-114) appears as

    Add(S, 2^(n-1));  Mod(S, 2^n) == 2^(n-1);  Div(., 2^(n+1)) * 2;  Div(., 2^n);  Where;  Clip(lo, hi)     (n > 0)
    Mul(S, 2^-n);  Clip(lo, hi)                                                                              (n <= 0)

every residual join (fix_resnet.py:40-54) as `Clip(Add(Mul(A, 2^k), B), -(2^31-1), 2^31-1)`, the max-pool detour as
Cast-MaxPool-Cast, FXQAvgPool2d as two ReduceSums, the classifier as Gemm — with the int32 weights as initializers named
by their state_dict keys.  ONNX's integer `Div` truncates where the model's `>>` floors, so the file is a lossy picture
of the network for negative values; this importer does not execute it, it recognises the templates and rebuilds the
*PyTorch* semantics the file was traced from — the graph of conv / add / pool / linear nodes libf8net.so plans.

The absolute fraction lengths are not in the file (only their differences are: the shifts).  `IntGraph.solve_fraclens`
picks the least assignment that satisfies the reference's asserts (0 <= input_fraclen <= 8, resp. 7 signed;
fix_quant_ops.py:91-98) — any assignment with the same shifts computes the same integers, and the logits the
reference returns are raw integers (`.float()` without rescaling, fix_resnet.py:383).
"""
from dataclasses import dataclass, field

import numpy as np

from . import onnx_io

I32_CLAMP = (1 << 31) - 1
AVGPOOL_SHIFT = 6      # FXQAvgPool2d(7).shiftnum, fix_quant_ops.py:121-122 (metadata only: absorbed by the next shift)


class OnnxImportError(ValueError):
    pass


# ------------------------------------------------------------------------------ the imported program

@dataclass
class IntOp:
    kind: str                      # input | conv | linear | add | maxpool | avgpool
    src: int = -1                  # tensor id (index into IntGraph.ops)
    src2: int = -1                 # add: the operand that is NOT shifted
    weight: np.ndarray = None
    bias: np.ndarray = None
    stride: int = 1
    pad: int = 0
    groups: int = 1
    kernel: int = 0
    shift: int = None              # conv/linear: requant n (None = input already in format); add: k of `src << k`
    signed: bool = False           # conv/linear: clamp range of the requant (+-127 vs 0..255)
    relu: bool = False
    swap: bool = False             # add: the shifted operand is the Add's second input (`res += x << k`) — cosmetic
    key: str = ''                  # state_dict key of the layer ('head.0', 'stage_0_layer_0.body.0', ...)
    shape: tuple = None            # input: (C, H, W)


@dataclass
class IntGraph:
    ops: list = field(default_factory=list)
    output: int = -1
    output_float: bool = True
    input_signed: bool = False

    def layer_keys(self):
        return [o.key for o in self.ops if o.kind in ('conv', 'linear')]

    # -- fraction lengths -------------------------------------------------------------------------------------------
    def solve_fraclens(self, input_fraclen=None):
        """Least fraction-length assignment consistent with the shifts.  Returns (V, layer) with V[t] = output_fraclen of
        tensor t and layer[t] = (input_fraclen, weight_fraclen) for conv / linear ops.

        Unknowns: V of every tensor.  With i_L = V[src_L] - n_L and w_L = V[out_L] - i_L the reference's invariants are
        difference constraints (0 <= i_L <= 8|7, 0 <= w_L <= 31, V[avgpool] <= 32, joins: V[b] = V[a] + k = V[out]);
        the least solution is the longest-path fixpoint from a zero node (Bellman-Ford)."""
        n = len(self.ops)
        ZERO = n
        edges = []                                   # (u, v, c): V[v] >= V[u] + c

        def ge(v, u, c):
            edges.append((u, v, c))

        def eq(v, u, c):                             # V[v] = V[u] + c
            ge(v, u, c)
            ge(u, v, -c)

        for t, o in enumerate(self.ops):
            if o.kind == 'input':
                if input_fraclen is None:
                    input_fraclen = 7 if self.input_signed else 8
                eq(t, ZERO, input_fraclen)
            elif o.kind in ('conv', 'linear'):
                lim = 7 if o.signed else 8
                if o.shift is None:                  # head conv: i = V[src]
                    ge(o.src, ZERO, 0)
                    ge(ZERO, o.src, -lim)
                    ge(t, o.src, 0)                  # w >= 0
                    ge(o.src, t, -31)                # w <= 31
                else:
                    ge(o.src, ZERO, o.shift)         # i >= 0
                    ge(ZERO, o.src, -(o.shift + lim))
                    ge(t, o.src, -o.shift)           # w = V[t] - V[src] + n >= 0
                    ge(o.src, t, o.shift - 31)
            elif o.kind == 'add':
                eq(o.src2, o.src, o.shift)
                eq(t, o.src2, 0)
            elif o.kind == 'maxpool':
                eq(t, o.src, 0)
            elif o.kind == 'avgpool':
                eq(t, o.src, AVGPOOL_SHIFT)
                ge(ZERO, t, -32)                     # fix_quant_ops.py:129
        V = [None] * (n + 1)
        V[ZERO] = 0
        for it in range(n + 2):
            changed = False
            for u, v, c in edges:
                if V[u] is not None and (V[v] is None or V[v] < V[u] + c):
                    V[v] = V[u] + c
                    changed = True
            if not changed:
                break
        else:
            raise OnnxImportError('no fraction-length assignment satisfies the shifts in this graph')
        if V[ZERO] != 0 or any(v is None for v in V):
            raise OnnxImportError('no fraction-length assignment satisfies the shifts in this graph')
        layer = {}
        for t, o in enumerate(self.ops):
            if o.kind in ('conv', 'linear'):
                i = V[o.src] - (o.shift or 0)
                layer[t] = (i, V[t] - i)
        return V[:n], layer

    def state_dict(self, input_fraclen=None):
        """Parameter dict keyed like the reference IntModel's state_dict (fix_quant_ops.py:705-713)."""
        _, layer = self.solve_fraclens(input_fraclen)
        sd = {}
        for t, o in enumerate(self.ops):
            if o.kind in ('conv', 'linear'):
                i, w = layer[t]
                sd[o.key + '.weight'] = o.weight
                sd[o.key + '.bias'] = o.bias
                sd[o.key + '.weight_fraclen'] = np.array(w, np.int32)
                sd[o.key + '.input_fraclen'] = np.array([i], np.int32)
        return sd

    # -- planning through the C ABI -----------------------------------------------------------------------------------
    def build_net(self, max_batch, hw=None, input_fraclen=None):
        """Record the graph through libf8net.so's builder and plan it (f8_net_finalize)."""
        from .net import F8Net
        V, layer = self.solve_fraclens(input_fraclen)
        net = F8Net()
        ids = {}
        for t, o in enumerate(self.ops):
            if o.kind == 'input':
                C, H, W = o.shape
                if hw is not None:
                    H = W = hw
                ids[t] = net.input(C, H, W, V[t])
            elif o.kind == 'conv':
                i, w = layer[t]
                ids[t] = net.conv(ids[o.src], o.weight, o.bias, stride=o.stride, pad=o.pad, groups=o.groups,
                                  weight_fl=w, input_fl=i, input_signed=o.signed, quant_input=o.shift is not None,
                                  relu=o.relu, label=o.key)
            elif o.kind == 'linear':
                i, w = layer[t]
                ids[t] = net.linear(ids[o.src], o.weight, o.bias, weight_fl=w, input_fl=i, input_signed=o.signed,
                                    quant_input=o.shift is not None, label=o.key)
            elif o.kind == 'add':
                ids[t] = net.add(ids[o.src], ids[o.src2], relu=o.relu)
            elif o.kind == 'maxpool':
                ids[t] = net.maxpool(ids[o.src], o.kernel, o.stride, o.pad)
            elif o.kind == 'avgpool':
                ids[t] = net.avgpool_sum(ids[o.src], AVGPOOL_SHIFT)
        net.output(ids[self.output], as_float=self.output_float)
        return net.finalize(max_batch)


# ------------------------------------------------------------------------------ template recognition

def _pow2(c, what):
    c = int(c)
    if c <= 0 or c & (c - 1):
        raise OnnxImportError(f'{what}: {c} is not a power of two')
    return c.bit_length() - 1


def _match_requant(e):
    """expr -> (source tensor id, n, signed) for the int_op_only_fix_quant templates, or None for a bare tensor."""
    if e[0] == 'T':
        return None
    if e[0] != 'clip':
        raise OnnxImportError(f'conv input is not a requantised tensor: {e[0]}')
    _, body, lo, hi = e
    if (lo, hi) == (-127, 127):
        signed = True
    elif (lo, hi) == (0, 255):
        signed = False
    else:
        raise OnnxImportError(f'requant clamp [{lo},{hi}] is neither [-127,127] nor [0,255]')
    if body[0] == 'mul' and body[1][0] == 'T':                      # n <= 0: input << -n (fix_quant_ops.py:105-106)
        return body[1][1], -_pow2(body[2], 'left shift'), signed
    if body[0] != 'where':
        raise OnnxImportError(f'unrecognised requant body {body[0]}')
    _, cond, tie, reg = body
    try:
        (_, (_, src_m, M), h) = cond                                # eq(mod(S, 2^n), 2^(n-1))
        assert cond[0] == 'eq' and cond[1][0] == 'mod'
        (_, (_, (_, src_t, h1), D1), two) = tie                     # mul(div(addc(S, h), 2^(n+1)), 2)
        assert tie[0] == 'mul' and tie[1][0] == 'div' and tie[1][1][0] == 'addc'
        (_, (_, src_r, h2), D2) = reg                               # div(addc(S, h), 2^n)
        assert reg[0] == 'div' and reg[1][0] == 'addc'
    except (ValueError, AssertionError, TypeError, IndexError):
        raise OnnxImportError('unrecognised round-half-even template') from None
    n = _pow2(M, 'requant modulus')
    if not (src_m == src_t == src_r and src_m[0] == 'T'):
        raise OnnxImportError('round-half-even template reads different tensors')
    if n < 1 or not (h == h1 == h2 == 1 << (n - 1) and D1 == 1 << (n + 1) and D2 == 1 << n and two == 2):
        raise OnnxImportError(f'round-half-even constants inconsistent for n={n}')
    return src_m[1], n, signed


def _layer_key(weight_name):
    if not weight_name.endswith('.weight'):
        raise OnnxImportError(f'initializer {weight_name!r} is not named like a state_dict weight')
    return weight_name[:-len('.weight')]


def import_graph(model, input_signed=False) -> IntGraph:
    """model: path / bytes of an ONNX file exported from a reference IntModel.  input_signed: FLAGS.normalize of the run
    that exported it (the head conv's input is signed 8-bit then, fix_train.py:683-687; nothing in the file says so)."""
    g = model if isinstance(model, onnx_io.Graph) else onnx_io.load_graph(model)
    if len(g.inputs) != 1 or len(g.outputs) != 1:
        raise OnnxImportError('expected one graph input and one output')
    iname, _, idims = g.inputs[0]
    if len(idims) != 4 or not all(isinstance(d, int) for d in idims[1:]):
        raise OnnxImportError(f'graph input dims {idims}: expected [batch, C, H, W]')
    uses = {}
    for nd in g.nodes:
        for s in nd.inputs:
            uses[s] = uses.get(s, 0) + 1
    ig = IntGraph(input_signed=bool(input_signed))
    ig.ops.append(IntOp('input', shape=tuple(idims[1:])))
    val = {iname: ('T', 0)}                          # ONNX value name -> expr | ('const', python number | ndarray)
    for k, a in g.initializers.items():
        val[k] = ('init', k)

    def const(name, what):
        v = val.get(name)
        if v is None or v[0] != 'const':
            raise OnnxImportError(f'{what}: operand {name!r} is not a constant')
        c = v[1]
        if isinstance(c, np.ndarray):
            if c.size != 1:
                raise OnnxImportError(f'{what}: constant {name!r} is not a scalar')
            c = c.reshape(-1)[0]
        f = float(c)
        if f != int(f):
            raise OnnxImportError(f'{what}: constant {f} is not an integer')
        return int(f)

    def init(name, what):
        v = val.get(name)
        if v is None or v[0] != 'init':
            raise OnnxImportError(f'{what}: {name!r} is not an initializer')
        a = g.initializers[v[1]]
        if a is None:
            raise OnnxImportError(f'{what}: initializer {v[1]!r} has no payload (skeleton file: fill_initializers first)')
        return v[1], a

    def is_const(name):
        return val.get(name, ('?',))[0] == 'const'

    def requant_src(name, what):
        e = val.get(name)
        if e is None or e[0] in ('const', 'init', 'opaque'):
            raise OnnxImportError(f'{what}: input {name!r} is not an activation')
        m = _match_requant(e)
        return (e[1], None, False) if m is None else m

    for nd in g.nodes:
        op, out = nd.op, nd.outputs[0]
        a = nd.attrs
        if op == 'Constant':
            val[out] = ('const', a['value'])
        elif op == 'Identity':
            val[out] = val[nd.inputs[0]]
        elif op == 'Cast':
            v = val[nd.inputs[0]]
            val[out] = v                                           # integer-valued throughout: casts carry no arithmetic
            if out == g.outputs[0][0]:
                ig.output_float = a['to'] == onnx_io.FLOAT
        elif op == 'Pow':
            val[out] = ('const', float(const(nd.inputs[0], 'Pow')) ** const(nd.inputs[1], 'Pow'))
        elif op in ('Shape', 'Gather', 'Unsqueeze', 'Concat'):
            val[out] = ('opaque',)                                 # the `x.view(x.size(0), -1)` plumbing before the FC
        elif op == 'Reshape':
            val[out] = val[nd.inputs[0]]
        elif op in ('Add', 'Mul', 'Div', 'Mod', 'Equal'):
            x, y = nd.inputs
            if op in ('Add', 'Mul') and is_const(x) and not is_const(y):
                x, y = y, x
            if is_const(x) and is_const(y):
                cx, cy = const(x, op), const(y, op)
                val[out] = ('const', {'Add': cx + cy, 'Mul': cx * cy}.get(op))
                if val[out][1] is None:
                    raise OnnxImportError(f'{op} of two constants')
            elif is_const(y):
                tag = {'Add': 'addc', 'Mul': 'mul', 'Div': 'div', 'Mod': 'mod', 'Equal': 'eq'}[op]
                val[out] = (tag, val[x], const(y, op))
            elif op == 'Add':
                val[out] = ('add', val[x], val[y])
            else:
                raise OnnxImportError(f'{op} of two activations')
        elif op == 'Where':
            val[out] = ('where', val[nd.inputs[0]], val[nd.inputs[1]], val[nd.inputs[2]])
        elif op == 'Clip':
            lo, hi = const(nd.inputs[1], 'Clip'), const(nd.inputs[2], 'Clip')
            e = val[nd.inputs[0]]
            if e[0] == 'add':                                      # residual join, fix_resnet.py:40-54
                # the exporter stores clamp_(min=-(2^31-1), max=2^31-1) as float32 constants, which round to -+2^31
                if -lo not in (I32_CLAMP, I32_CLAMP + 1) or hi not in (I32_CLAMP, I32_CLAMP + 1):
                    raise OnnxImportError(f'residual clamp [{lo},{hi}]')
                l, r = e[1], e[2]
                swap = l[0] != 'mul'
                if swap:
                    l, r = r, l
                if l[0] != 'mul' or l[1][0] != 'T' or r[0] != 'T':
                    raise OnnxImportError('residual join is not Add(Mul(A, 2^k), B)')
                ig.ops.append(IntOp('add', src=l[1][1], src2=r[1], shift=_pow2(l[2], 'residual shift'), swap=swap))
                val[out] = ('T', len(ig.ops) - 1)
            else:
                val[out] = ('clip', e, lo, hi)
        elif op == 'Relu':
            e = val[nd.inputs[0]]
            if e[0] != 'T' or ig.ops[e[1]].kind not in ('conv', 'add') or uses.get(nd.inputs[0], 0) != 1:
                raise OnnxImportError('Relu on something other than a single-use conv / join result')
            ig.ops[e[1]].relu = True                               # nn.ReLU(inplace=True) on the producer's tensor
            val[out] = e
        elif op == 'Conv':
            src, n, signed = requant_src(nd.inputs[0], 'Conv')
            wname, w = init(nd.inputs[1], 'Conv weight')
            b = init(nd.inputs[2], 'Conv bias')[1] if len(nd.inputs) > 2 else None
            k = a['kernel_shape']
            pads, st = a.get('pads', [0, 0, 0, 0]), a.get('strides', [1, 1])
            if k[0] != k[1] or len(set(pads)) != 1 or st[0] != st[1] or any(d != 1 for d in a.get('dilations', [1, 1])):
                raise OnnxImportError(f'Conv {wname}: only square kernels / symmetric pads / no dilation')
            ig.ops.append(IntOp('conv', src=src, weight=np.array(w, np.int32),
                                bias=None if b is None else np.array(b, np.int32), stride=st[0],
                                pad=pads[0], groups=a.get('group', 1), kernel=k[0], shift=n, signed=signed,
                                key=_layer_key(wname)))
            if n is None:
                ig.ops[-1].signed = ig.input_signed
            val[out] = ('T', len(ig.ops) - 1)
        elif op == 'Gemm':
            if a.get('transB', 0) != 1 or a.get('alpha', 1.0) != 1.0 or a.get('beta', 1.0) != 1.0:
                raise OnnxImportError('Gemm: expected x @ W^T + b')
            src, n, signed = requant_src(nd.inputs[0], 'Gemm')
            wname, w = init(nd.inputs[1], 'Gemm weight')
            b = init(nd.inputs[2], 'Gemm bias')[1] if len(nd.inputs) > 2 else None
            ig.ops.append(IntOp('linear', src=src, weight=np.array(w, np.int32),
                                bias=None if b is None else np.array(b, np.int32), shift=n, signed=signed,
                                key=_layer_key(wname)))
            val[out] = ('T', len(ig.ops) - 1)
        elif op == 'MaxPool':
            e = val[nd.inputs[0]]
            k, st, pads = a['kernel_shape'], a.get('strides', [1, 1]), a.get('pads', [0, 0, 0, 0])
            if e[0] != 'T' or k[0] != k[1] or st[0] != st[1] or len(set(pads)) != 1 or a.get('ceil_mode', 0):
                raise OnnxImportError('MaxPool: unsupported form')
            ig.ops.append(IntOp('maxpool', src=e[1], kernel=k[0], stride=st[0], pad=pads[0]))
            val[out] = ('T', len(ig.ops) - 1)
        elif op == 'ReduceSum':                                    # x.sum(-1).sum(-1), fix_quant_ops.py:130
            e = val[nd.inputs[0]]
            if a.get('axes') != [-1] or a.get('keepdims', 1) != 0:
                raise OnnxImportError('ReduceSum: expected axes=[-1], keepdims=0')
            if e[0] == 'T':
                val[out] = ('rsum', e)
            elif e[0] == 'rsum':
                ig.ops.append(IntOp('avgpool', src=e[1][1]))
                val[out] = ('T', len(ig.ops) - 1)
            else:
                raise OnnxImportError('ReduceSum: unsupported operand')
        else:
            raise OnnxImportError(f'unsupported ONNX op {op}')
    e = val.get(g.outputs[0][0])
    if e is None or e[0] != 'T':
        raise OnnxImportError('graph output is not a layer result')
    ig.output = e[1]
    return ig


def detect_arch(ig: IntGraph):
    """Name of the topology table entry whose layer keys this graph carries, or None."""
    from . import topology
    keys = sorted(ig.layer_keys())
    for arch in ('resnet18', 'resnet50', 'mobilenet_v1', 'mobilenet_v2'):
        if sorted(topology.get(arch).layer_keys()) == keys:
            return arch
    return None


def int_model_from_onnx(model, normalize=False, input_fraclen=None):
    """ONNX file -> our IntModel (the module tree with the reference's state_dict keys) for one of the four nets.
    normalize / input_fraclen: FLAGS.normalize of the exporting run and its head.input_fraclen — with normalize the
    float image is scaled by 2^head.input_fraclen before it enters the net (fix_train.py:683-687), and that scale is
    not in the file (default 7; without normalize the input is u8 at fraclen 8, fix_train.py:689-692)."""
    from . import topology
    from .int_model import from_params
    ig = import_graph(model, input_signed=normalize)
    arch = detect_arch(ig)
    if arch is None:
        raise OnnxImportError('layer keys match none of resnet18 / resnet50 / mobilenet_v1 / mobilenet_v2; use '
                              'import_graph(...).build_net(...) for a free-form graph')
    return from_params(topology.get(arch, normalize=normalize), ig.state_dict(input_fraclen))
