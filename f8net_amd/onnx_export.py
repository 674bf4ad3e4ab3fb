"""Writer for `int_op_only_model.onnx` files, the counterpart of the reference's
`onnx_export` (/root/reference/myutils/export.py:4-31, called from fix_train.py:948-954).

The reference gets its file by tracing the PyTorch IntModel with `torch.onnx.export(opset_version=11)`; here the same
opset-11 vocabulary is written directly from the integer graph — per op the node sequence the tracer produces:

    int_op_only_fix_quant, n > 0   Add, Mod(fmod=0), Cast(i64), Equal, Pow/Cast/Div, Pow/Cast/Mul, Pow/Cast/Div, Where,
                                   Clip(f32 bounds), Cast(i32)                      (fix_quant_ops.py:99-112)
    int_op_only_fix_quant, n <= 0  Pow/Cast/Mul, Clip, Cast(i32)                    (fix_quant_ops.py:105-112)
    conv / ReLU                    Conv(int32 W, B initializers), Relu, Cast(i32)
    residual join                  Pow/Cast/Mul, Add, Clip(-+2^31 as f32), Cast(i32) [, Relu, Cast]   (fix_resnet.py:40-54,77)
    head max-pool                  Cast(f32), MaxPool, Cast(i32)                     (fix_resnet.py:358-359)
    FXQAvgPool2d                   Cast(i64), ReduceSum(-1) x2, Cast(i32)            (fix_quant_ops.py:130-133)
    x.view(x.size(0), -1)          Shape, Gather, Unsqueeze, Concat, Reshape         (fix_resnet.py:371)
    classifier                     Gemm(transB=1), Cast(f32)                         (fix_resnet.py:383)

Input `input` int32 [batch_size, C, H, W], output `output` float32 [batch_size, classes], dynamic batch axis, initializers
named by state_dict key — as in the reference's files, including the tracer's merging of the identical arithmetic when
two requants of the max-pool result share a shift (ResNet-50 `stage_0_layer_0`: body.0 and shortcut.0).  Node names differ.
"""
import numpy as np

from . import onnx_io, topology
from .onnx_import import AVGPOOL_SHIFT, IntGraph, IntOp
from .onnx_io import Node


def _fl(params, key, name):
    return int(np.asarray(params[f'{key}.{name}']).reshape(-1)[0])


def graph_from_params(spec: topology.NetSpec, params: dict, hw: int = 224) -> IntGraph:
    """The walk IntModel.forward does (fix_resnet.py:354-383, fix_mobilenet_v2.py:209-241, fix_mobilenet_v1.py:122-147)
    recorded as an IntGraph: shifts from the exported fraction lengths."""
    params = {k: (v.detach().cpu().numpy() if hasattr(v, 'detach') else np.asarray(v)) for k, v in params.items()}
    ig = IntGraph(input_signed=spec.normalize)
    ig.ops.append(IntOp('input', shape=(3, hw, hw)))
    fl = {0: _fl(params, spec.head.key, 'input_fraclen')}

    def conv(src, c, quant=True):
        i, w = _fl(params, c.key, 'input_fraclen'), _fl(params, c.key, 'weight_fraclen')
        ig.ops.append(IntOp('conv', src=src, weight=params[c.key + '.weight'].astype(np.int32),
                            bias=params[c.key + '.bias'].astype(np.int32), stride=c.stride, pad=c.pad, groups=c.groups,
                            kernel=c.k, shift=(fl[src] - i) if quant else None, signed=c.signed_in, relu=c.relu,
                            key=c.key))
        fl[len(ig.ops) - 1] = i + w
        return len(ig.ops) - 1

    def join(res, x, relu, ge):
        a, b = fl[res], fl[x]
        if a > b or (ge and a == b):                                # x << (a - b)   fix_resnet.py:43-44 / mbv2 :37-38
            op = IntOp('add', src=x, src2=res, shift=a - b, relu=relu, swap=True)
        else:                                                       # res << (b - a)
            op = IntOp('add', src=res, src2=x, shift=b - a, relu=relu)
        ig.ops.append(op)
        fl[len(ig.ops) - 1] = max(a, b)
        return len(ig.ops) - 1

    t = conv(0, spec.head, quant=False)
    if spec.head_maxpool:
        ig.ops.append(IntOp('maxpool', src=t, kernel=3, stride=2, pad=1))
        fl[len(ig.ops) - 1] = fl[t]
        t = len(ig.ops) - 1
    ge = spec.arch.startswith('mobilenet')
    for b in spec.blocks:
        x = r = t
        for c in b.body:
            r = conv(r, c)
        if b.shortcut is not None:
            r = join(r, conv(x, b.shortcut), b.post_relu, ge)
        elif b.residual:
            r = join(r, x, b.post_relu, ge)
        t = r
    if spec.tail is not None:
        t = conv(t, spec.tail)
    ig.ops.append(IntOp('avgpool', src=t))
    fl[len(ig.ops) - 1] = fl[t] + AVGPOOL_SHIFT
    t = len(ig.ops) - 1
    k = spec.fc_key
    ig.ops.append(IntOp('linear', src=t, weight=params[k + '.weight'].astype(np.int32),
                        bias=params[k + '.bias'].astype(np.int32), shift=fl[t] - _fl(params, k, 'input_fraclen'),
                        signed=spec.fc_signed_in, key=k))
    ig.output = len(ig.ops) - 1
    ig.output_float = True
    return ig


class _Writer:
    def __init__(self):
        self.nodes = []
        self.n = 0

    def emit(self, op, inputs, attrs=None):
        self.n += 1
        out = f'/{op}_{self.n}_output_0'
        self.nodes.append(Node(op, list(inputs), [out], name=f'/{op}_{self.n}', attrs=dict(attrs or {})))
        return out

    def const(self, value, dtype):
        return self.emit('Constant', [], {'value': np.array(value, dtype)})

    def cast(self, x, to):
        return self.emit('Cast', [x], {'to': to})

    def pow2(self, e):                                               # `1 << e` traced as int(2.0 ** e)
        e = self.const(float(e), np.float32)
        return self.cast(self.emit('Pow', [self.const(2.0, np.float32), e]), onnx_io.INT32)


def export_graph(ig: IntGraph) -> bytes:
    """IntGraph -> serialized ModelProto."""
    w = _Writer()
    inits = {}
    name = {}
    shared = {}
    for t, o in enumerate(ig.ops):
        if o.kind == 'input':
            name[t] = 'input'
            in_dims = ['batch_size'] + list(o.shape)
            continue
        if o.kind in ('conv', 'linear'):
            s = name[o.src]
            if o.kind == 'linear':                                   # x.view(x.size(0), -1), fix_resnet.py:371
                b0 = w.emit('Gather', [w.emit('Shape', [s]), w.const(0, np.int64)], {'axis': 0})
                cat = w.emit('Concat', [w.emit('Unsqueeze', [b0], {'axes': [0]}), w.const([-1], np.int64)], {'axis': 0})
                s = w.emit('Reshape', [s, cat])
            if o.shift is not None:
                n = o.shift
                lo, hi = (-127.0, 127.0) if o.signed else (0.0, 255.0)
                if (s, n) in shared and ig.ops[o.src].kind == 'maxpool':
                    # the tracer merges the identical arithmetic of two requants of one tensor (body.0 / shortcut.0 at
                    # equal input_fraclen) — but only where the tensor has no in-place writers: the `.int()` of the
                    # max-pool detour, not the clamp_/ReLU(inplace) results that open later stages.  Where / Clip stay.
                    e, tie, reg = shared[(s, n)]
                    v = w.emit('Where', [e, tie, reg]) if n > 0 else tie
                elif n > 0:
                    h = 1 << (n - 1)
                    a = w.emit('Add', [s, w.const(h, np.int32)])
                    m = w.emit('Mod', [s, w.const(1 << n, np.int32)], {'fmod': 0})
                    e = w.emit('Equal', [w.cast(m, onnx_io.INT64), w.const(h, np.int64)])
                    tie = w.emit('Mul', [w.emit('Div', [a, w.pow2(n + 1)]), w.pow2(1)])
                    reg = w.emit('Div', [a, w.pow2(n)])
                    shared[(s, n)] = (e, tie, reg)
                    v = w.emit('Where', [e, tie, reg])
                else:
                    v = w.emit('Mul', [s, w.pow2(-n)])
                    shared[(s, n)] = (None, v, None)
                v = w.emit('Clip', [v, w.const(lo, np.float32), w.const(hi, np.float32)])
                s = w.cast(v, onnx_io.INT32)
            inits[o.key + '.weight'] = o.weight
            ins = [s, o.key + '.weight']
            if o.bias is not None:
                inits[o.key + '.bias'] = o.bias
                ins.append(o.key + '.bias')
            if o.kind == 'conv':
                y = w.emit('Conv', ins, {'dilations': [1, 1], 'group': int(o.groups), 'kernel_shape': [o.kernel] * 2,
                                         'pads': [o.pad] * 4, 'strides': [o.stride] * 2})
            else:
                y = w.emit('Gemm', ins, {'alpha': 1.0, 'beta': 1.0, 'transB': 1})
        elif o.kind == 'add':
            m = w.emit('Mul', [name[o.src], w.pow2(o.shift)])
            v = w.emit('Add', [name[o.src2], m] if o.swap else [m, name[o.src2]])   # `res += x`: res first
            v = w.emit('Clip', [v, w.const(-2147483647.0, np.float32), w.const(2147483647.0, np.float32)])
            y = w.cast(v, onnx_io.INT32)
        elif o.kind == 'maxpool':
            v = w.emit('MaxPool', [w.cast(name[o.src], onnx_io.FLOAT)],
                       {'ceil_mode': 0, 'dilations': [1, 1], 'kernel_shape': [o.kernel] * 2, 'pads': [o.pad] * 4,
                        'strides': [o.stride] * 2})
            y = w.cast(v, onnx_io.INT32)
        elif o.kind == 'avgpool':
            v = w.cast(name[o.src], onnx_io.INT64)
            v = w.emit('ReduceSum', [v], {'axes': [-1], 'keepdims': 0})
            v = w.emit('ReduceSum', [v], {'axes': [-1], 'keepdims': 0})
            y = w.cast(v, onnx_io.INT32)
        else:
            raise ValueError(o.kind)
        if o.relu:
            y = w.cast(w.emit('Relu', [y]), onnx_io.INT32)
        name[t] = y
    w.cast(name[ig.output], onnx_io.FLOAT if ig.output_float else onnx_io.INT32)
    w.nodes[-1].outputs[0] = 'output'
    classes = ig.ops[ig.output].weight.shape[0] if ig.ops[ig.output].kind == 'linear' else '?'
    g = onnx_io.Graph(w.nodes, inits, [('input', onnx_io.INT32, in_dims)],
                      [('output', onnx_io.FLOAT if ig.output_float else onnx_io.INT32, ['batch_size', classes])],
                      opset=11, producer='f8net_amd', name='main_graph')
    return onnx_io.model_proto(g)


def onnx_export(model, data_shape, output_file):
    """Same call shape as the reference's `onnx_export(model, data_shape, data_dtype, device, output_file)`
    (myutils/export.py:4) minus what a tracer needs: `model` is our IntModel, data_shape = [1, 3, H, W]."""
    ig = graph_from_params(model.spec, model.state_dict(), hw=int(data_shape[2]))
    data = export_graph(ig)
    with open(output_file, 'wb') as fh:
        fh.write(data)
    return len(data)
