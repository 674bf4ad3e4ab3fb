"""Minimal ONNX (protobuf wire format) reader / writer — no `onnx` package needed.

The reference ships its integer models as `int_op_only_model.onnx`, written by
`torch.onnx.export(..., opset_version=11)` (/root/reference/myutils/export.py:4-31, called from
fix_train.py:948-954).  This module understands exactly the slice of onnx.proto those files use:

    ModelProto      ir_version=1 producer_name=2 producer_version=3 graph=7 opset_import=8
    OperatorSetId   domain=1 version=2
    GraphProto      node=1 name=2 initializer=5 input=11 output=12
    NodeProto       input=1 output=2 name=3 op_type=4 attribute=5
    AttributeProto  name=1 f=2 i=3 s=4 t=5 floats=7 ints=8 type=20
    TensorProto     dims=1 data_type=2 float_data=4 int32_data=5 int64_data=7 name=8 raw_data=9
    ValueInfoProto  name=1 type=2{tensor_type=1{elem_type=1 shape=2{dim=1{dim_value=1 dim_param=2}}}}

Messages are decoded into ordered `(field, wire_type, value)` lists, so anything this module does not
interpret survives a decode -> encode round trip byte for byte.
"""
import struct
from dataclasses import dataclass, field

import numpy as np

# TensorProto.DataType
FLOAT, UINT8, INT8, INT32, INT64, BOOL, DOUBLE = 1, 2, 3, 6, 7, 9, 11
_NP = {FLOAT: np.dtype('<f4'), UINT8: np.dtype('u1'), INT8: np.dtype('i1'), INT32: np.dtype('<i4'),
       INT64: np.dtype('<i8'), BOOL: np.dtype('?'), DOUBLE: np.dtype('<f8')}
_DT = {v: k for k, v in _NP.items()}
# AttributeProto.AttributeType
A_FLOAT, A_INT, A_STRING, A_TENSOR, A_FLOATS, A_INTS = 1, 2, 3, 4, 6, 7


# ------------------------------------------------------------------------------ wire format

def _varint(b, i):
    r = s = 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        s += 7
        if c < 0x80:
            return r, i


def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        c = v & 0x7F
        v >>= 7
        if v:
            out.append(c | 0x80)
        else:
            out.append(c)
            return bytes(out)


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def decode(b):
    """bytes -> [(field, wire_type, value)]; value is an int (wire 0) or a memoryview (wire 1, 2, 5)."""
    b = memoryview(b)
    i, n, out = 0, len(b), []
    while i < n:
        key, i = _varint(b, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(b, i)
        elif wt == 1:
            v = b[i:i + 8]
            i += 8
        elif wt == 2:
            ln, i = _varint(b, i)
            v = b[i:i + ln]
            i += ln
        elif wt == 5:
            v = b[i:i + 4]
            i += 4
        else:
            raise ValueError(f'onnx_io: unsupported wire type {wt} at byte {i}')
        out.append((f, wt, v))
    if i != n:
        raise ValueError('onnx_io: truncated message')
    return out


def encode(fields):
    out = bytearray()
    for f, wt, v in fields:
        out += _enc_varint((f << 3) | wt)
        if wt == 0:
            out += _enc_varint(v)
        elif wt == 2:
            out += _enc_varint(len(v))
            out += v
        else:
            out += v
    return bytes(out)


def _packed_varints(wt, v):
    if wt == 0:
        return [_signed(v)]
    out, i = [], 0
    while i < len(v):
        x, i = _varint(v, i)
        out.append(_signed(x))
    return out


# ------------------------------------------------------------------------------ typed view

@dataclass
class Node:
    op: str
    inputs: list
    outputs: list
    name: str = ''
    attrs: dict = field(default_factory=dict)


@dataclass
class Graph:
    nodes: list
    initializers: dict          # name -> np.ndarray
    inputs: list                # [(name, elem_type, dims)] with dims entries int or str (symbolic)
    outputs: list
    opset: int = 0
    producer: str = ''
    name: str = ''
    init_dims: dict = field(default_factory=dict)   # name -> dims (also for payload-less skeleton initializers)


def _tensor(buf):
    dims, dt, name, raw = [], None, '', None
    f32, i32, i64 = [], [], []
    for f, wt, v in decode(buf):
        if f == 1:
            dims += _packed_varints(wt, v)
        elif f == 2:
            dt = v
        elif f == 8:
            name = bytes(v).decode()
        elif f == 9:
            raw = v
        elif f == 4:
            f32 += list(np.frombuffer(v, '<f4')) if wt == 2 else [struct.unpack('<f', v)[0]]
        elif f == 5:
            i32 += _packed_varints(wt, v)
        elif f == 7:
            i64 += _packed_varints(wt, v)
    if dt not in _NP:
        raise ValueError(f'onnx_io: tensor {name!r}: unsupported data_type {dt}')
    if raw is not None:
        a = np.frombuffer(raw, _NP[dt])
    elif dt == FLOAT:
        a = np.asarray(f32, '<f4')
    elif dt == INT64:
        a = np.asarray(i64, '<i8')
    else:
        a = np.asarray(i32, '<i8').astype(_NP[dt])
    want = int(np.prod(dims)) if dims else 1
    if a.size != want:
        if a.size == 0:
            a = None                 # skeleton: payload stripped (see strip_initializers)
        else:
            raise ValueError(f'onnx_io: tensor {name!r}: {a.size} elements for dims {dims}')
    return name, (a.reshape(dims) if a is not None else None), dims, dt


def _attr(buf):
    name, val, ints, floats = '', None, [], []
    typ = 0
    for f, wt, v in decode(buf):
        if f == 1:
            name = bytes(v).decode()
        elif f == 2:
            val = struct.unpack('<f', v)[0]
        elif f == 3:
            val = _signed(v)
        elif f == 4:
            val = bytes(v)
        elif f == 5:
            val = _tensor(v)[1]
        elif f == 7:
            floats += list(np.frombuffer(v, '<f4')) if wt == 2 else [struct.unpack('<f', v)[0]]
        elif f == 8:
            ints += _packed_varints(wt, v)
        elif f == 20:
            typ = v
    if typ == A_INTS or (val is None and ints):
        val = ints
    elif typ == A_FLOATS or (val is None and floats):
        val = floats
    return name, val


def _node(buf):
    n = Node('', [], [])
    for f, wt, v in decode(buf):
        if f == 1:
            n.inputs.append(bytes(v).decode())
        elif f == 2:
            n.outputs.append(bytes(v).decode())
        elif f == 3:
            n.name = bytes(v).decode()
        elif f == 4:
            n.op = bytes(v).decode()
        elif f == 5:
            k, a = _attr(v)
            n.attrs[k] = a
    return n


def _value_info(buf):
    name, et, dims = '', 0, []
    for f, wt, v in decode(buf):
        if f == 1:
            name = bytes(v).decode()
        elif f == 2:
            for f2, _, v2 in decode(v):
                if f2 != 1:
                    continue
                for f3, _, v3 in decode(v2):
                    if f3 == 1:
                        et = v3
                    elif f3 == 2:
                        for f4, _, v4 in decode(v3):
                            if f4 != 1:
                                continue
                            d = '?'
                            for f5, _, v5 in decode(v4):
                                if f5 == 1:
                                    d = _signed(v5)
                                elif f5 == 2:
                                    d = bytes(v5).decode()
                            dims.append(d)
    return name, et, dims


def load_graph(model) -> Graph:
    """model: path or bytes of a serialized ModelProto."""
    if isinstance(model, str) or hasattr(model, '__fspath__'):
        with open(model, 'rb') as fh:
            model = fh.read()
    g = Graph([], {}, [], [])
    gbuf = None
    for f, wt, v in decode(model):
        if f == 7:
            gbuf = v
        elif f == 2:
            g.producer = bytes(v).decode()
        elif f == 8:
            for f2, _, v2 in decode(v):
                if f2 == 2:
                    g.opset = max(g.opset, v2)
    if gbuf is None:
        raise ValueError('onnx_io: no graph in model')
    for f, wt, v in decode(gbuf):
        if f == 1:
            g.nodes.append(_node(v))
        elif f == 2:
            g.name = bytes(v).decode()
        elif f == 5:
            name, a, dims, _ = _tensor(v)
            g.initializers[name] = a
            g.init_dims[name] = list(dims)
        elif f == 11:
            g.inputs.append(_value_info(v))
        elif f == 12:
            g.outputs.append(_value_info(v))
    g.inputs = [i for i in g.inputs if i[0] not in g.initializers]
    return g


# ------------------------------------------------------------------------------ writing

def tensor_proto(name, a, raw=True):
    a = np.asarray(a)                        # (ascontiguousarray would turn a 0-d scalar into shape [1])
    dt = _DT.get(a.dtype.newbyteorder('<') if a.dtype.byteorder == '>' else a.dtype)
    if dt is None:
        raise ValueError(f'onnx_io: cannot serialise dtype {a.dtype}')
    fields = [(1, 0, int(d)) for d in a.shape] + [(2, 0, dt)]
    if name:
        fields.append((8, 2, name.encode()))
    if raw:
        fields.append((9, 2, a.astype(_NP[dt]).tobytes()))
    return encode(fields)


def attr_proto(name, v):
    fields = [(1, 2, name.encode())]
    if isinstance(v, float):
        fields += [(2, 5, struct.pack('<f', v)), (20, 0, A_FLOAT)]
    elif isinstance(v, (int, np.integer)):
        fields += [(3, 0, int(v)), (20, 0, A_INT)]
    elif isinstance(v, bytes):
        fields += [(4, 2, v), (20, 0, A_STRING)]
    elif isinstance(v, np.ndarray):
        fields += [(5, 2, tensor_proto('', v)), (20, 0, A_TENSOR)]
    elif isinstance(v, (list, tuple)):
        fields += [(8, 0, int(x)) for x in v] + [(20, 0, A_INTS)]
    else:
        raise TypeError(f'onnx_io: attribute {name}: {type(v)}')
    return encode(fields)


def node_proto(n: Node):
    fields = [(1, 2, s.encode()) for s in n.inputs] + [(2, 2, s.encode()) for s in n.outputs]
    if n.name:
        fields.append((3, 2, n.name.encode()))
    fields.append((4, 2, n.op.encode()))
    fields += [(5, 2, attr_proto(k, v)) for k, v in n.attrs.items()]
    return encode(fields)


def value_info_proto(name, elem_type, dims):
    dl = []
    for d in dims:
        dl.append((1, 2, encode([(2, 2, d.encode())] if isinstance(d, str) else [(1, 0, int(d))])))
    tt = encode([(1, 0, elem_type), (2, 2, encode(dl))])
    return encode([(1, 2, name.encode()), (2, 2, encode([(1, 2, tt)]))])


def model_proto(g: Graph, ir_version=6, producer_version=''):
    gf = [(1, 2, node_proto(n)) for n in g.nodes]
    gf.append((2, 2, (g.name or 'main_graph').encode()))
    gf += [(5, 2, tensor_proto(k, v)) for k, v in g.initializers.items()]
    gf += [(11, 2, value_info_proto(*i)) for i in g.inputs]
    gf += [(12, 2, value_info_proto(*o)) for o in g.outputs]
    mf = [(1, 0, ir_version), (2, 2, (g.producer or 'f8net_amd').encode())]
    if producer_version:
        mf.append((3, 2, producer_version.encode()))
    mf.append((7, 2, encode(gf)))
    mf.append((8, 2, encode([(2, 0, g.opset or 11)])))
    return encode(mf)


# ------------------------------------------------------------------------------ skeletons
# A full-size ONNX file of these nets is tens of MB of int32 weights.  Test fixtures keep the graph and drop the
# payload of the large initializers (dims / dtype / name stay); `fill_initializers` puts payloads back.

def _map_initializers(model, fn):
    out = []
    for f, wt, v in decode(model):
        if f != 7:
            out.append((f, wt, bytes(v) if wt != 0 else v))
            continue
        gf = []
        for f2, wt2, v2 in decode(v):
            if f2 == 5:
                v2 = fn(v2)
            gf.append((f2, wt2, bytes(v2) if wt2 != 0 else v2))
        out.append((7, 2, encode(gf)))
    return encode(out)


def strip_initializers(model, min_bytes=4096):
    """Returns (skeleton_bytes, [names of stripped initializers])."""
    names = []

    def fn(buf):
        fl = decode(buf)
        raw = [v for f, _, v in fl if f == 9]
        if not raw or len(raw[0]) < min_bytes:
            return buf
        names.append(bytes([v for f, _, v in fl if f == 8][0]).decode())
        return encode([(f, wt, bytes(v) if wt != 0 else v) for f, wt, v in fl if f != 9])
    return _map_initializers(model, fn), names


def fill_initializers(skeleton, arrays: dict):
    def fn(buf):
        fl = decode(buf)
        name = bytes([v for f, _, v in fl if f == 8][0]).decode()
        if name not in arrays or any(f == 9 for f, _, _ in fl):
            return buf
        dt = [v for f, _, v in fl if f == 2][0]
        dims = [d for f, wt, v in fl if f == 1 for d in _packed_varints(wt, v)]
        a = np.ascontiguousarray(arrays[name]).astype(_NP[dt])
        if list(a.shape) != dims:
            raise ValueError(f'onnx_io: initializer {name}: shape {a.shape} != {dims}')
        return encode([(f, wt, bytes(v) if wt != 0 else v) for f, wt, v in fl] + [(9, 2, a.tobytes())])
    return _map_initializers(skeleton, fn)
