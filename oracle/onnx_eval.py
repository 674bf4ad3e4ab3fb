"""ONNX-semantics evaluator (numpy) for the opset-11 slice the int_op_only files use.  TEST INFRASTRUCTURE, like the rest
of oracle/: it lets the tests ask whether two ONNX files (the reference's traced one and the one
f8net_amd.onnx_export writes) denote the same function *as ONNX defines it* — integer Div truncating toward zero, Mod
with fmod=0 following the divisor's sign, Clip bounds as float32, MaxPool on float32.  Conv / Gemm reuse the C oracle's
wrapping int32 arithmetic."""
import numpy as np

from . import oracle

_NP = {1: np.float32, 6: np.int32, 7: np.int64, 9: np.bool_, 11: np.float64}


def _div(a, b):
    if np.issubdtype(np.result_type(a, b), np.integer):
        q = np.abs(a.astype(np.int64)) // np.abs(np.asarray(b).astype(np.int64))
        return (q * np.sign(a.astype(np.int64)) * np.sign(b)).astype(np.result_type(a, b))
    return a / b


def run(graph, x):
    """graph: f8net_amd.onnx_io.Graph; x: the single input.  Returns the single output."""
    v = dict(graph.initializers)
    v[graph.inputs[0][0]] = x
    for n in graph.nodes:
        i = [v[k] for k in n.inputs]
        a = n.attrs
        op = n.op
        if op == 'Constant':
            r = a['value']
        elif op == 'Identity':
            r = i[0]
        elif op == 'Cast':
            r = np.asarray(i[0]).astype(_NP[a['to']])
        elif op == 'Pow':
            r = np.power(i[0], i[1]).astype(np.asarray(i[0]).dtype)
        elif op == 'Add':
            with np.errstate(over='ignore'):
                r = i[0] + i[1]
        elif op == 'Mul':
            with np.errstate(over='ignore'):
                r = i[0] * i[1]
        elif op == 'Div':
            r = _div(i[0], i[1])
        elif op == 'Mod':
            assert a.get('fmod', 0) == 0
            r = np.mod(i[0], i[1])
        elif op == 'Equal':
            r = i[0] == i[1]
        elif op == 'Where':
            r = np.where(i[0], i[1], i[2])
        elif op == 'Clip':
            # torch's exporter leaves an int32 operand with float32 bounds; runtimes that accept it compare in float
            r = np.clip(i[0].astype(np.float64), float(i[1]), float(i[2]))
        elif op == 'Relu':
            r = np.maximum(i[0], 0)
        elif op == 'Conv':
            assert a['pads'][0] == a['pads'][2] and a['strides'][0] == a['strides'][1]
            r = oracle.conv2d(i[0].astype(np.int32), i[1], i[2] if len(i) > 2 else None, a['strides'][0], a['pads'][0],
                              a.get('group', 1))
        elif op == 'MaxPool':
            r = oracle.maxpool(i[0].astype(np.int32), a['kernel_shape'][0], a['strides'][0], a['pads'][0]).astype(np.float32)
        elif op == 'ReduceSum':
            r = i[0].sum(axis=tuple(a['axes']), keepdims=bool(a.get('keepdims', 1)))
        elif op == 'Shape':
            r = np.array(i[0].shape, np.int64)
        elif op == 'Gather':
            r = np.take(i[0], i[1], axis=a.get('axis', 0))
        elif op == 'Unsqueeze':
            r = np.expand_dims(i[0], tuple(a['axes']))
        elif op == 'Concat':
            r = np.concatenate([np.atleast_1d(t) for t in i], axis=a['axis'])
        elif op == 'Reshape':
            r = i[0].reshape([int(d) for d in i[1]])
        elif op == 'Gemm':
            assert a.get('transB', 0) == 1
            r = oracle.linear(i[0].astype(np.int32), i[1], i[2] if len(i) > 2 else None)
        else:
            raise NotImplementedError(op)
        v[n.outputs[0]] = r
    return v[graph.outputs[0][0]]
