"""`int_op_only_model.onnx` import / export.

Fixtures (`oracle/gen_golden.py --child onnx:<net>`): the reference IntModel, holding `synth.make_params(seed=4321)`,
exported by torch exactly as /root/reference/myutils/export.py:4-31 does it; kept are the file minus the payload of its
large initializers (regenerated here from the same seed and put back with `fill_initializers`) and the logits the PyTorch
IntModel returned for a 2-image batch.  `resnet18_lshift` uses fraction lengths that force left-shift requants (n < 0).

  importer: file -> integer program -> the oracle reproduces the PyTorch logits bit for bit (as a free-form graph and
            through the IntModel-keyed state_dict);
  exporter: topology + parameters -> file whose nodes equal the reference's file node for node (ops, wiring, attributes,
            constants, initializers; only value names differ) and which denotes the same function under ONNX semantics.
"""
import os

import numpy as np
import pytest

from f8net_amd import onnx_export, onnx_import, onnx_io, synth, topology
from oracle import onnx_eval, oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = ['resnet18', 'resnet18_lshift', 'resnet50', 'mobilenet_v1', 'mobilenet_v2']
LSHIFT = {'stage_0_layer_0.body.0': (6, 0), 'stage_0_layer_0.body.2': (8, 5),
          'stage_1_layer_0.body.0': (5, 1), 'stage_1_layer_0.body.2': (7, 6)}


def load_case(case):
    z = np.load(os.path.join(GOLD, f'onnx_{case}.npz'))
    arch = case.replace('_lshift', '')
    spec = topology.get(arch, normalize=bool(z['normalize']))
    fr = topology.R50_NVIDIA_FRACLENS if arch == 'resnet50' else (LSHIFT if case.endswith('_lshift') else None)
    params = synth.make_params(spec, seed=int(z['seed']), fraclens=fr)
    full = onnx_io.fill_initializers(z['skeleton'].tobytes(), {str(k): params[str(k)] for k in z['stripped']})
    assert len(full) == int(z['full_bytes'])
    x, x_fl = synth.make_input(spec, params, 2, int(z['hw']), seed=int(z['input_seed']))
    return z, spec, params, full, x, x_fl


@pytest.fixture(scope='module', autouse=True)
def _build_oracle():
    oracle.build()


def test_wire_format_round_trips_byte_for_byte():
    z = np.load(os.path.join(GOLD, 'onnx_resnet18.npz'))
    sk = z['skeleton'].tobytes()

    def rec(buf, depth):
        fl = onnx_io.decode(buf)
        assert onnx_io.encode([(f, wt, bytes(v) if wt else v) for f, wt, v in fl]) == bytes(buf)
        return fl
    top = rec(sk, 0)
    graph = [v for f, _, v in top if f == 7][0]
    for f, wt, v in rec(graph, 1):
        if f == 1:
            rec(v, 2)
    g = onnx_io.load_graph(sk)
    assert g.opset == 11 and g.producer == 'pytorch'
    assert g.inputs == [('input', onnx_io.INT32, ['batch_size', 3, 64, 64])]
    assert g.outputs == [('output', onnx_io.FLOAT, ['batch_size', 1000])]
    stripped = [str(k) for k in z['stripped']]
    assert all(g.initializers[k] is None for k in stripped)
    # strip(fill(skeleton)) == skeleton
    filled = onnx_io.fill_initializers(sk, {k: np.zeros(g.init_dims[k], np.int32) for k in stripped})
    sk2, names = onnx_io.strip_initializers(filled, 4096)
    assert sk2 == sk and names == stripped


@pytest.mark.parametrize('case', CASES)
def test_import_reproduces_the_pytorch_logits(case):
    z, spec, params, full, x, x_fl = load_case(case)
    ig = onnx_import.import_graph(full, input_signed=spec.normalize)
    assert onnx_import.detect_arch(ig) == spec.arch
    assert sorted(ig.layer_keys()) == sorted(spec.layer_keys())
    # as a free-form graph
    np.testing.assert_array_equal(oracle.graph_forward(ig, x, x_fl), z['logits'])
    # through the IntModel-keyed state_dict: weights / biases verbatim, fraction lengths equivalent (same shifts)
    sd = ig.state_dict(x_fl)
    for k in spec.layer_keys():
        np.testing.assert_array_equal(sd[k + '.weight'], params[k + '.weight'])
        np.testing.assert_array_equal(sd[k + '.bias'], params[k + '.bias'])
        i, w = int(sd[k + '.input_fraclen'][0]), int(sd[k + '.weight_fraclen'])
        assert 0 <= i <= 8 and 0 <= w <= 31
    np.testing.assert_array_equal(oracle.net_forward(spec, sd, x, x_fl), z['logits'])
    # the shifts are the ones the true fraction lengths imply
    want = onnx_export.graph_from_params(spec, params, hw=int(z['hw']))
    assert [(o.kind, o.src, o.src2, o.shift, o.signed, o.relu, o.swap, o.key) for o in ig.ops] == \
        [(o.kind, o.src, o.src2, o.shift, o.signed, o.relu, o.swap, o.key) for o in want.ops]
    if case == 'resnet18_lshift':
        assert min(o.shift for o in ig.ops if o.kind == 'conv' and o.shift is not None) == -2


def _canon(g):
    ids = {g.inputs[0][0]: 'input'}
    for k in g.initializers:
        ids[k] = 'init:' + k
    out = []
    for n in g.nodes:
        attrs = tuple(sorted((k, (str(v.dtype), v.shape, v.tobytes()) if isinstance(v, np.ndarray) else
                              (tuple(v) if isinstance(v, list) else v)) for k, v in n.attrs.items()))
        out.append((n.op, tuple(ids[i] for i in n.inputs), attrs))
        for o in n.outputs:
            ids[o] = len(ids)
    return out, ids[g.outputs[0][0]]


@pytest.mark.parametrize('case', CASES)
def test_export_writes_the_file_the_reference_writes(case):
    z, spec, params, full, x, x_fl = load_case(case)
    ref = onnx_io.load_graph(full)
    mine = onnx_io.load_graph(onnx_export.export_graph(onnx_export.graph_from_params(spec, params, hw=int(z['hw']))))
    assert (mine.opset, mine.inputs, mine.outputs) == (ref.opset, ref.inputs, ref.outputs)
    assert sorted(mine.initializers) == sorted(ref.initializers)
    for k, v in ref.initializers.items():
        assert mine.initializers[k].dtype == v.dtype
        np.testing.assert_array_equal(mine.initializers[k], v, err_msg=k)
    assert _canon(mine) == _canon(ref)
    # and, independently of the structural check: same function under ONNX's own semantics
    y_ref, y_mine = onnx_eval.run(ref, x), onnx_eval.run(mine, x)
    np.testing.assert_array_equal(y_mine, y_ref)
    if not spec.arch == 'mobilenet_v2':
        # ResNets / MBV1 requantise non-negative tensors only, where ONNX Div (truncating) and `>>` (flooring) agree;
        # MobileNet-V2's linear bottlenecks feed negatives, so its ONNX file is NOT the network (see onnx_import.py)
        np.testing.assert_array_equal(y_ref, z['logits'])
    else:
        assert not np.array_equal(y_ref, z['logits'])


def test_export_import_round_trip_of_our_int_model(tmp_path):
    torch = pytest.importorskip('torch')
    from f8net_amd import int_model
    spec = topology.get('mobilenet_v1')
    params = synth.make_params(spec, seed=99)
    m = int_model.from_params(spec, params)
    path = str(tmp_path / 'int_op_only_model.onnx')
    n = onnx_export.onnx_export(m, [1, 3, 224, 224], path)        # myutils/export.py:4 call shape
    assert os.path.getsize(path) == n
    m2 = onnx_import.int_model_from_onnx(path)
    assert m2.spec.arch == 'mobilenet_v1'
    sd, sd2 = m.state_dict(), m2.state_dict()
    for k in spec.layer_keys():
        assert torch.equal(sd[k + '.weight'], sd2[k + '.weight']) and torch.equal(sd[k + '.bias'], sd2[k + '.bias'])
    x, x_fl = synth.make_input(spec, params, 1, 64, seed=3)
    p2 = {k: v.numpy() for k, v in sd2.items()}
    np.testing.assert_array_equal(oracle.net_forward(spec, p2, x, x_fl), oracle.net_forward(spec, params, x, x_fl))


def test_importer_refuses_what_it_does_not_recognise():
    z, spec, params, full, x, x_fl = load_case('resnet18')
    g = onnx_io.load_graph(full)
    # a requant whose rounding constant is off by one is not int_op_only_fix_quant
    bad = onnx_io.load_graph(full)
    c = next(n for n in bad.nodes if n.op == 'Constant' and n.attrs['value'].dtype == np.int32)
    c.attrs['value'] = np.array(int(c.attrs['value']) + 1, np.int32)
    with pytest.raises(onnx_import.OnnxImportError):
        onnx_import.import_graph(bad)
    bad = onnx_io.load_graph(full)
    next(n for n in bad.nodes if n.op == 'Relu').op = 'Sigmoid'
    with pytest.raises(onnx_import.OnnxImportError, match='unsupported ONNX op Sigmoid'):
        onnx_import.import_graph(bad)
    # skeleton without payloads
    with pytest.raises(onnx_import.OnnxImportError, match='no payload'):
        onnx_import.import_graph(z['skeleton'].tobytes())
    # inconsistent shifts: no fraction-length assignment
    ig = onnx_import.import_graph(g)
    first = next(o for o in ig.ops if o.kind == 'conv' and o.shift is not None)
    first.shift = 40
    with pytest.raises(onnx_import.OnnxImportError, match='fraction-length'):
        ig.solve_fraclens()


@pytest.mark.parametrize('case', ['resnet50', 'mobilenet_v2'])
def test_imported_graph_plans_like_the_topology_table(case):
    """C ABI, no GPU: the free-form graph from the file plans into the same launches as the IntModel walk."""
    from f8net_amd.net import build_net
    z, spec, params, full, x, x_fl = load_case(case)
    ig = onnx_import.import_graph(full, input_signed=spec.normalize)
    a = ig.build_net(max_batch=8, hw=224, input_fraclen=x_fl)
    b = build_net(spec, params, max_batch=8, hw=224)
    assert a.num_launches == b.num_launches
    assert [a.launch_kernel(i) for i in range(a.num_launches)] == [b.launch_kernel(i) for i in range(b.num_launches)]


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES)
def test_gpu_runs_the_imported_file_bit_exact(case):
    torch = pytest.importorskip('torch')
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    z, spec, params, full, x, x_fl = load_case(case)
    ig = onnx_import.import_graph(full, input_signed=spec.normalize)
    net = ig.build_net(max_batch=2, input_fraclen=x_fl)
    got = net.run(torch.from_numpy(x).cuda()).cpu().numpy()
    np.testing.assert_array_equal(got, z['logits'])
    m = onnx_import.int_model_from_onnx(full, normalize=spec.normalize)
    xt = torch.from_numpy(x).cuda()
    setattr(xt, 'output_fraclen', int(m.head[0].input_fraclen.item()))
    np.testing.assert_array_equal(m(xt).cpu().numpy(), z['logits'])
