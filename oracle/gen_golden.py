#!/usr/bin/env python3
"""Generate golden fixtures by IMPORTING the reference (build container only).

    python oracle/gen_golden.py            # writes tests/golden/*.npz

The reference (/root/reference, read-only) is a Python package; it is imported here, driven
exactly as its own entry point drives it, and only DATA (inputs / expected outputs / checksums)
is written to tests/golden/.  No reference source travels.  /root/reference does not exist on the
GPU box; nothing under tests/ -m gpu, smoke() or bench.py runs this script.

What is restated from the reference's driver (it cannot be imported: it needs torchvision,
pytorchcv and CUDA at import / model build time, /root/reference/fix_train.py:22,38,269):
  * flag broadcast onto the QAT modules        fix_train.py:270-295
  * int_op_only conversion                      fix_train.py:930-934
  * input quantisation                          fix_train.py:683-692
One shim: torch>=2 refuses `param.data = int_tensor` on a grad-requiring Parameter
(fix_quant_ops.py:705-706 worked on the pinned torch 1.11), so new Conv2d/Linear parameters are
created with requires_grad=False while `int_model()` runs.  No reference file is modified.

One yml per process (the reference's FLAGS is an import-time singleton, myutils/config.py:152-178),
so this script re-executes itself per model with `--child`.
"""
import argparse
import importlib
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
GOLD = os.path.join(REPO, 'tests', 'golden')

YMLS = {
    'resnet18': 'apps/imagenet/resnet18/conventional/res18_fix_quant_test_int_op_only.yml',
    'resnet50': 'apps/imagenet/resnet50/tiny_finetuning/res50_fix_quant_nvidia_pretrained_test_int_op_only_on_cpu.yml',
    'mobilenet_v1': 'apps/imagenet/mobilenetv1/conventional/mbv1_fix_quant_test_int_op_only_on_cpu.yml',
    'mobilenet_v2': 'apps/imagenet/mobilenetv2/conventional/mbv2_fix_quant_test_int_op_only_on_cpu.yml',
}
# Depths the reference's `Model` builds (fix_resnet.py:418-451: block_type_dict / block_setting_dict) but ships no yml for: the yml of the
# same block type with FLAGS.depth overridden before the model is constructed (round 5: ResNet-34 / -101 / -152 — the 23- and 36-block stages
# force the chain cut at kChainMaxBlocks and the identity-first chain instances through whole networks)
DEPTH_OVERRIDE = {'resnet34': ('resnet18', 34), 'resnet101': ('resnet50', 101), 'resnet152': ('resnet50', 152)}
for _a, (_base, _d) in DEPTH_OVERRIDE.items():
    YMLS[_a] = YMLS[_base]
DEEP = list(DEPTH_OVERRIDE)


def checksum(a: np.ndarray):
    """(plain sum, position-weighted sum) in wrapping int64 — order- and value-sensitive."""
    v = np.ascontiguousarray(a).reshape(-1).astype(np.int64)
    with np.errstate(over='ignore'):
        wgt = (np.arange(v.size, dtype=np.int64) % 65521) + 1
        return np.array([v.sum(), (v * wgt).sum()], dtype=np.int64)


# --------------------------------------------------------------------------- child: one model

EXPORT_CASES = {
    # name: (arch, yml) — the two export regimes the shipped int_op_only ymls use
    'resnet18_metric': ('resnet18', 'apps/imagenet/resnet18/conventional/res18_fix_quant_test_int_op_only.yml'),
    'resnet18_gridsearch': ('resnet18', 'apps/imagenet/resnet18/tiny_finetuning/res18_fix_quant_ptcv_pretrained_test_int_op_only_on_cpu.yml'),
    'resnet50_gridsearch': ('resnet50', YMLS['resnet50']),
    'mobilenet_v1_metric': ('mobilenet_v1', YMLS['mobilenet_v1']),
    'mobilenet_v2_metric': ('mobilenet_v2', YMLS['mobilenet_v2']),
}


def build_reference_float_model(arch, yml=None, float_state=None):
    """The reference's FLOAT (fake-quant) model in eval mode with the flag broadcast of fix_train.py:270-295 — what `int_model()` converts, and what
    the `int_infer` evaluation mode (fix_quant_ops.py:418-431) runs on."""
    import torch
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.argv = ['gen_golden', f'app:{os.path.join(REF, yml or YMLS[arch])}', 'bs:1']
    from myutils.config import FLAGS
    if arch in DEPTH_OVERRIDE:
        FLAGS.depth = DEPTH_OVERRIDE[arch][1]
    model_lib = importlib.import_module(FLAGS.model)
    from models.fix_quant_ops import ReLUClipFXQConvBN, ReLUClipFXQLinear
    torch.manual_seed(0)
    model = model_lib.Model(FLAGS.num_classes)
    # -- fix_train.py:270-295
    for m in model.modules():
        if isinstance(m, (ReLUClipFXQConvBN, ReLUClipFXQLinear)):
            m.set_weight_format(FLAGS.weight_format)
            m.set_input_format(FLAGS.input_format)
            m.rescale_type = getattr(FLAGS, 'rescale_type', 'constant')
            m.set_alpha()
            m.floating = getattr(FLAGS, 'floating_model', False)
            m.floating_wo_clip = getattr(FLAGS, 'floating_wo_clip', False)
            m.format_type = getattr(FLAGS, 'format_type', None)
            m.format_from_metric = getattr(FLAGS, 'format_from_metric', False)
            m.metric = getattr(FLAGS, 'metric', None)
            m.format_grid_search = getattr(FLAGS, 'format_grid_search', False)
            m.set_metric_func()
            m.register_input_format(FLAGS.input_format,
                                    momentum=getattr(FLAGS, 'momentum_for_metric', 0.1))
            m.no_clipping = getattr(FLAGS, 'no_clipping', False)
            m.input_fraclen_sharing = getattr(FLAGS, 'input_fraclen_sharing', False)
            m.quant_bias = getattr(FLAGS, 'quant_bias', False)
            m.int_infer = getattr(FLAGS, 'int_infer', False)
        if isinstance(m, ReLUClipFXQConvBN):
            m.rescale_forward = getattr(FLAGS, 'rescale_forward_conv', False)
        if isinstance(m, ReLUClipFXQLinear):
            m.rescale_forward = getattr(FLAGS, 'rescale_forward', True)
    model.eval()
    if float_state is not None:
        # exporter parity: the float model's parameters / buffers are overwritten with our synthetic state
        sd = model.state_dict()
        want = {k for k in sd if not k.endswith('num_batches_tracked')}
        assert want == set(float_state), sorted(want ^ set(float_state))[:8]
        with torch.no_grad():
            for k, v in float_state.items():
                sd[k].copy_(torch.from_numpy(np.asarray(v)).reshape(sd[k].shape))
    return model, FLAGS


def build_reference_int_model(arch, yml=None, float_state=None, keep_float=False):
    import torch
    import torch.nn as nn
    model, FLAGS = build_reference_float_model(arch, yml, float_state)
    if keep_float:      # int_infer golden: the float model's own forward, taken BEFORE the conversion flags the modules int_op_only
        keep_float(model, FLAGS)
    # -- fix_train.py:930-934 with the requires_grad shim
    orig_conv, orig_lin = nn.Conv2d.__init__, nn.Linear.__init__

    def conv_init(self, *a, **k):
        orig_conv(self, *a, **k)
        for p in self.parameters():
            p.requires_grad_(False)

    def lin_init(self, *a, **k):
        orig_lin(self, *a, **k)
        for p in self.parameters():
            p.requires_grad_(False)

    model.apply(lambda m: setattr(m, 'int_op_only', True))
    nn.Conv2d.__init__, nn.Linear.__init__ = conv_init, lin_init
    try:
        with torch.no_grad():
            int_model = model.int_model().cpu()
    finally:
        nn.Conv2d.__init__, nn.Linear.__init__ = orig_conv, orig_lin
    int_model.apply(lambda m: setattr(m, 'int_op_only', True))
    return int_model, FLAGS


def child_model(arch):
    import torch
    import torch.nn as nn
    sys.path.insert(0, REPO)
    from f8net_amd import synth, topology
    int_model, FLAGS = build_reference_int_model(arch)
    normalize = bool(getattr(FLAGS, 'normalize', False))
    spec = topology.get(arch, normalize=normalize)

    # structural check: our topology table == the reference export
    sd = int_model.state_dict()
    ref_keys = sorted({k.rsplit('.', 1)[0] for k in sd})
    assert ref_keys == sorted(spec.layer_keys()), (ref_keys, spec.layer_keys())
    mods = dict(int_model.named_modules())
    for c in spec.convs():
        m = mods[c.key]
        assert isinstance(m, nn.Conv2d)
        assert (m.in_channels, m.out_channels, m.kernel_size[0], m.stride[0], m.padding[0], m.groups) == \
            (c.cin, c.cout, c.k, c.stride, c.pad, c.groups), c.key
        assert bool(m.input_symmetric) == c.signed_in, (c.key, m.input_symmetric)
    assert bool(mods[spec.fc_key].input_symmetric) == spec.fc_signed_in

    out = {}
    for seed in (1234,):
        # the reference's own learned fraclen tables where its logs hold one (ResNet-50 NVIDIA run, MobileNet-V2 log)
        params = synth.reference_params(spec, seed=seed)
        # overwrite the exported layers in place with our integers
        with torch.no_grad():
            for key in spec.layer_keys():
                m = mods[key]
                m.weight.data = torch.from_numpy(params[key + '.weight']).clone()
                m.bias.data = torch.from_numpy(params[key + '.bias']).clone()
                m.weight_fraclen.copy_(torch.from_numpy(params[key + '.weight_fraclen']))
                m.input_fraclen.copy_(torch.from_numpy(params[key + '.input_fraclen']))
        for hw, n in ((64, 2), (224, 1)):
            x_np, x_fl = synth.make_input(spec, params, n, hw, seed=7)
            caps = {}

            def mk(name):
                def hook(mod, inp, outp):
                    caps[name] = checksum(outp.detach().numpy())
                return hook
            handles = []
            for key in spec.layer_keys():
                handles.append(mods[key].register_forward_hook(mk(key)))
            for b in spec.blocks:
                handles.append(mods[b.name].register_forward_hook(mk(b.name)))
            x = torch.from_numpy(x_np)
            setattr(x, 'output_fraclen', x_fl)
            with torch.no_grad():
                logits = int_model(x)
            for h in handles:
                h.remove()
            tag = f's{seed}_hw{hw}_n{n}'
            out[f'{tag}/logits'] = logits.numpy().astype(np.float32)
            if hw == 64:
                # the reference's float-carried "integize" evaluation of the same IntModel (the non-int_op_only
                # branches: fix_resnet.py:78-118,384-409, fix_mobilenet_v2.py / _v1.py likewise): real-valued float tensors,
                # requantised with fix_quant, convs in float32.  Fed the real value of the same integers.
                # (the integize export keeps the integer weights in float tensors, fix_quant_ops.py:700-707: same values)
                int_model.apply(lambda m: setattr(m, 'int_op_only', False))
                layers = [mods[key] for key in spec.layer_keys()]
                for m in layers:
                    m.weight.data = m.weight.data.float()
                    m.bias.data = m.bias.data.float()
                xr = torch.from_numpy(x_np).float() / float(2 ** x_fl)
                with torch.no_grad():
                    lg = int_model(xr)
                for m in layers:
                    m.weight.data = m.weight.data.int()
                    m.bias.data = m.bias.data.int()
                int_model.apply(lambda m: setattr(m, 'int_op_only', True))
                out[f'{tag}/integize_logits'] = lg.numpy().astype(np.float32)
                out[f'{tag}/integize_equal'] = np.array(bool(np.array_equal(lg.numpy(), logits.numpy())))
            assert np.count_nonzero(out[f'{tag}/logits']) > 0.9 * logits.numel(), 'degenerate logits'
            names = sorted(caps)
            out[f'{tag}/cap_names'] = np.array(names)
            out[f'{tag}/cap_sums'] = np.stack([caps[k] for k in names])
    out['normalize'] = np.array(normalize)
    np.savez_compressed(os.path.join(GOLD, f'net_{arch}.npz'), **out)
    print(f'[gen_golden] {arch}: wrote net_{arch}.npz ({len(out)} arrays)')


def child_export(case):
    """Exporter parity: the reference's own int_model() on a float model holding our synthetic state;
    the fixture keeps checksums of the int32 weights, the biases and fraclens in full."""
    import torch
    sys.path.insert(0, REPO)
    from f8net_amd import synth, topology
    arch, yml = EXPORT_CASES[case]
    spec0 = topology.get(arch)
    fstate = synth.make_float_state(spec0, seed=77)
    int_model, FLAGS = build_reference_int_model(arch, yml=yml, float_state=fstate)
    sd = int_model.state_dict()
    out = {'flags': np.array([int(bool(getattr(FLAGS, k, False))) for k in
                              ('normalize', 'format_from_metric', 'format_grid_search', 'no_clipping', 'input_fraclen_sharing',
                               'quant_avgpool', 'pool_fusing', 'rescale_forward', 'rescale_forward_conv')], dtype=np.int32)}
    names, sums = [], []
    for k in sorted(sd):
        v = sd[k].numpy()
        assert v.dtype == np.int32, (k, v.dtype)
        if k.endswith('.weight'):
            names.append(k)
            sums.append(checksum(v))
            out[f'head/{k}'] = v.reshape(-1)[:64].copy()
        else:
            out[f'full/{k}'] = v.copy()
    out['weight_names'] = np.array(names)
    out['weight_sums'] = np.stack(sums)
    np.savez_compressed(os.path.join(GOLD, f'export_{case}.npz'), **out)
    wfl = [int(sd[k]) for k in sorted(sd) if k.endswith('weight_fraclen')]
    print(f'[gen_golden] export {case}: {len(names)} layers, weight fraclens {sorted(set(wfl))}, wrote export_{case}.npz')


def child_intinfer(case):
    """`int_infer` evaluation mode (fix_quant_ops.py:418-431): the reference's FLOAT model — eval mode, `int_infer: True` as every
    shipped test yml sets it — on a real-valued float batch, BEFORE `int_model()` touches it.  The fixture keeps its logits; the batch is
    `synth.rand_uniform_int(9, 'intinfer', ...) / 2^(head input fraclen)` (regenerated by the tests)."""
    import torch
    sys.path.insert(0, REPO)
    from f8net_amd import synth, topology
    arch, yml = EXPORT_CASES[case]
    fstate = synth.make_float_state(topology.get(arch), seed=77)
    got = {}

    def run_float(model, FLAGS):
        assert getattr(FLAGS, 'int_infer', False), 'the yml must evaluate in int_infer mode'
        norm = bool(getattr(FLAGS, 'normalize', False))
        hfl = int(torch.round(model.head[0].get_input_fraclen()).item())
        xi = synth.rand_uniform_int(9, 'intinfer', (2, 3, 64, 64), -127 if norm else 0, 127 if norm else 255).astype(np.float32)
        with torch.no_grad():
            got['logits'] = model(torch.from_numpy(xi / float(2 ** hfl))).numpy().copy()
        got['meta'] = np.array([hfl, int(norm)], dtype=np.int32)

    _, FLAGS = build_reference_int_model(arch, yml=yml, float_state=fstate, keep_float=run_float)
    out = {'logits': got['logits'], 'meta': got['meta'],
           'flags': np.array([int(bool(getattr(FLAGS, k, False))) for k in
                              ('normalize', 'format_from_metric', 'format_grid_search', 'no_clipping', 'input_fraclen_sharing',
                               'quant_avgpool', 'pool_fusing', 'rescale_forward', 'rescale_forward_conv')], dtype=np.int32)}
    np.savez_compressed(os.path.join(GOLD, f'intinfer_{case}.npz'), **out)
    print(f'[gen_golden] int_infer {case}: head fraclen {got["meta"][0]}, logits {got["logits"].shape} max |.| {np.abs(got["logits"]).max():.4f}, wrote intinfer_{case}.npz')


ONNX_LSHIFT = {'stage_0_layer_0.body.0': (6, 0), 'stage_0_layer_0.body.2': (8, 5),      # n = 6 - 8 = -2
               'stage_1_layer_0.body.0': (5, 1), 'stage_1_layer_0.body.2': (7, 6)}      # n = 6 - 7 = -1


def child_onnx(arch):
    """ONNX importer fixtures: the reference IntModel holding our synthetic integers is exported exactly
    as the reference does it (myutils/export.py:4-31: torch.onnx.export, opset 11, batch 1, dynamic batch axis); the
    fixture keeps the file with the payload of its large initializers dropped (the test regenerates them from the same
    seed) and the logits the PyTorch IntModel returns for a 2-image batch."""
    import io
    import torch
    sys.path.insert(0, REPO)
    from f8net_amd import onnx_io, synth, topology
    tag, arch = arch, arch.split('+')[0]                   # 'resnet18+lshift': fraclens that force left-shift requants
    int_model, FLAGS = build_reference_int_model(arch)
    normalize = bool(getattr(FLAGS, 'normalize', False))
    spec = topology.get(arch, normalize=normalize)
    mods = dict(int_model.named_modules())
    fr = topology.R50_NVIDIA_FRACLENS if arch == 'resnet50' else (ONNX_LSHIFT if tag.endswith('+lshift') else None)
    params = synth.make_params(spec, seed=4321, fraclens=fr)
    with torch.no_grad():
        for key in spec.layer_keys():
            m = mods[key]
            m.weight.data = torch.from_numpy(params[key + '.weight']).clone()
            m.bias.data = torch.from_numpy(params[key + '.bias']).clone()
            m.weight_fraclen.copy_(torch.from_numpy(params[key + '.weight_fraclen']))
            m.input_fraclen.copy_(torch.from_numpy(params[key + '.input_fraclen']))
    hw = 64
    x_np, x_fl = synth.make_input(spec, params, 2, hw, seed=11)
    x = torch.from_numpy(x_np)
    setattr(x, 'output_fraclen', x_fl)
    with torch.no_grad():
        logits = int_model(x).numpy().astype(np.float32)
    assert np.count_nonzero(logits) > 0.9 * logits.size, 'degenerate logits'
    # torch's exporter only needs the `onnx` package (absent here) to splice onnxscript functions in; there are none
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda proto, custom_opsets: proto
    # myutils/export.py:7-13 draws randn().to(int32) — its values are irrelevant to the trace; the attribute the
    # int forward asserts on (fix_resnet.py:353) has to be there, as it is when fix_train.py:683-692 makes the input
    dummy = torch.from_numpy(x_np[:1].copy())
    setattr(dummy, 'output_fraclen', x_fl)
    buf = io.BytesIO()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        torch.onnx.export(int_model, dummy, buf, export_params=True, opset_version=11, do_constant_folding=True,
                          input_names=['input'], output_names=['output'],
                          dynamic_axes={'input': {0: 'batch_size'}, 'output': {0: 'batch_size'}}, dynamo=False)
    full = buf.getvalue()
    skeleton, stripped = onnx_io.strip_initializers(full, 4096)
    g = onnx_io.load_graph(full)
    for name in stripped:                                   # the test refills these from synth.make_params
        assert name in params and np.array_equal(g.initializers[name], params[name]), name
    out = {'skeleton': np.frombuffer(skeleton, np.uint8), 'stripped': np.array(stripped), 'seed': np.array(4321),
           'input_seed': np.array(11), 'hw': np.array(hw), 'normalize': np.array(normalize), 'logits': logits,
           'full_bytes': np.array(len(full))}
    for key in spec.layer_keys():
        out[f'fl/{key}'] = np.array([int(params[key + '.input_fraclen'].reshape(-1)[0]),
                                     int(params[key + '.weight_fraclen'].reshape(-1)[0])])
    np.savez_compressed(os.path.join(GOLD, f'onnx_{tag.replace("+", "_")}.npz'), **out)
    print(f'[gen_golden] onnx {tag}: {len(full)} B file -> {len(skeleton)} B skeleton, {len(stripped)} stripped')


def child_ops():
    """Op-level known answers from the reference's own functions / the torch ops it calls."""
    import torch
    import torch.nn as nn
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    sys.argv = ['gen_golden', f'app:{os.path.join(REF, YMLS["resnet18"])}', 'bs:1']
    from models.fix_quant_ops import int_op_only_fix_quant, FXQAvgPool2d, FXQMaxPool2d
    from f8net_amd import synth
    out = {}
    # --- requant KATs (SURVEY.md App. B input vector + edge values + random)
    kat_in = np.array([-13, -12, -11, -10, -9, -8, -7, -6, -5, -4, -3, -2, -1, 0, 1, 2, 3, 4, 5, 6, 7, 8,
                       9, 10, 11, 12, 13, 2000, -2000, 1020, 1022, 1018], dtype=np.int32)
    edge = np.array([2**31 - 1, -2**31, -2**31 + 1, 2**31 - 2, 2**30, -2**30, 2**30 - 1, 255, 256, 254,
                     127, 128, -127, -128, 65535, -65536, 1 << 23, -(1 << 23)], dtype=np.int32)
    rnd = synth.rand_normal_int(3, 'requant', (4096,), 30000.0).astype(np.int32)
    rnd2 = synth.rand_uniform_int(4, 'requant2', (2048,), -2**31, 2**31 - 1).astype(np.int32)
    vec = np.concatenate([kat_in, edge, rnd, rnd2])
    out['requant/in'] = vec
    cases = []
    for signed in (True, False):
        for dst_fl in range(0, 8 if signed else 9):
            for src_fl in (0, 3, 5, 6, 8, 9, 11, 14, 19, 22):
                if src_fl - dst_fl > 20 or src_fl - dst_fl < -6:
                    continue
                r = int_op_only_fix_quant(torch.from_numpy(vec.copy()), 8, dst_fl, src_fl, signed)
                assert r.dtype == torch.int32 and r.output_fraclen == dst_fl
                cases.append((dst_fl, src_fl, int(signed)))
                out[f'requant/out_{dst_fl}_{src_fl}_{int(signed)}'] = r.numpy()
    out['requant/cases'] = np.array(cases, dtype=np.int32)
    # --- integer conv / linear exactly as the reference executes them: nn.Conv2d with int32 params
    geoms = [  # (N, C, H, W, K, k, stride, pad, groups)
        (2, 3, 17, 19, 8, 7, 2, 3, 1), (2, 16, 9, 9, 24, 3, 1, 1, 1), (1, 16, 10, 10, 8, 3, 2, 1, 1),
        (2, 32, 7, 7, 16, 1, 1, 0, 1), (2, 32, 8, 8, 16, 1, 2, 0, 1), (2, 24, 9, 11, 24, 3, 1, 1, 24),
        (1, 24, 10, 8, 24, 3, 2, 1, 24), (1, 3, 12, 12, 8, 3, 2, 1, 1),
    ]
    for gi, (N, C, H, W, K, k, s, p, g) in enumerate(geoms):
        x = synth.rand_uniform_int(11, f'cx{gi}', (N, C, H, W), -127, 255).astype(np.int32)
        w = synth.rand_uniform_int(12, f'cw{gi}', (K, C // g, k, k), -127, 127).astype(np.int32)
        b = synth.rand_normal_int(13, f'cb{gi}', (K,), 5e4).astype(np.int32)
        conv = nn.Conv2d(C, K, k, stride=s, padding=p, groups=g, bias=True)
        conv.weight.requires_grad_(False)
        conv.bias.requires_grad_(False)
        conv.weight.data = torch.from_numpy(w)
        conv.bias.data = torch.from_numpy(b)
        with torch.no_grad():
            y = conv(torch.from_numpy(x))
        assert y.dtype == torch.int32
        out[f'conv/{gi}/geom'] = np.array([N, C, H, W, K, k, s, p, g], dtype=np.int32)
        out[f'conv/{gi}/x'], out[f'conv/{gi}/w'], out[f'conv/{gi}/b'] = x, w, b
        out[f'conv/{gi}/y'] = y.numpy()
    # wrap-around (conv(65536*65536) == 0)
    conv = nn.Conv2d(1, 1, 1, bias=True)
    for prm in conv.parameters():
        prm.requires_grad_(False)
    conv.weight.data = torch.full((1, 1, 1, 1), 65536, dtype=torch.int32)
    conv.bias.data = torch.tensor([7], dtype=torch.int32)
    with torch.no_grad():
        out['conv/wrap_y'] = conv(torch.full((1, 1, 2, 2), 65537, dtype=torch.int32)).numpy()
    x = synth.rand_uniform_int(21, 'lx', (3, 40), -127, 255).astype(np.int32)
    w = synth.rand_uniform_int(22, 'lw', (10, 40), -127, 127).astype(np.int32)
    b = synth.rand_normal_int(23, 'lb', (10,), 1e5).astype(np.int32)
    fc = nn.Linear(40, 10)
    for prm in fc.parameters():
        prm.requires_grad_(False)
    fc.weight.data, fc.bias.data = torch.from_numpy(w), torch.from_numpy(b)
    with torch.no_grad():
        y = fc(torch.from_numpy(x))
    out['linear/x'], out['linear/w'], out['linear/b'], out['linear/y'] = x, w, b, y.numpy()
    out['linear/y_float'] = y.float().numpy()
    # --- pools
    ap = FXQAvgPool2d(7)
    ap.int_op_only = True
    x = synth.rand_normal_int(31, 'ap', (2, 5, 7, 7), 3e5).astype(np.int32)
    t = torch.from_numpy(x)
    setattr(t, 'output_fraclen', 13)
    r = ap(t)
    out['avgpool/x'], out['avgpool/y'] = x, r.numpy()
    out['avgpool/fl'] = np.array([13, r.output_fraclen], dtype=np.int32)
    x = np.maximum(synth.rand_normal_int(32, 'mp', (2, 4, 11, 12), 2e5), 0).astype(np.int32)
    mp = nn.MaxPool2d(3, 2, 1)
    out['maxpool/x'] = x
    out['maxpool/y'] = mp(torch.from_numpy(x).float()).int().numpy()          # fix_resnet.py:359
    out['maxpool/y_fxq'] = FXQMaxPool2d(3, 2, 1)(torch.from_numpy(x)).numpy()  # quant_maxpool: True
    # --- residual align-add (fix_resnet.py:40-54), run through torch ops as the reference does
    a = synth.rand_normal_int(41, 'ra', (2, 6, 5, 5), 4e5).astype(np.int32)
    bb = synth.rand_normal_int(42, 'rb', (2, 6, 5, 5), 4e5).astype(np.int32)
    a.reshape(-1)[:4] = [2**31 - 1, -2**31, 2**30, -2**30]
    bb.reshape(-1)[:4] = [1, -1, 2**30, -2**30]
    out['add/res'], out['add/x'] = a, bb
    for res_fl, x_fl in ((11, 9), (9, 12), (10, 10)):
        res, x = torch.from_numpy(a.copy()), torch.from_numpy(bb.copy())
        if res_fl > x_fl:
            x = x << (res_fl - x_fl)
            res += x
        else:
            res = res << (x_fl - res_fl)
            res += x
        res.clamp_(max=(1 << 31) - 1, min=-(1 << 31) + 1)
        out[f'add/out_{res_fl}_{x_fl}'] = res.numpy()
    # --- input quantisation (fix_train.py:683-692)
    from models.fix_quant_ops import fix_quant
    img = (synth.rand_uniform_int(51, 'img', (1, 3, 8, 8), 0, 1000).astype(np.float32) / 1000.0)
    out['inq/img'] = img
    out['inq/u8'] = (255 * torch.from_numpy(img.copy())).round_().int().numpy()
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(1, 3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(1, 3, 1, 1)
    xn = ((img - mean) / std).astype(np.float32)
    fl = torch.tensor([5], dtype=torch.int32)
    q = (fix_quant(torch.from_numpy(xn.copy()), 8, fl * 1.0, 1, True)[0] * (2 ** fl)).int()
    out['inq/xn'], out['inq/s8_fl5'] = xn, q.numpy()
    # --- the same from the decoder's uint8 pixels: torchvision is absent, so the tensor statements of
    #     transforms.ToTensor (`img.to(float32).div(255)`) and transforms.Normalize (`tensor.sub_(mean).div_(std)`),
    #     fix_train.py:299-329, are executed as written on torch; every pixel value occurs in every channel
    u8 = np.concatenate([np.tile(np.arange(256, dtype=np.uint8).reshape(1, 1, 16, 16), (1, 3, 1, 1)),
                         synth.rand_uniform_int(52, 'u8img', (1, 3, 16, 16), 0, 255).astype(np.uint8)])
    t = torch.from_numpy(u8.copy()).to(dtype=torch.float32).div(255)
    out['inq8/u8'] = u8
    out['inq8/plain'] = (255 * t.clone()).round_().int().numpy()                   # fix_train.py:689-692
    tn = t.clone()
    tn.sub_(torch.from_numpy(mean)).div_(torch.from_numpy(std))
    out['inq8/mean'], out['inq8/std'] = mean.reshape(3), std.reshape(3)
    for fl8 in (4, 5, 6):
        flt = torch.tensor([fl8], dtype=torch.int32)
        out[f'inq8/s8_fl{fl8}'] = (fix_quant(tn.clone(), 8, flt * 1.0, 1, True)[0] * (2 ** flt)).int().numpy()   # fix_train.py:683-687
    # --- scoring (fix_train.py:697-704; fix_train.py itself cannot be imported here — torchvision / pytorchcv are absent —
    #     so its five tensor statements are executed as written, on torch, with FLAGS.topk = [1, 5])
    logits = synth.rand_normal_int(61, 'logits', (12, 40), 3.0e4).astype(np.float32)      # integer-valued, like int_op_only logits
    target = synth.rand_uniform_int(62, 'target', (12,), 0, 39).astype(np.int64)
    am = np.argsort(-logits, axis=1, kind='stable')
    target[0], target[1], target[2] = am[0, 0], am[1, 4], am[2, 5]              # top-1 hit, top-5 edge hit, just outside top-5
    output, tgt = torch.from_numpy(logits), torch.from_numpy(target)
    topk = [1, 5]
    _, pred = output.topk(max(topk))
    pred = pred.t()
    correct = pred.eq(tgt.view(1, -1).expand_as(pred))
    correct_k = []
    for k in topk:
        correct_k.append(correct[:k].float().sum(0))
    out['topk/logits'], out['topk/target'] = logits, target
    out['topk/correct'] = torch.stack(correct_k, 0).numpy()
    np.savez_compressed(os.path.join(GOLD, 'ops.npz'), **out)
    print(f'[gen_golden] ops: wrote ops.npz ({len(out)} arrays)')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--child', default=None)
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    if args.child == 'ops':
        child_ops()
    elif args.child and args.child.startswith('onnx:'):
        child_onnx(args.child.split(':', 1)[1])
    elif args.child and args.child.startswith('export:'):
        child_export(args.child.split(':', 1)[1])
    elif args.child and args.child.startswith('intinfer:'):
        child_intinfer(args.child.split(':', 1)[1])
    elif args.child:
        child_model(args.child)
    else:
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
        base = [a for a in YMLS if a not in DEEP]
        for c in ['ops'] + list(YMLS) + [f'export:{c}' for c in EXPORT_CASES] + [f'intinfer:{c}' for c in EXPORT_CASES] + [f'onnx:{a}' for a in base + ['resnet18+lshift']]:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), '--child', c], env=env)


if __name__ == '__main__':
    main()
