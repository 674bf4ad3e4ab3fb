"""`int_infer` evaluation mode (reference: models/fix_quant_ops.py:418-431, 916-929, fix_resnet.py:158-187, 489-505).

`oracle/gen_golden.py --child intinfer:<case>` ran the reference's FLOAT model (eval, `int_infer: True` as the shipped test ymls set it) holding
`synth.make_float_state(seed=77)` on a real-valued batch and stored its logits.  Here the same float state goes through
`export.export_int_state(..., int_infer_eval=True)` — the integers that mode computes with — and the integer network runs on the batch quantised as
the float head quantises it: on the CPU oracle (not gpu) and on the GPU through `IntModel.forward_int_infer` (gpu).  The mode is float-carried in
the reference (`not bit-exact by construction`, README.md:76); wherever float32 holds its accumulators exactly the two agree to the last bit, which
the synthetic nets do — the tolerance below only allows for the reference's float sums."""
import dataclasses
import os

import numpy as np
import pytest

from f8net_amd import synth, topology

torch = pytest.importorskip('torch')
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
FLAG_NAMES = ('normalize', 'format_from_metric', 'format_grid_search', 'no_clipping', 'input_fraclen_sharing',
              'quant_avgpool', 'pool_fusing', 'rescale_forward', 'rescale_forward_conv')
CASES = [('resnet18_metric', 'resnet18'), ('resnet18_gridsearch', 'resnet18'), ('resnet50_gridsearch', 'resnet50'),
         ('mobilenet_v1_metric', 'mobilenet_v1'), ('mobilenet_v2_metric', 'mobilenet_v2')]
RTOL, ATOL = 0.0, 2e-6          # logits are k / 2^11 .. k / 2^13: one float32 ulp of slack for the reference's float pool / residual sums


def _setup(case, arch):
    from f8net_amd import export
    g = np.load(os.path.join(GOLD, f'intinfer_{case}.npz'))
    cfg = export.ExportConfig(**dict(zip(FLAG_NAMES, (bool(v) for v in g['flags']))))
    spec = topology.get(arch, normalize=cfg.normalize)
    fstate = synth.make_float_state(topology.get(arch), seed=77)
    hfl, norm = (int(v) for v in g['meta'])
    assert bool(norm) == cfg.normalize
    xi = synth.rand_uniform_int(9, 'intinfer', (2, 3, 64, 64), -127 if norm else 0, 127 if norm else 255)
    return g, cfg, spec, fstate, hfl, xi


@pytest.mark.parametrize('case,arch', CASES)
def test_int_infer_export_on_the_oracle_matches_the_reference_float_model(case, arch):
    from f8net_amd import export
    from oracle import oracle
    oracle.build()
    g, cfg, spec, fstate, hfl, xi = _setup(case, arch)
    sd = export.export_int_state(spec, fstate, dataclasses.replace(cfg, int_infer_eval=True))
    params = {k: v.numpy() for k, v in sd.items()}
    assert int(params['head.0.input_fraclen'].reshape(-1)[0]) == hfl
    y = oracle.net_forward(spec, params, xi.astype(np.int32), hfl)
    fc = spec.fc_key
    scale = float(2 ** (int(params[fc + '.weight_fraclen'].reshape(-1)[0]) + int(params[fc + '.input_fraclen'].reshape(-1)[0])))
    np.testing.assert_allclose(y / scale, g['logits'], rtol=RTOL, atol=ATOL)
    # the mode is NOT the exported IntModel on ResNets / MobileNet-V1 (the pool's 64 / 49 never reaches a conv there: ExportConfig.int_infer_eval)
    sd0 = export.export_int_state(spec, fstate, cfg)
    same = all(torch.equal(sd0[k], sd[k]) for k in sd)
    assert same == (arch == 'mobilenet_v2')


@pytest.mark.gpu
@pytest.mark.parametrize('case,arch', CASES)
def test_forward_int_infer_on_the_gpu_matches_the_reference_float_model(case, arch):
    from f8net_amd import export
    g, cfg, spec, fstate, hfl, xi = _setup(case, arch)
    m = export.int_infer_model_from_float(arch, fstate, cfg).cuda()
    x = torch.from_numpy(xi.astype(np.float32) / float(2 ** hfl)).cuda()
    got = m.forward_int_infer(x).cpu().numpy()
    np.testing.assert_allclose(got, g['logits'], rtol=RTOL, atol=ATOL)
