"""Randomised topologies through the planner (GPU) vs the CPU oracle: mixes of bottleneck / basic / inverted-residual /
depthwise-separable blocks with random widths, strides, signed inputs and fraction lengths, at the spatial sizes where the
special kernels kick in (56 / 28 / 14 / 7 wide: fused blocks, dual-GEMM joins, LDS-patch 3x3, fused head).  Bit-exact."""
import os

import numpy as np
import pytest

from f8net_amd import synth, topology
from f8net_amd.topology import BlockSpec, ConvSpec, NetSpec
from oracle import oracle

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


def _rnd(seed, key, lo, hi):
    return int(synth.rand_uniform_int(seed, key, (), lo, hi))


def random_spec(seed):
    """A small ResNet-ish / MobileNet-ish net: head (7x7/2 + pool, or 3x3/1), 3-4 random blocks, avg-pool, FC."""
    r = lambda key, lo, hi: _rnd(seed, key, lo, hi)
    resnet_head = r('head', 0, 1) == 1
    normalize = r('norm', 0, 1) == 1
    if resnet_head:
        head = ConvSpec('head.0', 3, 64, 7, 2, 3, signed_in=normalize, relu=True)
        ch = 64
    else:
        ch = [32, 64][r('hc', 0, 1)]
        head = ConvSpec('head.0', 3, ch, 3, 1, 1, signed_in=normalize, relu=True)
    blocks = []
    for bi in range(r('nblk', 3, 4)):
        kind = r(f'k{bi}', 0, 3)
        name = f'stage_{bi}_layer_0'
        stride = 2 if r(f's{bi}', 0, 3) == 0 else 1
        if kind == 0:       # bottleneck
            outp = [64, 128, 256, 512][r(f'o{bi}', 0, 3)]
            mid = outp // 4 if outp >= 128 else 32
            body = [ConvSpec(f'{name}.body.0', ch, mid, 1, 1, 0, relu=True),
                    ConvSpec(f'{name}.body.2', mid, mid, 3, stride, 1, relu=True),
                    ConvSpec(f'{name}.body.4', mid, outp, 1, 1, 0)]
            sc = None if (stride == 1 and ch == outp) else ConvSpec(f'{name}.shortcut.0', ch, outp, 1, stride, 0)
            blocks.append(BlockSpec(name, body, sc, residual=True, post_relu=True))
        elif kind == 1:     # basic
            outp = [32, 64, 128, 256][r(f'o{bi}', 0, 3)]
            body = [ConvSpec(f'{name}.body.0', ch, outp, 3, stride, 1, relu=True),
                    ConvSpec(f'{name}.body.2', outp, outp, 3, 1, 1)]
            sc = None if (stride == 1 and ch == outp) else ConvSpec(f'{name}.shortcut.0', ch, outp, 1, stride, 0)
            blocks.append(BlockSpec(name, body, sc, residual=True, post_relu=True))
        elif kind == 2:     # inverted residual (signed input on the first conv)
            outp = [24, 32, 64, 96][r(f'o{bi}', 0, 3)]
            e = ch * [1, 2][r(f't{bi}', 0, 1)] if ch <= 128 else ch
            body = [ConvSpec(f'{name}.body.0', ch, e, 1, 1, 0, signed_in=True, relu=True),
                    ConvSpec(f'{name}.body.2', e, e, 3, stride, 1, groups=e, relu=True),
                    ConvSpec(f'{name}.body.4', e, outp, 1, 1, 0)]
            blocks.append(BlockSpec(name, body, None, residual=(stride == 1 and ch == outp), post_relu=False))
        else:               # depthwise separable
            outp = [32, 64, 128, 160][r(f'o{bi}', 0, 3)]
            body = [ConvSpec(f'{name}.body.0', ch, ch, 3, stride, 1, groups=ch, relu=True),
                    ConvSpec(f'{name}.body.2', ch, outp, 1, 1, 0, relu=True)]
            blocks.append(BlockSpec(name, body, None, residual=False, post_relu=False))
        ch = outp
    return NetSpec(f'fuzz{seed}', head, resnet_head, blocks, None, 'classifier.0', ch, 20, normalize=normalize), resnet_head


@pytest.mark.parametrize('seed', list(range(24)))
def test_random_topology_matches_oracle(seed):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from f8net_amd.net import build_net
    oracle.build()
    dev = torch.device('cuda', 0)
    spec, resnet_head = random_spec(1000 + seed)
    # spatial size after the head: 56 / 28 / 14 / 7 (special kernels) or something ragged
    target = [56, 28, 14, 7, 20, 9][_rnd(seed, 'hw', 0, 5)]
    hw = target * 4 if resnet_head else target
    n = [1, 2, 3, 5][_rnd(seed, 'n', 0, 3)]
    params = synth.make_params(spec, seed=seed)
    x, x_fl = synth.make_input(spec, params, n, hw, seed=seed + 1)
    net = build_net(spec, params, max_batch=n, hw=hw)
    got = net.run(torch.from_numpy(x).to(dev)).cpu().numpy()
    want = oracle.net_forward(spec, params, x, x_fl)
    np.testing.assert_array_equal(got, want, err_msg=net.describe())


@pytest.mark.parametrize('seed', list(range(8)))
def test_resnet_like_random_fraclens(seed):
    """Truncated ResNet-50s (real channel counts at 56 / 28 / 14 wide, so the fused bottleneck, stage-opening, dual-GEMM,
    patch and fused-head kernels are all planned) with random fraction lengths, batch sizes and cut points."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from f8net_amd.net import build_net
    oracle.build()
    dev = torch.device('cuda', 0)
    full = topology.get('resnet50', num_classes=24, normalize=(seed % 2 == 0))
    nblk = [3, 4, 7, 8, 10][_rnd(seed, 'cut', 0, 4)]
    blocks = full.blocks[:nblk]
    spec = NetSpec(f'r50cut{seed}', full.head, True, blocks, None, 'classifier.0', blocks[-1].body[-1].cout, 24, normalize=full.normalize)
    fr = {}
    if seed % 3 != 2:       # shortcut and body.0 of a downsample block share one input format (as in the real fraclen tables)
        base = synth.make_params(spec, seed=seed)
        for b in blocks:
            if b.shortcut is not None:
                in_fl = int(base[b.body[0].key + '.input_fraclen'][0])
                fr[b.shortcut.key] = (in_fl, int(base[b.shortcut.key + '.weight_fraclen']))
    n = [1, 2, 3][_rnd(seed, 'n', 0, 2)]
    # A net cut after an early stage pools a 56x56 / 28x28 map: with full-range weights its sum can exceed 2^32 - 1, where the
    # reference asserts (fix_quant_ops.py:132).  Every case must RUN, so the weight spread shrinks until the reference accepts.
    want = None
    for w_sigma in (24.0, 8.0, 2.5, 1.0):
        params = synth.make_params(spec, seed=seed, fraclens=fr, w_sigma=w_sigma)
        x, x_fl = synth.make_input(spec, params, n, 224, seed=seed + 7)
        try:
            want = oracle.net_forward(spec, params, x, x_fl)
            break
        except AssertionError:
            continue
    assert want is not None, 'no weight spread keeps the avg-pool sum inside the range the reference accepts'
    net = build_net(spec, params, max_batch=n, hw=224)
    plan = net.describe()
    if not any(k.startswith('F8_') for k in os.environ):     # default planner (tuning switches change the plan, not the results)
        assert ('fused_bottleneck_R' in plan or 'stage_chain_x' in plan) and 'stem7x7s2+maxpool3x3s2' in plan
    got = net.run(torch.from_numpy(x).to(dev)).cpu().numpy()
    np.testing.assert_array_equal(got, want, err_msg=plan)
