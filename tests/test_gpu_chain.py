"""GPU parity of the stage-chain launch (f8_chain.hip): ALL consecutive bottleneck blocks of a ResNet stage in one launch, the
int32 residual stream in registers, halo rows exchanged between workgroups.  Every case is a net of its own through the C ABI,
compared bit for bit with the oracle's IntBlock.forward (fix_resnet.py:26-77) applied block after block."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from f8net_amd import synth, topology
from f8net_amd.net import F8Net
from oracle import oracle


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return torch.device('cuda:0')


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _params(convs, fls, seed, tag):
    p = {}
    for c in convs:
        in_fl, w_fl = fls[c.key]
        p[c.key + '.weight'] = np.clip(synth.rand_normal_int(seed, c.key + 'w' + tag, (c.cout, c.cin, c.k, c.k), 30.0), -127, 127).astype(np.int32)
        p[c.key + '.bias'] = synth.rand_normal_int(seed + 1, c.key + 'b' + tag, (c.cout,), 2.0 ** (in_fl + w_fl)).astype(np.int32)
        p[c.key + '.weight_fraclen'] = np.array(w_fl, np.int32)
        p[c.key + '.input_fraclen'] = np.array([in_fl], np.int32)
    return p


def _stage(C, MID, nblk, cin0, variant):
    """Block specs + fraclens of `nblk` bottleneck blocks; cin0 != C: the first one is the stage-opening block (1x1 shortcut)."""
    blocks, fls = [], {}
    for k in range(nblk):
        cin = cin0 if (k == 0 and cin0 != C) else C
        name = f's.{k}'
        body = [topology.ConvSpec(name + '.body.0', cin, MID, 1, 1, 0, relu=True),
                topology.ConvSpec(name + '.body.2', MID, MID, 3, 1, 1, relu=True),
                topology.ConvSpec(name + '.body.4', MID, C, 1, 1, 0)]
        sc = topology.ConvSpec(name + '.shortcut.0', cin, C, 1, 1, 0) if cin != C else None
        blocks.append(topology.BlockSpec(name, body, sc, residual=True, post_relu=True))
        if variant == 'acc_shifts_left':          # the stream's fraclen stays at its maximum: body.4 results shift left (the usual case)
            f0, f2, f4 = [(4, 7), (3, 6), (3, 5 + (k % 2))][0], (3, 6), (3, 5 + (k % 2))
            fsc = (4, 7)
        else:                                      # every block raises the stream's fraclen: the residual shifts left
            f0, f2, f4 = (4, 7), (4, 7), (min(5 + k, 7), 7)
            fsc = (4, 6)
        fls[name + '.body.0'], fls[name + '.body.2'], fls[name + '.body.4'] = f0, f2, f4
        if sc is not None:
            fls[name + '.shortcut.0'] = fsc
    return blocks, fls


# C, MID, H(=W), blocks, cin0, N   — N beyond 256 / tiles-per-image makes a workgroup walk several images
CHAINS = [
    (1024, 256, 14, 3, 1024, 5),
    (1024, 256, 14, 5, 1024, 70),
    (512, 128, 28, 3, 512, 3),
    (512, 128, 28, 2, 512, 41),
    (256, 64, 56, 2, 256, 2),
    (256, 64, 56, 3, 64, 2),
    (256, 64, 56, 3, 64, 21),
    # 7x7 (ResNet-50 stage 3): f8_cchain.hip — clusters of eight workgroups over four images; N = 5: one whole group + one image, 130: 33 groups on 32 clusters
    (2048, 512, 7, 2, 2048, 5),
    (2048, 512, 7, 3, 2048, 130),
]


@pytest.mark.parametrize('cfg', CHAINS, ids=lambda g: 'x'.join(map(str, g)))
@pytest.mark.parametrize('variant', ['acc_shifts_left', 'res_shifts_left'])
@pytest.mark.parametrize('tail', ['int32_out', 'int8_out'])
def test_stage_chain_matches_oracle(dev, cfg, variant, tail):
    if tail == 'int8_out' and cfg[5] > 8:
        pytest.skip('the int8 tail is covered at the small batch')
    _run_stage_chain(dev, cfg, variant, tail)


@pytest.mark.parametrize('cfg', [CHAINS[0], CHAINS[2], CHAINS[4], CHAINS[5], CHAINS[7]], ids=lambda g: 'x'.join(map(str, g)))    # (CHAINS[7]: the 7x7 cluster chain has an integer instance only)
@pytest.mark.parametrize('mode', ['requant_float=1', 'bias_near_2^31'])
def test_stage_chain_float_requant_instances(dev, cfg, mode):
    """The float-converter requantisation (option `requant_float = 1`; the default plans the INTEGER form — shift / round-half-even / clamp,
    fix_quant_ops.py:99-112; no float instruction — in every instance) is planned only where it is provably exact; a bias next to 2^31 makes
    body.0 / body.2 accumulators unboundable (f8_net.cpp conv_acc_bounded): the reference's `v + 2^(n-1)` wraps there (the value turns
    negative, the clamp makes it 0) and the launch must take the integer instance by itself — and equal the oracle either way."""
    for tail in ('int32_out', 'int8_out'):
        _run_stage_chain(dev, cfg, 'acc_shifts_left', tail, options={'requant_float': 1},      # (the default, 0, is what every other test of this file runs)
                         big_bias=(mode != 'requant_float=1'))


def test_stage_chain_whose_stream_the_planner_cannot_bound(dev):
    """The int32 stream is requantised through the float converter only while the planner can bound it (f8_net.cpp tensor_amax: the shifted sum of
    the blocks' accumulator bounds).  A body.4 bias next to 2^31 leaves body.0 / body.2 bounded and the STREAM unbounded: its `v + 2^(n-1)` wraps in
    the reference (the value turns negative, the clamp makes it 0 where the float form would saturate to 255) — the launch must take the integer
    instance by itself and equal the oracle."""
    for tail in ('int32_out', 'int8_out'):
        _run_stage_chain(dev, CHAINS[5], 'acc_shifts_left', tail, big_bias='stream')


def _run_stage_chain(dev, cfg, variant, tail, options=None, big_bias=False):
    C, MID, HW, nblk, cin0, N = cfg
    blocks, fls = _stage(C, MID, nblk, cin0, variant)
    convs = [c for b in blocks for c in (b.body + ([b.shortcut] if b.shortcut is not None else []))]
    tailc = topology.ConvSpec('tail.0', C, 64, 1, 1, 0)
    tail2 = topology.ConvSpec('tail.1', C, 32, 1, 1, 0)
    fls['tail.0'], fls['tail.1'] = (3, 6), (5, 7)
    tail2.signed_in = True
    params = _params(convs + [tailc, tail2], fls, 11, variant)
    if big_bias == 'stream':
        # channel C - 1 of the stream is a CONSTANT through the first two blocks (zero weights, biases only), chosen below so that after the second
        # block it sits inside the last 2^(n-1) values below 2^31: the next body.0's rounding add wraps for every pixel of that channel
        for k in ('s.0.body.4', 's.0.shortcut.0', 's.1.body.4'):
            params[k + '.weight'][C - 1] = 0
        params['s.0.body.4.bias'][C - 1], params['s.0.shortcut.0.bias'][C - 1] = 12345, 54321
    elif big_bias:    # accumulators within 2^(n-1) of 2^31 in body.0 and body.2 of the second block: the rounding add wraps
        params['s.1.body.0.bias'][[3, MID - 1]] = [2 ** 31 - 50, 2 ** 31 - 2 ** 12]
        params['s.1.body.2.bias'][[0, 17]] = [2 ** 31 - 2 ** 10, 2 ** 31 - 7]
    x_fl = 9
    scale = 3.0e3 if cin0 == C else 60.0
    x = synth.rand_normal_int(7, 'chainx' + variant, (N, cin0, HW, HW), scale).astype(np.int32)
    if cin0 == C:
        x.reshape(-1)[:3] = [2**31 - 1, -2**31, 2**30]       # wrap / clamp corners of the residual join
    else:
        x = np.abs(x)

    if big_bias == 'stream':
        w0, fl0 = oracle.block_forward(blocks[0], params, x, x_fl)
        z0 = int(w0[0, C - 1, 0, 0])
        assert (w0[:, C - 1] == z0).all() and z0 > 0
        acc_shl = fl0 - sum(fls['s.1.body.4'])                 # the stream keeps its fraclen, body.4's accumulator shifts left ('acc_shifts_left')
        half = 1 << (fl0 - fls['s.2.body.0'][0] - 1)
        assert acc_shl >= 0 and half >= 2 ** acc_shl
        target = 2 ** 31 - 1 - ((2 ** 31 - 1 - z0) % (1 << acc_shl))
        assert 2 ** 31 - half <= target < 2 ** 31
        params['s.1.body.4.bias'][C - 1] = (target - z0) >> acc_shl

    net = F8Net()
    for k, v in (options or {}).items():
        net.set_option(k, v)
    t = net.input(cin0, HW, HW, x_fl)
    r = t
    for b in blocks:
        xin = r
        for c in b.body:
            r = net.conv(r, params[c.key + '.weight'], params[c.key + '.bias'], stride=1, pad=c.pad, groups=1,
                         weight_fl=fls[c.key][1], input_fl=fls[c.key][0], input_signed=c.signed_in, quant_input=True, relu=c.relu)
        if b.shortcut is not None:
            c = b.shortcut
            sx = net.conv(xin, params[c.key + '.weight'], params[c.key + '.bias'], stride=1, pad=0, groups=1,
                          weight_fl=fls[c.key][1], input_fl=fls[c.key][0], input_signed=c.signed_in, quant_input=True, relu=False)
            r = net.add(r, sx, relu=True)
        else:
            r = net.add(r, xin, relu=True)
    if tail == 'int8_out':
        # two consumers with different int8 formats: the launch writes both forms of the stage output, no int32 form
        y0 = net.conv(r, params['tail.0.weight'], params['tail.0.bias'], stride=1, pad=0, groups=1, weight_fl=6, input_fl=3,
                      input_signed=False, quant_input=True, relu=False)
        net.output(y0, as_float=False)
    else:
        net.output(r, as_float=False)
    net.finalize(N)
    plan = net.describe()
    assert f'stage_chain_x{nblk}' in plan, plan
    got = net.run(_t(x, dev)).cpu().numpy()
    net.check()

    w, fl = x, x_fl
    for b in blocks:
        w, fl = oracle.block_forward(b, params, w, fl)
    if big_bias == 'stream':     # the stream channel sits in the wrap window in the oracle (and stays a plain positive int32 there)
        w1, _ = oracle.block_forward(blocks[1], params, w0, fl0)
        assert (w1[:, C - 1] == target).all()
    elif big_bias:    # the wrap really happens in the oracle: those mid channels requantise to 0 although their accumulators are huge
        w0, fl0 = oracle.block_forward(blocks[0], params, x, x_fl)
        m1, _ = oracle._conv_layer(blocks[1].body[0], params, w0, fl0)
        assert ((m1[:, 3] == 0) | (m1[:, 3] > 2 ** 30)).all() and (m1[:, 3] == 0).any() and (m1[:, 3] > 2 ** 30).any()   # some sums passed 2^31, wrapped, and the ReLU made them 0
    if tail == 'int8_out':
        w, fl = oracle._conv_layer(tailc, params, w, fl)
        got = got.reshape(N, 64, HW, HW)
    else:
        got = got.reshape(N, C, HW, HW)
    assert net.output_fraclen == fl
    np.testing.assert_array_equal(got, w)
    # a second run of the same handle (flags / ticket are re-armed per launch) and a smaller, ragged batch
    if N > 2:
        got2 = net.run(_t(x[:N - 1], dev)).cpu().numpy().reshape((N - 1,) + w.shape[1:])
        net.check()
        np.testing.assert_array_equal(got2, w[:N - 1])


# ... starting with the JOIN of the stage-opening block whose 3x3 and shortcut have stride 2 (fix_resnet.py:55-77; ResNet-50 stage 1): its body.0 +
# body.2 run on f8_opener.hip (P12: mid2 -> HBM, int8), its join (body.4 + 1x1 / 2 shortcut) is the first block of the chain launch (TAIL)
TAIL_CHAINS = [(512, 128, 28, 3, 256, 3), (512, 128, 28, 1, 256, 5), (512, 128, 28, 3, 256, 37),   # C, MID, H = W of the stage, identity blocks, CIN0, N
               (1024, 256, 14, 2, 512, 3), (1024, 256, 14, 5, 512, 70),          # ResNet-50 stage 2: the opener's convs are launches of their own, its dual-GEMM join opens the chain
               (2048, 512, 7, 2, 1024, 5), (2048, 512, 7, 1, 1024, 130)]        # stage 3: the same on the cluster kernel (f8_cchain.hip)


@pytest.mark.parametrize('cfg', TAIL_CHAINS, ids=lambda g: 'x'.join(map(str, g)))
@pytest.mark.parametrize('variant', ['body_shifts_left', 'shortcut_shifts_left', 'two_input_forms', 'requant_float=1', 'fuse_tail=0'])
def test_stage_chain_opened_by_a_stride2_block_matches_oracle(dev, cfg, variant):
    """`two_input_forms`: body.0 and the shortcut read the block input in DIFFERENT int8 formats — the planner does not fuse body.0 + body.2 on the
    opener kernel (pass 1d) but the dual-GEMM join still opens the chain; `fuse_tail=0`: the round-3 plan (whole opener in one launch / a dual-GEMM
    launch, identity chain behind it)."""
    C, MID, HW, nid, CIN0, N = cfg
    if N > 8 and variant not in ('body_shifts_left', 'requant_float=1'):
        pytest.skip('format variants are covered at the small batch')
    HWI = 2 * HW
    name = 'o.0'
    body = [topology.ConvSpec(name + '.body.0', CIN0, MID, 1, 1, 0, relu=True), topology.ConvSpec(name + '.body.2', MID, MID, 3, 2, 1, relu=True),
            topology.ConvSpec(name + '.body.4', MID, C, 1, 1, 0)]
    sc = topology.ConvSpec(name + '.shortcut.0', CIN0, C, 1, 2, 0)
    opener = topology.BlockSpec(name, body, sc, residual=True, post_relu=True)
    idb, fls = _stage(C, MID, nid, C, 'acc_shifts_left')
    if variant == 'shortcut_shifts_left':
        fls[name + '.body.0'], fls[name + '.body.2'], fls[name + '.body.4'], fls[name + '.shortcut.0'] = (4, 7), (3, 6), (4, 7), (4, 6)     # body.4 11 > shortcut 10
    elif variant == 'two_input_forms':
        fls[name + '.body.0'], fls[name + '.body.2'], fls[name + '.body.4'], fls[name + '.shortcut.0'] = (4, 7), (3, 6), (3, 6), (3, 7)     # body.0 reads fl 4, the shortcut fl 3
    else:
        fls[name + '.body.0'], fls[name + '.body.2'], fls[name + '.body.4'], fls[name + '.shortcut.0'] = (4, 7), (3, 6), (3, 6), (4, 7)     # body.4 9 < shortcut 11
    blocks = [opener] + idb
    convs = [c for b in blocks for c in b.body] + [sc]
    pre = topology.ConvSpec('pre.0', CIN0, CIN0, 1, 1, 0)          # a conv in front: the opener's input is not the network input
    fls['pre.0'] = (4, 7)
    params = _params(convs + [pre], fls, 51, variant)
    x_fl = 9
    x = synth.rand_normal_int(23, 'tailx' + variant, (N, CIN0, HWI, HWI), 3.0e3).astype(np.int32)

    net = F8Net()
    if variant == 'requant_float=1':
        net.set_option('requant_float', 1)
    if variant == 'fuse_tail=0':
        net.set_option('fuse_tail', 0)
    t = net.input(CIN0, HWI, HWI, x_fl)
    r = net.conv(t, params['pre.0.weight'], params['pre.0.bias'], stride=1, pad=0, groups=1, weight_fl=7, input_fl=4, input_signed=False,
                 quant_input=True, relu=True)
    for b in blocks:
        xin = r
        for c in b.body:
            r = net.conv(r, params[c.key + '.weight'], params[c.key + '.bias'], stride=c.stride, pad=c.pad, groups=1,
                         weight_fl=fls[c.key][1], input_fl=fls[c.key][0], input_signed=False, quant_input=True, relu=c.relu)
        if b.shortcut is not None:
            c = b.shortcut
            xin = net.conv(xin, params[c.key + '.weight'], params[c.key + '.bias'], stride=2, pad=0, groups=1,
                           weight_fl=fls[c.key][1], input_fl=fls[c.key][0], input_signed=False, quant_input=True, relu=False)
            r = net.add(r, xin, relu=True)
        else:
            r = net.add(r, xin, relu=True)
    net.output(r, as_float=False)
    net.finalize(N)
    plan = net.describe()
    if variant == 'fuse_tail=0':
        assert '_tail' not in plan and '_p12' not in plan, plan
        if nid >= 2:
            assert f'stage_chain_x{nid}:' in plan, plan
    else:
        assert f'stage_chain_x{nid + 1}_tail' in plan, plan
        assert ('fused_opener_s2_p12' in plan) == (HW == 28 and variant != 'two_input_forms'), plan
    got = net.run(_t(x, dev)).cpu().numpy().reshape(N, C, HW, HW)
    net.check()

    w, fl = oracle._conv_layer(pre, params, x, x_fl)
    w = np.maximum(w, 0)
    for b in blocks:
        w, fl = oracle.block_forward(b, params, w, fl)
    assert net.output_fraclen == fl
    np.testing.assert_array_equal(got, w)
    got2 = net.run(_t(x[:N - 1], dev)).cpu().numpy().reshape((N - 1,) + w.shape[1:])     # second run, ragged batch
    net.check()
    np.testing.assert_array_equal(got2, w[:N - 1])


# ---------------------------------------------------------------------------------------------------------------------------------
# BasicBlock chains (f8_bchain.hip): ResNet-18 / 34 identity blocks of one stage in one launch
def _basic_stage(C, nblk, variant):
    blocks, fls = [], {}
    for k in range(nblk):
        name = f'b.{k}'
        body = [topology.ConvSpec(name + '.body.0', C, C, 3, 1, 1, relu=True), topology.ConvSpec(name + '.body.2', C, C, 3, 1, 1)]
        blocks.append(topology.BlockSpec(name, body, None, residual=True, post_relu=True))
        if variant == 'acc_shifts_left':
            fls[name + '.body.0'], fls[name + '.body.2'] = (4, 7), (3, 5 + (k % 2))
        else:
            fls[name + '.body.0'], fls[name + '.body.2'] = (4, 7), (min(5 + k, 7), 7)
    return blocks, fls


BCHAINS = [(64, 56, 2, 3), (64, 56, 3, 40), (128, 28, 1, 5), (128, 28, 2, 70), (256, 14, 1, 4), (256, 14, 3, 131)]   # C, H = W, blocks, N


@pytest.mark.parametrize('cfg', BCHAINS, ids=lambda g: 'x'.join(map(str, g)))
@pytest.mark.parametrize('variant', ['acc_shifts_left', 'res_shifts_left'])
@pytest.mark.parametrize('tail', ['int32_out', 'int8_out'])
def test_basic_block_chain_matches_oracle(dev, cfg, variant, tail):
    if tail == 'int8_out' and cfg[3] > 8:
        pytest.skip('the int8 tails are covered at the small batch')
    _run_basic_chain(dev, cfg, variant, tail)


@pytest.mark.parametrize('cfg', [BCHAINS[0], (128, 28, 2, 5), (256, 14, 2, 4)], ids=lambda g: 'x'.join(map(str, g)))
@pytest.mark.parametrize('mode', ['requant_float=1', 'bias_near_2^31'])
def test_basic_block_chain_float_requant_instances(dev, cfg, mode):
    """As test_stage_chain_float_requant_instances, for bchain_kernel: the float-converter instance by option, the integer one by itself where the first
    conv's accumulators cannot be bounded (bias next to 2^31: the reference's rounding add wraps)."""
    for tail in ('int32_out', 'int8_out'):
        _run_basic_chain(dev, cfg, 'acc_shifts_left', tail, options={'requant_float': 1},      # (the default, 0, is what every other test of this file runs)
                         big_bias=(mode != 'requant_float=1'))


def _run_basic_chain(dev, cfg, variant, tail, options=None, big_bias=False):
    C, HW, nblk, N = cfg
    blocks, fls = _basic_stage(C, nblk, variant)
    convs = [c for b in blocks for c in b.body]
    t0 = topology.ConvSpec('tail.0', C, 64, 1, 1, 0)
    t1 = topology.ConvSpec('tail.1', C, 32, 1, 1, 0)
    fls['tail.0'], fls['tail.1'] = (3, 6), (5, 7)
    params = _params(convs + [t0, t1], fls, 31, variant)
    x_fl = 9
    x = synth.rand_normal_int(17, 'bchainx' + variant, (N, C, HW, HW), 3.0e3).astype(np.int32)
    x.reshape(-1)[:3] = [2**31 - 1, -2**31, 2**30]
    pre = topology.ConvSpec('pre.0', C, C, 1, 1, 0)              # a conv in front: the chain's input is not the network input
    fls['pre.0'] = (4, 7)
    params.update(_params([pre], fls, 33, variant))
    if big_bias:
        params[f'b.{nblk - 1}.body.0.bias'][[2, C - 1]] = [2 ** 31 - 50, 2 ** 31 - 2 ** 12]

    net = F8Net()
    for k, v in (options or {}).items():
        net.set_option(k, v)
    t = net.input(C, HW, HW, x_fl)
    r = net.conv(t, params['pre.0.weight'], params['pre.0.bias'], stride=1, pad=0, groups=1, weight_fl=7, input_fl=4, input_signed=False,
                 quant_input=True, relu=True)
    for b in blocks:
        xin = r
        for c in b.body:
            r = net.conv(r, params[c.key + '.weight'], params[c.key + '.bias'], stride=1, pad=c.pad, groups=1,
                         weight_fl=fls[c.key][1], input_fl=fls[c.key][0], input_signed=False, quant_input=True, relu=c.relu)
        r = net.add(r, xin, relu=True)
    if tail == 'int8_out':
        # the stage output is only read as int8 (the second format of a downsample block's body.0 / shortcut.0 pair is covered by the
        # whole-net ResNet-18 goldens: its stage-1 / stage-2 chains write two int8 forms)
        y0 = net.conv(r, params['tail.0.weight'], params['tail.0.bias'], stride=1, pad=0, groups=1, weight_fl=6, input_fl=3,
                      input_signed=False, quant_input=True, relu=False)
        net.output(y0, as_float=False)
    else:
        net.output(r, as_float=False)
    net.finalize(N)
    plan = net.describe()
    assert f'basic_chain_x{nblk}' in plan, plan
    got = net.run(_t(x, dev)).cpu().numpy()
    net.check()

    w, fl = oracle._conv_layer(pre, params, x, x_fl)
    w = np.maximum(w, 0)
    for b in blocks:
        w, fl = oracle.block_forward(b, params, w, fl)
    if tail == 'int8_out':
        w, fl = oracle._conv_layer(t0, params, w, fl)
        got = got.reshape(N, 64, HW, HW)
    else:
        got = got.reshape(N, C, HW, HW)
    assert net.output_fraclen == fl
    np.testing.assert_array_equal(got, w)
    if N > 2:
        got2 = net.run(_t(x[:N - 1], dev)).cpu().numpy().reshape((N - 1,) + w.shape[1:])
        net.check()
        np.testing.assert_array_equal(got2, w[:N - 1])


# ... starting with the stage-opening block: 3x3 / 2 (ReLU) -> 3x3, 1x1 / 2 shortcut, join (fix_resnet.py:55-77), then identity blocks
BCHAINS_DS = [(128, 28, 1, 3), (128, 28, 2, 70), (256, 14, 1, 5), (256, 14, 3, 131)]   # C, H = W of the stage, identity blocks, N


@pytest.mark.parametrize('cfg', BCHAINS_DS, ids=lambda g: 'x'.join(map(str, g)))
@pytest.mark.parametrize('variant', ['one_input_form', 'two_input_forms', 'signed_mid'])
def test_basic_block_chain_with_opening_block_matches_oracle(dev, cfg, variant):
    """The opening block reads the previous stage's output at twice the resolution: its 3x3 / 2 and its shortcut each in their own
    int8 format (`two_input_forms`: different fraclens, the shortcut's shifted left in the join; `one_input_form`: the same buffer,
    the second conv's result shifted left); `signed_mid`: a symmetric (signed) input of the second conv — the kernel's general path."""
    C, HW, nid, N = cfg
    CIN, HWI = C // 2, 2 * HW
    name = 'd.0'
    body = [topology.ConvSpec(name + '.body.0', CIN, C, 3, 2, 1, relu=True),
            topology.ConvSpec(name + '.body.2', C, C, 3, 1, 1, signed_in=(variant == 'signed_mid'))]
    sc = topology.ConvSpec(name + '.shortcut.0', CIN, C, 1, 2, 0)
    opener = topology.BlockSpec(name, body, sc, residual=True, post_relu=True)
    idb, fls = _basic_stage(C, nid, 'acc_shifts_left' if variant != 'two_input_forms' else 'res_shifts_left')
    if variant == 'two_input_forms':
        fls[name + '.body.0'], fls[name + '.body.2'], fls[name + '.shortcut.0'] = (4, 7), (4, 7), (3, 6)     # shortcut 9 < body 11
    else:
        fls[name + '.body.0'], fls[name + '.body.2'], fls[name + '.shortcut.0'] = (4, 7), (3, 6), (4, 7)     # body 9 < shortcut 11
    blocks = [opener] + idb
    convs = [c for b in blocks for c in b.body] + [sc]
    pre = topology.ConvSpec('pre.0', CIN, CIN, 1, 1, 0)
    fls['pre.0'] = (4, 7)
    params = _params(convs + [pre], fls, 41, variant)
    x_fl = 9
    x = synth.rand_normal_int(19, 'bchainds' + variant, (N, CIN, HWI, HWI), 3.0e3).astype(np.int32)

    net = F8Net()
    t = net.input(CIN, HWI, HWI, x_fl)
    r = net.conv(t, params['pre.0.weight'], params['pre.0.bias'], stride=1, pad=0, groups=1, weight_fl=7, input_fl=4, input_signed=False,
                 quant_input=True, relu=True)
    for b in blocks:
        xin = r
        for c in b.body:
            r = net.conv(r, params[c.key + '.weight'], params[c.key + '.bias'], stride=c.stride, pad=c.pad, groups=1,
                         weight_fl=fls[c.key][1], input_fl=fls[c.key][0], input_signed=c.signed_in, quant_input=True, relu=c.relu)
        if b.shortcut is not None:
            c = b.shortcut
            xin = net.conv(xin, params[c.key + '.weight'], params[c.key + '.bias'], stride=2, pad=0, groups=1,
                           weight_fl=fls[c.key][1], input_fl=fls[c.key][0], input_signed=False, quant_input=True, relu=False)
        r = net.add(r, xin, relu=True)
    net.output(r, as_float=False)
    net.finalize(N)
    plan = net.describe()
    assert f'basic_chain_x{nid + 1}_ds' in plan, plan
    assert sum(k in plan for k in ('conv3x3', 'conv1x1s2', '_res:')) == 0, plan     # every conv of the stage runs inside the chain launch
    got = net.run(_t(x, dev)).cpu().numpy().reshape(N, C, HW, HW)
    net.check()

    w, fl = oracle._conv_layer(pre, params, x, x_fl)
    w = np.maximum(w, 0)
    for b in blocks:
        w, fl = oracle.block_forward(b, params, w, fl)
    assert net.output_fraclen == fl
    np.testing.assert_array_equal(got, w)
    got2 = net.run(_t(x[:N - 1], dev)).cpu().numpy().reshape((N - 1,) + w.shape[1:])
    net.check()
    np.testing.assert_array_equal(got2, w[:N - 1])
