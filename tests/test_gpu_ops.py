"""GPU parity, op level: HIP kernels (through the C ABI) vs the CPU oracle. Bit-exact."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from f8net_amd import synth
from oracle import oracle


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    from f8net_amd import _lib
    assert _lib.lib().f8_device_count() >= 1
    return torch.device('cuda:0')


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_requant_matches_oracle(dev):
    from f8net_amd import ops
    edge = np.array([2**31 - 1, -2**31, -2**31 + 1, 2**30, -2**30, 255, 256, 127, 128, -127, -128, 0, 1, -1],
                    dtype=np.int32)
    v = np.concatenate([edge, synth.rand_normal_int(5, 'rq', (100000,), 40000.0).astype(np.int32),
                        synth.rand_uniform_int(6, 'rq2', (50000,), -2**31, 2**31 - 1).astype(np.int32)])
    x = _t(v, dev)
    for signed in (True, False):
        for dst in range(0, 8 if signed else 9):
            for src in (0, 4, 6, 7, 9, 12, 15, 21, 30):
                if src - dst > 30:
                    continue
                got = ops.int_op_only_fix_quant(x, 8, dst, src, signed)
                assert got.output_fraclen == dst and got.dtype == torch.int32
                np.testing.assert_array_equal(got.cpu().numpy(), oracle.requant(v, dst, src, signed),
                                              err_msg=f'{signed=} {dst=} {src=}')


def test_requant_asserts_like_reference(dev):
    from f8net_amd import ops
    x = torch.zeros(8, dtype=torch.int32, device=dev)
    with pytest.raises(AssertionError):
        ops.int_op_only_fix_quant(x, 8, 8, 10, True)
    with pytest.raises(AssertionError):
        ops.int_op_only_fix_quant(x, 8, 9, 10, False)
    with pytest.raises(AssertionError):
        ops.int_op_only_fix_quant(x, 8, 3.0, 10, False)
    with pytest.raises(ValueError):
        ops.int_op_only_fix_quant(torch.zeros(8, dtype=torch.int32), 8, 3, 10, False)   # CPU tensor: no CPU path
    assert ops.int_op_only_fix_quant(x[:0], 8, 3, 10, False).numel() == 0                # empty input


def test_relu_add_align(dev):
    from f8net_amd import ops
    a = synth.rand_normal_int(41, 'ra', (3, 7, 5, 5), 4e8).astype(np.int32)
    b = synth.rand_normal_int(42, 'rb', (3, 7, 5, 5), 4e8).astype(np.int32)
    a.reshape(-1)[:4] = [2**31 - 1, -2**31, 2**30, -2**30]
    b.reshape(-1)[:4] = [1, -1, 2**30, -2**30]
    for rf, xf in ((11, 9), (9, 12), (10, 10), (3, 20)):
        res = _t(a.copy(), dev)
        got, fl = ops.add_align_(res, _t(b, dev), rf, xf)
        want, wfl = oracle.add_align(a, b, rf, xf)
        assert fl == wfl == max(rf, xf) and got.output_fraclen == fl
        np.testing.assert_array_equal(got.cpu().numpy(), want)
    r = ops.relu_(_t(a.copy(), dev))
    np.testing.assert_array_equal(r.cpu().numpy(), np.maximum(a, 0))


# (N, C, H, W, K, k, stride, pad, groups, signed_in): every geometry class of SURVEY.md App. C at reduced size
CONV_GEOMS = [
    (2, 3, 32, 32, 64, 7, 2, 3, 1, False),     # ResNet stem, u8 input
    (2, 3, 30, 34, 64, 7, 2, 3, 1, True),      # ResNet stem, normalised (signed) input, ragged size
    (3, 3, 32, 32, 32, 3, 2, 1, 1, False),     # MobileNet stem
    (2, 64, 14, 14, 64, 1, 1, 0, 1, False),    # 1x1
    (2, 256, 14, 14, 64, 1, 1, 0, 1, False),   # 1x1 reduce
    (2, 64, 14, 14, 256, 1, 1, 0, 1, False),   # 1x1 expand
    (2, 64, 14, 14, 64, 3, 1, 1, 1, False),    # 3x3
    (2, 128, 14, 14, 128, 3, 2, 1, 1, False),  # 3x3 stride 2
    (2, 256, 14, 14, 512, 1, 2, 0, 1, False),  # strided 1x1 shortcut
    (1, 512, 7, 7, 512, 3, 1, 1, 1, False),    # 7x7 stage
    (5, 64, 9, 11, 64, 3, 1, 1, 1, False),     # ragged M (not a tile multiple)
    (2, 16, 12, 12, 96, 1, 1, 0, 1, True),     # MBV2 expand, signed input, C=16 (padded to 32)
    (2, 144, 8, 8, 24, 1, 1, 0, 1, False),     # MBV2 project 144->24
    (2, 96, 7, 7, 160, 1, 1, 0, 1, False),     # cout 160 (partial 128-wide tile)
    (2, 32, 12, 12, 32, 3, 1, 1, 32, False),   # depthwise
    (2, 144, 9, 9, 144, 3, 2, 1, 144, False),  # depthwise stride 2, C=144
    (2, 24, 8, 8, 24, 3, 1, 1, 24, True),      # depthwise signed input
    (130, 64, 4, 4, 64, 1, 1, 0, 1, False),    # many images, tiny maps
    # shapes served by the LDS-patch 3x3 kernel (f8_conv3x3.hip)
    (3, 64, 56, 56, 64, 3, 1, 1, 1, False),    # 2 rows per tile
    (2, 128, 28, 28, 128, 3, 1, 1, 1, True),   # 4 rows per tile, signed input (single bias class)
    (3, 256, 14, 14, 256, 3, 1, 1, 1, False),  # half an image per tile
    (2, 256, 14, 14, 96, 3, 1, 1, 1, False),   # cout 96: partial 128-wide cout tile
    (5, 512, 7, 7, 512, 3, 1, 1, 1, False),    # two images per tile, odd image count
    (2, 64, 28, 56, 192, 3, 1, 1, 1, False),   # H != W
]


@pytest.mark.parametrize('geom', CONV_GEOMS, ids=lambda g: 'x'.join(map(str, g)))
def test_conv_matches_oracle(dev, geom):
    from f8net_amd import ops
    N, C, H, W, K, k, s, p, g, signed = geom
    lo, hi = (-127, 127) if signed else (0, 255)
    x = synth.rand_uniform_int(11, f'x{geom}', (N, C, H, W), lo, hi).astype(np.int32)
    w = synth.rand_uniform_int(12, f'w{geom}', (K, C // g, k, k), -127, 127).astype(np.int32)
    b = synth.rand_normal_int(13, f'b{geom}', (K,), 3e5).astype(np.int32)
    conv = ops.F8Conv2d(C, K, k, stride=s, padding=p, groups=g, input_symmetric=signed)
    conv.weight.data = torch.from_numpy(w)
    conv.bias.data = torch.from_numpy(b)
    conv.input_fraclen.fill_(5)
    conv.weight_fraclen.fill_(6)
    got = conv(_t(x, dev)).cpu().numpy()
    want = oracle.conv2d(x, w, b, s, p, g)
    assert got.shape == want.shape
    np.testing.assert_array_equal(got, want)


def test_conv_extreme_values_wrap(dev):
    """All-extreme operands: the offset trick and the accumulator must stay exact (mod 2^32)."""
    from f8net_amd import ops
    N, C, H, W, K = 1, 512, 6, 6, 64
    x = np.full((N, C, H, W), 255, dtype=np.int32)
    w = np.full((K, C, 3, 3), -127, dtype=np.int32)
    w[1::2] = 127
    b = np.full((K,), 2**31 - 1, dtype=np.int32)
    b[::3] = -2**31
    conv = ops.F8Conv2d(C, K, 3, stride=1, padding=1)
    conv.weight.data, conv.bias.data = torch.from_numpy(w), torch.from_numpy(b)
    got = conv(_t(x, dev)).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.conv2d(x, w, b, 1, 1))


def test_linear_pools(dev):
    from f8net_amd import ops
    x = synth.rand_uniform_int(21, 'lx', (7, 2048), 0, 255).astype(np.int32)
    w = synth.rand_uniform_int(22, 'lw', (1000, 2048), -127, 127).astype(np.int32)
    b = synth.rand_normal_int(23, 'lb', (1000,), 1e6).astype(np.int32)
    fc = ops.F8Linear(2048, 1000)
    fc.weight.data, fc.bias.data = torch.from_numpy(w), torch.from_numpy(b)
    got = fc(_t(x, dev)).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.linear(x, w, b))

    a = synth.rand_normal_int(31, 'ap', (3, 96, 7, 7), 3e7).astype(np.int32)
    t = _t(a, dev)
    setattr(t, 'output_fraclen', 13)
    r = ops.FXQAvgPool2d(7)(t)
    assert r.output_fraclen == 19
    np.testing.assert_array_equal(r.cpu().numpy(), oracle.avgpool_sum(a))

    m = np.maximum(synth.rand_normal_int(32, 'mp', (2, 64, 17, 18), 2e5), 0).astype(np.int32)
    got = ops.F8MaxPool2d(3, 2, 1)(_t(m, dev)).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.maxpool(m))


# (C, MID, H, W, N): shapes the fused bottleneck kernel is instantiated for, at small batch
FUSED_SHAPES = [(256, 64, 56, 56, 2), (512, 128, 28, 28, 3), (256, 64, 4, 56, 1)]


@pytest.mark.parametrize('shape', FUSED_SHAPES, ids=lambda g: 'x'.join(map(str, g)))
@pytest.mark.parametrize('variant', ['res_shifts_left', 'acc_shifts_left_signed_mid'])
def test_fused_bottleneck_block_matches_oracle(dev, shape, variant):
    """One bottleneck identity block as a net of its own: the planner maps it onto
    fused_bottleneck_kernel; result (int32 block output) must equal IntBlock.forward's (oracle)."""
    from f8net_amd import topology
    from f8net_amd.net import F8Net
    C, MID, H, W, N = shape
    b = topology.BlockSpec('blk', [topology.ConvSpec('blk.body.0', C, MID, 1, 1, 0, relu=True),
                                   topology.ConvSpec('blk.body.2', MID, MID, 3, 1, 1, relu=True),
                                   topology.ConvSpec('blk.body.4', MID, C, 1, 1, 0)], None, residual=True, post_relu=True)
    if variant == 'res_shifts_left':
        fls, x_fl = {'blk.body.0': (4, 7), 'blk.body.2': (3, 6), 'blk.body.4': (5, 7)}, 9          # out fl 12 > x fl 9
    else:
        fls, x_fl = {'blk.body.0': (2, 5), 'blk.body.2': (4, 7), 'blk.body.4': (1, 6)}, 11         # out fl 7 < x fl 11
        b.body[1].signed_in = True        # exercise the plain (unbiased) LDS patch and its zero border
    params = {}
    for c in b.body:
        in_fl, w_fl = fls[c.key]
        params[c.key + '.weight'] = np.clip(synth.rand_normal_int(5, c.key + 'w' + variant, (c.cout, c.cin, c.k, c.k), 30.0), -127, 127).astype(np.int32)
        params[c.key + '.bias'] = synth.rand_normal_int(6, c.key + 'b', (c.cout,), 2.0 ** (in_fl + w_fl)).astype(np.int32)
        params[c.key + '.weight_fraclen'] = np.array(w_fl, np.int32)
        params[c.key + '.input_fraclen'] = np.array([in_fl], np.int32)
    x = synth.rand_normal_int(7, 'blkx' + variant, (N, C, H, W), 3.0e3 if variant == 'res_shifts_left' else 4.0e5).astype(np.int32)
    x.reshape(-1)[:3] = [2**31 - 1, -2**31, 2**30]       # wrap / clamp corners of the residual join
    net = F8Net()
    t = net.input(C, H, W, x_fl)
    r = t
    for c in b.body:
        r = net.conv(r, params[c.key + '.weight'], params[c.key + '.bias'], stride=1, pad=c.pad, groups=1,
                     weight_fl=fls[c.key][1], input_fl=fls[c.key][0], input_signed=c.signed_in, quant_input=True, relu=c.relu)
    r = net.add(r, t, relu=True)
    net.output(r, as_float=False)
    net.finalize(N)
    if W in (56, 28) and H % (2 if W == 56 else 4) == 0:
        assert 'fused_bottleneck' in net.describe(), net.describe()
    got = net.run(_t(x, dev)).cpu().numpy().reshape(N, C, H, W)
    want, want_fl = oracle.block_forward(b, params, x, x_fl)
    assert net.output_fraclen == want_fl
    np.testing.assert_array_equal(got, want)


DUAL_SHAPES = [(64, 64, 256, 1, 24, 24, 3), (256, 128, 512, 2, 28, 28, 2), (128, 64, 96, 2, 15, 15, 2)]   # Cin, MID, Cout, stride, H, W, N


@pytest.mark.parametrize('shape', DUAL_SHAPES, ids=lambda g: 'x'.join(map(str, g)))
@pytest.mark.parametrize('variant', ['body_shifts_left', 'shortcut_shifts_left'])
def test_downsample_block_dual_gemm_matches_oracle(dev, shape, variant):
    """A bottleneck downsample block (fix_resnet.py:26-77 with a shortcut conv): body.4 and shortcut.0 are planned as
    ONE dual-GEMM launch; the int32 block output must equal the oracle's IntBlock.forward, wrap and clamp included."""
    from f8net_amd import topology
    from f8net_amd.net import F8Net
    Cin, MID, Cout, stride, H, W, N = shape
    body = [topology.ConvSpec('blk.body.0', Cin, MID, 1, 1, 0, relu=True),
            topology.ConvSpec('blk.body.2', MID, MID, 3, stride, 1, relu=True),
            topology.ConvSpec('blk.body.4', MID, Cout, 1, 1, 0)]
    sc = topology.ConvSpec('blk.shortcut.0', Cin, Cout, 1, stride, 0)
    b = topology.BlockSpec('blk', body, sc, residual=True, post_relu=True)
    if variant == 'body_shifts_left':
        fls = {'blk.body.0': (4, 7), 'blk.body.2': (3, 6), 'blk.body.4': (3, 5), 'blk.shortcut.0': (5, 7)}     # 8 vs 12
    else:
        fls = {'blk.body.0': (4, 7), 'blk.body.2': (3, 6), 'blk.body.4': (6, 7), 'blk.shortcut.0': (4, 6)}     # 13 vs 10
    x_fl = 9
    params = {}
    for c in body + [sc]:
        in_fl, w_fl = fls[c.key]
        params[c.key + '.weight'] = np.clip(synth.rand_normal_int(15, c.key + 'w' + variant, (c.cout, c.cin, c.k, c.k), 50.0), -127, 127).astype(np.int32)
        params[c.key + '.bias'] = synth.rand_normal_int(16, c.key + 'b', (c.cout,), 2.0 ** 27).astype(np.int32)   # joins wrap / clamp
        params[c.key + '.weight_fraclen'] = np.array(w_fl, np.int32)
        params[c.key + '.input_fraclen'] = np.array([in_fl], np.int32)
    x = synth.rand_normal_int(17, 'dsx' + variant, (N, Cin, H, W), 2.0e3).astype(np.int32)
    net = F8Net()
    t = net.input(Cin, H, W, x_fl)
    r = t
    for c in body:
        r = net.conv(r, params[c.key + '.weight'], params[c.key + '.bias'], stride=c.stride, pad=c.pad, groups=1,
                     weight_fl=fls[c.key][1], input_fl=fls[c.key][0], input_signed=c.signed_in, quant_input=True, relu=c.relu)
    s = net.conv(t, params[sc.key + '.weight'], params[sc.key + '.bias'], stride=stride, pad=0, groups=1,
                 weight_fl=fls[sc.key][1], input_fl=fls[sc.key][0], input_signed=sc.signed_in, quant_input=True, relu=False)
    r = net.add(r, s, relu=True)
    net.output(r, as_float=False)
    net.finalize(N)
    plan = net.describe()
    assert ('_dual:' in plan) == (Cin % 64 == 0 and MID % 64 == 0 and Cout > 32), plan
    P, Q = (H - 1) // stride + 1, (W - 1) // stride + 1
    got = net.run(_t(x, dev)).cpu().numpy().reshape(N, Cout, P, Q)
    want, want_fl = oracle.block_forward(b, params, x, x_fl)
    assert net.output_fraclen == want_fl
    np.testing.assert_array_equal(got, want)
    assert (np.abs(want.astype(np.int64)) > 2**30).any()       # the join ran at full int32 width (wrap-around arithmetic)


WREG_SHAPES = [(512, 256, 1024, 12, 3), (1024, 512, 2048, 14, 2), (1024, 512, 2048, 6, 5), (1024, 512, 2048, 28, 1)]   # Cin, MID, Cout, H, N


@pytest.mark.parametrize('shape', WREG_SHAPES, ids=lambda g: 'x'.join(map(str, g)))
@pytest.mark.parametrize('signed_mid', [False, True])
def test_weights_in_registers_1x1_matches_oracle(dev, shape, signed_mid):
    """body.0 of the stage-2 / stage-3 opening blocks runs on conv1x1_wreg_kernel (f8_wreg.hip): pixel counts that are not a
    multiple of the 64-pixel tile, grids below one workgroup per XCD, both int8 output conventions.  Three runs each: the kernel's
    counted vmcnt wait once returned with a DMA in flight (scheduler-hoisted weight loads), which showed as sporadic mismatches."""
    from f8net_amd import topology
    from f8net_amd.net import F8Net
    Cin, MID, Cout, H, N = shape
    body = [topology.ConvSpec('blk.body.0', Cin, MID, 1, 1, 0, relu=not signed_mid),
            topology.ConvSpec('blk.body.2', MID, MID, 3, 2, 1, relu=True),
            topology.ConvSpec('blk.body.4', MID, Cout, 1, 1, 0)]
    body[1].signed_in = signed_mid
    sc = topology.ConvSpec('blk.shortcut.0', Cin, Cout, 1, 2, 0)
    b = topology.BlockSpec('blk', body, sc, residual=True, post_relu=True)
    fls = {'blk.body.0': (4, 7), 'blk.body.2': (3, 6), 'blk.body.4': (3, 5), 'blk.shortcut.0': (5, 7)}
    x_fl = 9
    params = {}
    for c in body + [sc]:
        in_fl, w_fl = fls[c.key]
        params[c.key + '.weight'] = np.clip(synth.rand_normal_int(25, c.key + 'w', (c.cout, c.cin, c.k, c.k), 40.0), -127, 127).astype(np.int32)
        params[c.key + '.bias'] = synth.rand_normal_int(26, c.key + 'b', (c.cout,), 2.0 ** 16).astype(np.int32)
        params[c.key + '.weight_fraclen'] = np.array(w_fl, np.int32)
        params[c.key + '.input_fraclen'] = np.array([in_fl], np.int32)
    x = synth.rand_normal_int(27, f'wregx{shape}', (N, Cin, H, H), 2.0e3).astype(np.int32)
    net = F8Net()
    t = net.input(Cin, H, H, x_fl)
    r = t
    for c in body:
        r = net.conv(r, params[c.key + '.weight'], params[c.key + '.bias'], stride=c.stride, pad=c.pad, groups=1,
                     weight_fl=fls[c.key][1], input_fl=fls[c.key][0], input_signed=c.signed_in, quant_input=True, relu=c.relu)
    s_ = net.conv(t, params[sc.key + '.weight'], params[sc.key + '.bias'], stride=2, pad=0, groups=1,
                  weight_fl=fls[sc.key][1], input_fl=fls[sc.key][0], input_signed=False, quant_input=True, relu=False)
    r = net.add(r, s_, relu=True)
    net.output(r, as_float=False)
    net.finalize(N)
    assert 'conv1x1s1_wreg:' in net.describe(), net.describe()
    want, want_fl = oracle.block_forward(b, params, x, x_fl)
    assert net.output_fraclen == want_fl
    xd = _t(x, dev)
    for _ in range(3):
        got = net.run(xd).cpu().numpy().reshape(want.shape)
        np.testing.assert_array_equal(got, want)


# Cin, MID, Cout, H, N, identity block?  (stage-2 opener, stage-3 opener at two sizes, 7x7 identity block, a one-tile launch)
WSTAT_SHAPES = [(512, 256, 1024, 12, 3, False), (1024, 512, 2048, 14, 2, False), (1024, 512, 2048, 6, 5, False),
                (2048, 512, 2048, 7, 5, True), (2048, 512, 2048, 7, 1, True), (512, 256, 1024, 4, 1, False),
                (512, 256, 1024, 28, 3, False), (1024, 512, 2048, 14, 5, False)]     # full-size maps: body.2 on conv3x3s2_wreg_kernel


@pytest.mark.parametrize('shape', WSTAT_SHAPES, ids=lambda g: 'x'.join(map(str, g)))
@pytest.mark.parametrize('variant', ['body_shifts_left', 'other_shifts_left_signed_mid'])
@pytest.mark.parametrize('tail', ['i32', 'both', 'two_formats'])
def test_weight_stationary_1x1_matches_oracle(dev, shape, variant, tail):
    """conv1x1_wstat_kernel (f8_wstat.hip), forced onto small launches with wstat_min_tiles = 0: body.0 of a stage-opening block
    (plain instance, K = 512 / 1024), its body.4 + strided shortcut as ONE dual GEMM (K = 512 + 256 on 8 waves, 1024 + 512 on 4),
    and body.4 of a 7x7 identity block with the int32 residual join; int32-only and int32 + int8 outputs, both align directions,
    pixel counts that are not a multiple of the 32-pixel tile, fewer tiles than workgroups.  Three runs each (counted vmcnt ring)."""
    from f8net_amd import topology
    from f8net_amd.net import F8Net
    Cin, MID, Cout, H, N, identity = shape
    st = 1 if identity else 2
    body = [topology.ConvSpec('blk.body.0', Cin, MID, 1, 1, 0, relu=True),
            topology.ConvSpec('blk.body.2', MID, MID, 3, st, 1, relu=True),
            topology.ConvSpec('blk.body.4', MID, Cout, 1, 1, 0)]
    sc = None if identity else topology.ConvSpec('blk.shortcut.0', Cin, Cout, 1, 2, 0)
    b = topology.BlockSpec('blk', body, sc, residual=True, post_relu=True)
    x_fl = 9
    fls = {'blk.body.0': (4, 7), 'blk.body.2': (3, 6), 'blk.body.4': (3, 5), 'blk.shortcut.0': (4, 7)}       # body out fl 8 < 11 / 9
    if variant == 'other_shifts_left_signed_mid':
        fls.update({'blk.body.4': (6, 7), 'blk.shortcut.0': (4, 6)})                                             # 13 > 10 / 9
        body[1].signed_in = True
        body[0].relu = False
    params = {}
    for c in body + ([sc] if sc else []):
        in_fl, w_fl = fls[c.key]
        params[c.key + '.weight'] = np.clip(synth.rand_normal_int(55, c.key + 'w' + variant, (c.cout, c.cin, c.k, c.k), 40.0), -127, 127).astype(np.int32)
        params[c.key + '.bias'] = synth.rand_normal_int(56, c.key + 'b', (c.cout,), 2.0 ** 27 if c.key.endswith(('body.4', 'shortcut.0')) else 2.0 ** 14).astype(np.int32)
        params[c.key + '.weight_fraclen'] = np.array(w_fl, np.int32)
        params[c.key + '.input_fraclen'] = np.array([in_fl], np.int32)
    x = synth.rand_normal_int(57, f'wsx{shape}{variant}', (N, Cin, H, H), 2.0e3 if not identity else 2.0 ** 27).astype(np.int32)
    net = F8Net()
    net.set_option('wstat_min_tiles', 0)
    if tail == 'two_formats':
        net.set_option('wstat_fast', 0)                        # the general epilogue (either shift direction, explicit ReLU floor)
    t = net.input(Cin, H, H, x_fl)
    r = t
    for c in body:
        r = net.conv(r, params[c.key + '.weight'], params[c.key + '.bias'], stride=c.stride, pad=c.pad, groups=1,
                     weight_fl=fls[c.key][1], input_fl=fls[c.key][0], input_signed=c.signed_in, quant_input=True, relu=c.relu)
    if sc:
        s_ = net.conv(t, params[sc.key + '.weight'], params[sc.key + '.bias'], stride=2, pad=0, groups=1,
                      weight_fl=fls[sc.key][1], input_fl=fls[sc.key][0], input_signed=False, quant_input=True, relu=False)
        r = net.add(r, s_, relu=True)
    else:
        r = net.add(r, t, relu=True)
    want, want_fl = oracle.block_forward(b, params, x, x_fl)
    assert (np.abs(want.astype(np.int64)) > 2**30).any()       # the join ran at full int32 width
    if tail == 'both':                                         # a consumer of the int8 copy, joined with the int32 stream
        wt = np.clip(synth.rand_normal_int(58, 'tailw', (Cout, Cout, 1, 1), 40.0), -127, 127).astype(np.int32)
        bt = synth.rand_normal_int(59, 'tailb', (Cout,), 1.0e4).astype(np.int32)
        in_fl_t, w_fl_t = 2, 6
        c2 = net.conv(r, wt, bt, stride=1, pad=0, groups=1, weight_fl=w_fl_t, input_fl=in_fl_t, input_signed=False, quant_input=True, relu=False)
        y = oracle.conv2d(oracle.requant(want, in_fl_t, want_fl, False), wt, bt, 1, 0)
        r = net.add(c2, r, relu=False)
        want, _ = oracle.add_align(y, want, in_fl_t + w_fl_t, want_fl)
    if tail == 'two_formats':                                  # two consumers that read the block output in DIFFERENT int8 formats
        ys = []
        cs = []
        for k, in_fl_t in enumerate((2, 3)):
            wt = np.clip(synth.rand_normal_int(60 + k, 'tailw2', (64, Cout, 1, 1), 40.0), -127, 127).astype(np.int32)
            cs.append(net.conv(r, wt, None, stride=1, pad=0, groups=1, weight_fl=6 - k, input_fl=in_fl_t, input_signed=bool(k), quant_input=True, relu=False))
            ys.append(oracle.conv2d(oracle.requant(want, in_fl_t, want_fl, bool(k)), wt, np.zeros(64, np.int32), 1, 0))
        r = net.add(cs[0], cs[1], relu=False)                  # both products have fraclen 8
        want, _ = oracle.add_align(ys[0], ys[1], 8, 8)
    net.output(r, as_float=False)
    net.finalize(N)
    plan = net.describe()
    if tail == 'two_formats':
        assert any('i8=2' in l for l in plan.splitlines() if 'wstat_dual' in l or 'wstat_res' in l), plan
    if identity:
        assert 'conv1x1_wstat_res:' in plan, plan
    else:
        assert 'conv1x1_wstat:' in plan and 'conv1x1_wstat_dual:' in plan, plan
        # the stride-2 3x3 of the real stage-2 / stage-3 openers: input patch in LDS, weights streamed (f8_s2conv.hip); image borders
        # through the border-class bias (unsigned mid) or real zeros (signed mid), odd image counts (workgroup groups of 4 images)
        assert ('conv3x3s2_wreg:' in plan) == ((MID, H) in ((256, 28), (512, 14))), plan
    xd = _t(x, dev)
    for _ in range(3):
        got = net.run(xd).cpu().numpy().reshape(want.shape)
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize('variant', ['body_shifts_left', 'shortcut_shifts_left', 'different_input_formats'])
def test_stage_opening_block_fused_matches_oracle(dev, variant):
    """ResNet-50's stage-0 opening block (64 -> 64 -> 64 -> 256 + 1x1 shortcut, stride 1, 56 wide) as a net of its own:
    planned as ONE fused launch when body.0 and the shortcut read the same int8 form of the block input, as separate
    launches otherwise; either way the int32 block output equals the oracle's IntBlock.forward."""
    from f8net_amd import topology
    from f8net_amd.net import F8Net
    Cin, MID, Cout, H, W, N = 64, 64, 256, 6, 56, 3
    body = [topology.ConvSpec('blk.body.0', Cin, MID, 1, 1, 0, relu=True),
            topology.ConvSpec('blk.body.2', MID, MID, 3, 1, 1, relu=True),
            topology.ConvSpec('blk.body.4', MID, Cout, 1, 1, 0)]
    sc = topology.ConvSpec('blk.shortcut.0', Cin, Cout, 1, 1, 0)
    b = topology.BlockSpec('blk', body, sc, residual=True, post_relu=True)
    fls = {'blk.body.0': (4, 7), 'blk.body.2': (3, 6), 'blk.body.4': (3, 5), 'blk.shortcut.0': (4, 7)}       # 8 vs 11
    if variant == 'shortcut_shifts_left':
        fls.update({'blk.body.4': (6, 7), 'blk.shortcut.0': (4, 6)})                                             # 13 vs 10
    if variant == 'different_input_formats':
        fls['blk.shortcut.0'] = (5, 7)
    x_fl = 9
    params = {}
    for c in body + [sc]:
        in_fl, w_fl = fls[c.key]
        params[c.key + '.weight'] = np.clip(synth.rand_normal_int(25, c.key + 'w' + variant, (c.cout, c.cin, c.k, c.k), 50.0), -127, 127).astype(np.int32)
        params[c.key + '.bias'] = synth.rand_normal_int(26, c.key + 'b', (c.cout,), 2.0 ** 27).astype(np.int32)
        params[c.key + '.weight_fraclen'] = np.array(w_fl, np.int32)
        params[c.key + '.input_fraclen'] = np.array([in_fl], np.int32)
    x = synth.rand_normal_int(27, 'dsf' + variant, (N, Cin, H, W), 2.0e3).astype(np.int32)
    net = F8Net()
    t = net.input(Cin, H, W, x_fl)
    r = t
    for c in body:
        r = net.conv(r, params[c.key + '.weight'], params[c.key + '.bias'], stride=1, pad=c.pad, groups=1,
                     weight_fl=fls[c.key][1], input_fl=fls[c.key][0], input_signed=False, quant_input=True, relu=c.relu)
    s = net.conv(t, params[sc.key + '.weight'], params[sc.key + '.bias'], stride=1, pad=0, groups=1,
                 weight_fl=fls[sc.key][1], input_fl=fls[sc.key][0], input_signed=False, quant_input=True, relu=False)
    r = net.add(r, s, relu=True)
    net.output(r, as_float=False)
    net.finalize(N)
    plan = net.describe()
    assert ('fused_bottleneck_ds' in plan) == (variant != 'different_input_formats'), plan
    got = net.run(_t(x, dev)).cpu().numpy().reshape(N, Cout, H, W)
    want, want_fl = oracle.block_forward(b, params, x, x_fl)
    assert net.output_fraclen == want_fl
    np.testing.assert_array_equal(got, want)
    assert (np.abs(want.astype(np.int64)) > 2**30).any()


OPENER_CASES = [(16, 3, v, t) for v in ('body_shifts_left', 'shortcut_shifts_left_signed_mid', 'different_input_formats') for t in ('i32', 'i8', 'both')] + \
               [(56, 2, 'body_shifts_left', 'both'), (56, 2, 'shortcut_shifts_left_signed_mid', 'both')]      # the full-height map once per fused variant


@pytest.mark.parametrize('H,N,variant,tail', OPENER_CASES, ids=lambda v: str(v))
def test_stage1_opening_block_stride2_fused_matches_oracle(dev, H, N, variant, tail):
    """ResNet-50's stage-1 opening block (256 -> 128 -> 3x3 / 2 -> 512 + strided 1x1 shortcut, 56 wide; fix_resnet.py:26-77) as
    a net of its own: ONE launch (f8_opener.hip) when body.0 and the shortcut read the same int8 form of the block input,
    separate launches otherwise.  `tail` decides which forms the block output needs: the int32 stream, an int8 copy (staged
    through LDS into 128-byte lines), or both.  Top image border (tile 0), several row tiles, wrap / clamp in the join."""
    from f8net_amd import topology
    from f8net_amd.net import F8Net
    Cin, MID, Cout, W = 256, 128, 512, 56
    body = [topology.ConvSpec('blk.body.0', Cin, MID, 1, 1, 0, relu=True),
            topology.ConvSpec('blk.body.2', MID, MID, 3, 2, 1, relu=True),
            topology.ConvSpec('blk.body.4', MID, Cout, 1, 1, 0)]
    sc = topology.ConvSpec('blk.shortcut.0', Cin, Cout, 1, 2, 0)
    b = topology.BlockSpec('blk', body, sc, residual=True, post_relu=True)
    fls = {'blk.body.0': (4, 7), 'blk.body.2': (3, 6), 'blk.body.4': (3, 5), 'blk.shortcut.0': (4, 7)}       # 8 vs 11
    if variant == 'shortcut_shifts_left_signed_mid':
        fls.update({'blk.body.4': (6, 7), 'blk.shortcut.0': (4, 6)})                                             # 13 vs 10
        body[1].signed_in = True          # plain (unbiased) LDS patch and its zero border
    if variant == 'different_input_formats':
        fls['blk.shortcut.0'] = (5, 7)
    x_fl = 9
    params = {}
    for c in body + [sc]:
        in_fl, w_fl = fls[c.key]
        params[c.key + '.weight'] = np.clip(synth.rand_normal_int(45, c.key + 'w' + variant, (c.cout, c.cin, c.k, c.k), 50.0), -127, 127).astype(np.int32)
        params[c.key + '.bias'] = synth.rand_normal_int(46, c.key + 'b', (c.cout,), 2.0 ** 27).astype(np.int32)   # joins wrap / clamp
        params[c.key + '.weight_fraclen'] = np.array(w_fl, np.int32)
        params[c.key + '.input_fraclen'] = np.array([in_fl], np.int32)
    x = synth.rand_normal_int(47, 'op2' + variant, (N, Cin, H, W), 2.0e3).astype(np.int32)
    net = F8Net()
    t = net.input(Cin, H, W, x_fl)
    r = t
    for c in body:
        r = net.conv(r, params[c.key + '.weight'], params[c.key + '.bias'], stride=c.stride, pad=c.pad, groups=1,
                     weight_fl=fls[c.key][1], input_fl=fls[c.key][0], input_signed=c.signed_in, quant_input=True, relu=c.relu)
    s = net.conv(t, params[sc.key + '.weight'], params[sc.key + '.bias'], stride=2, pad=0, groups=1,
                 weight_fl=fls[sc.key][1], input_fl=fls[sc.key][0], input_signed=False, quant_input=True, relu=False)
    r = net.add(r, s, relu=True)
    want, want_fl = oracle.block_forward(b, params, x, x_fl)
    assert (np.abs(want.astype(np.int64)) > 2**30).any()       # the join ran at full int32 width
    P, Q = H // 2, W // 2
    if tail != 'i32':                                          # a consumer that needs the int8 copy of the block output
        wt = np.clip(synth.rand_normal_int(48, 'tailw', (Cout if tail == 'both' else 32, Cout, 1, 1), 40.0), -127, 127).astype(np.int32)
        bt = synth.rand_normal_int(49, 'tailb', (wt.shape[0],), 1.0e4).astype(np.int32)
        in_fl_t, w_fl_t = 2, 6
        c2 = net.conv(r, wt, bt, stride=1, pad=0, groups=1, weight_fl=w_fl_t, input_fl=in_fl_t, input_signed=False, quant_input=True, relu=False)
        y = oracle.conv2d(oracle.requant(want, in_fl_t, want_fl, False), wt, bt, 1, 0)
        if tail == 'both':                                     # ... and the int32 stream as a residual operand
            c2 = net.add(c2, r, relu=False)
            y, _ = oracle.add_align(y, want, in_fl_t + w_fl_t, want_fl)
        r, want = c2, y
    net.output(r, as_float=False)
    net.finalize(N)
    plan = net.describe()
    assert ('fused_opener_s2' in plan) == (variant != 'different_input_formats'), plan
    got = net.run(_t(x, dev)).cpu().numpy().reshape(N, -1, P, Q)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize('case', [(3, 112, 224, True), (2, 224, 224, False), (1, 56, 64, False)], ids=lambda c: 'x'.join(map(str, c)))
def test_fused_head_stem_conv_maxpool(dev, case):
    """ResNet head as ONE launch (f8_stem.hip): 7x7/2 conv + ReLU + [requant] + 3x3/2 max-pool, followed by a 1x1 conv that
    fixes the pool output's int8 format.  Non-square maps, odd batch, signed (normalize) and unsigned inputs, image borders."""
    from f8net_amd.net import F8Net
    N, H, W, signed = case
    lo, hi = (-127, 127) if signed else (0, 255)
    x = synth.rand_uniform_int(31, f'hx{case}', (N, 3, H, W), lo, hi).astype(np.int32)
    w1 = np.clip(synth.rand_normal_int(32, 'hw1', (64, 3, 7, 7), 40.0), -127, 127).astype(np.int32)
    b1 = synth.rand_normal_int(33, 'hb1', (64,), 3.0e4).astype(np.int32)
    w2 = np.clip(synth.rand_normal_int(34, 'hw2', (32, 64, 1, 1), 40.0), -127, 127).astype(np.int32)
    b2 = synth.rand_normal_int(35, 'hb2', (32,), 1.0e3).astype(np.int32)
    in_fl, w_fl, fl2 = (5 if signed else 8), 6, 4
    net = F8Net()
    t = net.input(3, H, W, in_fl)
    c = net.conv(t, w1, b1, stride=2, pad=3, groups=1, weight_fl=w_fl, input_fl=in_fl, input_signed=signed, quant_input=False, relu=True)
    p = net.maxpool(c, 3, 2, 1)
    o = net.conv(p, w2, b2, stride=1, pad=0, groups=1, weight_fl=5, input_fl=fl2, input_signed=False, quant_input=True, relu=False)
    net.output(o, as_float=False)
    net.finalize(N)
    P, Q = H // 4, W // 4
    assert ('stem7x7s2+maxpool3x3s2' in net.describe()) == (P % 7 == 0 and Q % 8 == 0), net.describe()
    got = net.run(_t(x, dev)).cpu().numpy().reshape(N, 32, P, Q)
    y = oracle.relu(oracle.conv2d(x, w1, b1, 2, 3))
    y = oracle.maxpool(y, 3, 2, 1)
    y = oracle.requant(y, fl2, in_fl + w_fl, False)
    want = oracle.conv2d(y, w2, b2, 1, 0)
    np.testing.assert_array_equal(got, want)
