"""GPU parity of the conv + average-pool launch (f8_pool.hip): the network's last 1x1 conv — with its residual join, where it hosts one — and
FXQAvgPool2d behind it in one launch.  Small nets of their own through the C ABI, compared bit for bit with the oracle's conv / align-add /
avgpool_sum (fix_quant_ops.py:126-134, fix_resnet.py:56-76) over ragged batches, maps of 1 .. 64 pixels, padded channel counts, and the shapes the
kernel does not take (they must keep the two launches and the same integers)."""
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


def _params(convs, fls, seed):
    p = {}
    for c in convs:
        in_fl, w_fl = fls[c.key]
        p[c.key + '.weight'] = np.clip(synth.rand_normal_int(seed, c.key + 'w', (c.cout, c.cin, c.k, c.k), 30.0), -127, 127).astype(np.int32)
        p[c.key + '.bias'] = synth.rand_normal_int(seed + 1, c.key + 'b', (c.cout,), 2.0 ** (in_fl + w_fl)).astype(np.int32)
        p[c.key + '.weight_fraclen'] = np.array(w_fl, np.int32)
        p[c.key + '.input_fraclen'] = np.array([in_fl], np.int32)
    return p


def _conv(net, t, c, params, fls):
    return net.conv(t, params[c.key + '.weight'], params[c.key + '.bias'], stride=1, pad=0, groups=1, weight_fl=fls[c.key][1], input_fl=fls[c.key][0],
                    input_signed=c.signed_in, quant_input=True, relu=c.relu)


# K (channels the pooled conv reads), COUT, map side, batch, residual join, fused launch expected
CASES = [
    (512, 2048, 7, 3, True, True),        # ResNet-50's last join, ragged batch
    (512, 2048, 7, 130, True, True),      # more images than walkers x 1: every workgroup walks, some one image more
    (512, 512, 7, 1, True, True),
    (512, 256, 8, 5, True, True),         # 64 pixels: both 32-lane halves full
    (320, 1280, 7, 7, False, True),       # MobileNet-V2's tail conv
    (300, 1280, 5, 4, False, True),       # channels padded to 320, 25 pixels
    (320, 256, 1, 9, False, True),        # a 1 x 1 map
    (320, 200, 7, 3, False, False),       # output channels not a multiple of 256: the two launches stay
    (256, 512, 7, 3, True, False),        # K the kernel has no instance for
]


@pytest.mark.parametrize('case', CASES, ids=lambda c: 'x'.join(map(str, c)))
@pytest.mark.parametrize('fuse', [1, 0])
def test_last_conv_and_average_pool_match_oracle(dev, case, fuse):
    K, COUT, HW, N, res, fused = case
    c0 = topology.ConvSpec('b.0', COUT, K, 1, 1, 0, relu=True)          # residual nets: block input (COUT channels) -> K channels
    c1 = topology.ConvSpec('b.2', K, COUT, 1, 1, 0, relu=not res)       # the pooled conv (MobileNet tail: ReLU behind it; ResNet: the join's ReLU)
    fc = topology.ConvSpec('fc', COUT, 10, 1, 1, 0)
    fls = {'b.0': (4, 7), 'b.2': (3, 6), 'fc': (2, 7)}
    params = _params([c0, c1, fc], fls, 21)
    cin = COUT if res else K
    x_fl = 10
    x = np.abs(synth.rand_normal_int(5, f'poolx{K}{COUT}', (N, cin, HW, HW), 400.0)).astype(np.int32)
    if res:
        x.reshape(-1)[:2] = [2 ** 31 - 1, 2 ** 30]                      # the join wraps / clamps like the reference's int32 add

    net = F8Net()
    net.set_option('fuse_pool', fuse)
    t = net.input(cin, HW, HW, x_fl)
    if res:
        y = _conv(net, _conv(net, t, c0, params, fls), c1, params, fls)
        y = net.add(y, t, relu=True)
    else:
        y = _conv(net, t, c1, params, fls)
    pooled = net.avgpool_sum(y)
    logits = net.linear(pooled, params['fc.weight'].reshape(10, COUT), params['fc.bias'], weight_fl=fls['fc'][1], input_fl=fls['fc'][0], input_signed=False)
    net.output(logits, as_float=False)
    net.finalize(N)
    plan = net.describe()
    assert ('+avgpool' in plan) == bool(fuse and fused), plan
    got = net.run(torch.from_numpy(x).to(dev)).cpu().numpy().reshape(N, 10)
    net.check()

    if res:
        w, fl = oracle._conv_layer(c0, params, x, x_fl)
        w, fl = oracle._conv_layer(c1, params, w, fl)
        w, fl = oracle.add_align(w, x, fl, x_fl)
        w = oracle.relu(w)
    else:
        w, fl = oracle._conv_layer(c1, params, x, x_fl)
    p = oracle.avgpool_sum(w)
    fl += oracle.AVGPOOL_SHIFT
    q = oracle.requant(p, fls['fc'][0], fl, False)
    want = oracle.linear(q, params['fc.weight'].reshape(10, COUT), params['fc.bias'])
    assert net.output_fraclen == fls['fc'][0] + fls['fc'][1]
    np.testing.assert_array_equal(got, want)
