"""Per-handle options (f8_net_set_option) and the torch.ops.f8net.* registration — the parts that need no GPU."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')

from f8net_amd import _lib, synth, topology
from f8net_amd.net import F8Net, build_net


def _one_conv_net():
    net = F8Net()
    t = net.input(32, 8, 8, 4)
    w = np.ones((32, 32, 1, 1), np.int32)
    t = net.conv(t, w, None, stride=1, pad=0, groups=1, weight_fl=5, input_fl=4, input_signed=False, quant_input=False)
    net.output(t, as_float=False)
    return net


def test_options_are_per_handle_and_validated():
    a, b = _one_conv_net(), _one_conv_net()
    assert a.get_option('split') == 2 and a.get_option('chunk56') == -1 and a.get_option('fuse_opener') == 1
    a.set_option('split', 1)
    assert a.get_option('split') == 1 and b.get_option('split') == 2            # two nets in one process may differ
    with pytest.raises(_lib.F8Error):
        a.set_option('no_such_key', 1)
    with pytest.raises(_lib.F8Error):
        a.set_option('split', 9)                                                # outside [1,4]
    a.finalize(2)
    with pytest.raises(_lib.F8Error):
        a.set_option('fuse_blocks', 0)                                          # decides the plan: before finalize only
    a.set_option('chunk56', 16)                                                 # scheduling keys may change between runs
    assert a.get_option('chunk56') == 16


def test_planning_options_change_the_plan(monkeypatch):
    spec = topology.get('resnet50', normalize=True)
    params = synth.reference_params(spec, seed=1234)

    def plan(**opts):
        from f8net_amd.net import record_net
        net = record_net(spec, params, hw=224)
        for k, v in opts.items():
            net.set_option(k, v)
        return net.finalize(128).describe()

    base = plan()
    assert 'fused_opener_s2' in base and 'stage_chain_x3_ds' in base and 'fused_bottleneck' not in base
    assert 'fused_bottleneck_ds' in plan(fuse_chain=0) and 'stage_chain' not in plan(fuse_chain=0)
    assert 'fused_opener_s2' not in plan(fuse_opener=0)
    assert 'fused_bottleneck' not in plan(fuse_blocks=0) and 'stage_chain' not in plan(fuse_blocks=0)
    assert '_dual:' not in plan(fuse_blocks=0, fuse_dual=0)
    # the environment only seeds the defaults of NEW handles
    monkeypatch.setenv('F8_FUSE_OPENER', '0')
    assert 'fused_opener_s2' not in plan()
    monkeypatch.delenv('F8_FUSE_OPENER')
    assert 'fused_opener_s2' in plan()


def test_torch_ops_are_registered_with_meta_shapes():
    import f8net_amd.torch_ops  # noqa: F401
    for name in ('requant', 'relu_', 'add_align_', 'conv2d', 'linear', 'avgpool_sum', 'maxpool', 'net_forward', 'net_forward_f32'):
        assert hasattr(torch.ops.f8net, name), name
    x = torch.empty((2, 16, 9, 9), dtype=torch.int32, device='meta')
    w = torch.empty((8, 16, 3, 3), dtype=torch.int32, device='meta')
    y = torch.ops.f8net.conv2d(x, w, None, 2, 1, 1, 6, 4, False)
    assert tuple(y.shape) == (2, 8, 5, 5) and y.dtype == torch.int32
    assert tuple(torch.ops.f8net.requant(x, 4, 9, True).shape) == (2, 16, 9, 9)
    assert tuple(torch.ops.f8net.maxpool(x, 3, 2, 1).shape) == (2, 16, 5, 5)
    assert tuple(torch.ops.f8net.avgpool_sum(x).shape) == (2, 16)
    # no CPU implementation: the dispatcher refuses CPU tensors
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.f8net.requant(torch.zeros((4,), dtype=torch.int32), 4, 9, True)
