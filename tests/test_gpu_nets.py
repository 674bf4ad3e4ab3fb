"""GPU parity, whole networks: fused HIP plan vs golden fixtures captured from the reference, vs the
CPU oracle on fresh seeds, and vs our own op-level path.  Bit-exact (integer arithmetic)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from f8net_amd import synth, topology
from oracle import oracle

ARCHS = ['resnet18', 'resnet50', 'mobilenet_v1', 'mobilenet_v2']
DEEP = ['resnet34', 'resnet101', 'resnet152']      # depths the reference's Model builds without a yml (fix_resnet.py:418-451): chains cut at kChainMaxBlocks, identity-first chain instances


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return torch.device('cuda:0')


def _golden_setup(arch, golden_dir):
    g = np.load(os.path.join(golden_dir, f'net_{arch}.npz'))
    spec = topology.get(arch, normalize=bool(g['normalize']))
    return g, spec, synth.reference_params(spec, seed=1234)


@pytest.mark.parametrize('arch', ARCHS + DEEP)
def test_fused_net_matches_reference_golden(arch, golden_dir, dev):
    from f8net_amd.net import build_net
    g, spec, params = _golden_setup(arch, golden_dir)
    for hw, n in ((64, 2), (224, 1)):
        x, x_fl = synth.make_input(spec, params, n, hw, seed=7)
        net = build_net(spec, params, max_batch=n, hw=hw)
        got = net.run(torch.from_numpy(x).to(dev)).cpu().numpy()
        np.testing.assert_array_equal(got, g[f's1234_hw{hw}_n{n}/logits'], err_msg=f'{arch} hw{hw}')


@pytest.mark.parametrize('arch', ARCHS + DEEP)
def test_fused_net_matches_oracle_fresh_seed(arch, dev):
    from f8net_amd.net import build_net
    spec = topology.get(arch, normalize=(arch == 'mobilenet_v2'))
    params = synth.make_params(spec, seed=77)
    n, hw = 5, 96
    x, x_fl = synth.make_input(spec, params, n, hw, seed=3)
    net = build_net(spec, params, max_batch=8, hw=hw)          # capacity > batch
    got = net.run(torch.from_numpy(x).to(dev)).cpu().numpy()
    want = oracle.net_forward(spec, params, x, x_fl)
    np.testing.assert_array_equal(got, want)
    # same plan, smaller batch, run twice (arena reuse must not leak state between runs)
    got2 = net.run(torch.from_numpy(x[:2]).to(dev)).cpu().numpy()
    np.testing.assert_array_equal(got2, want[:2])


@pytest.mark.parametrize('hw', [220, 212, 200])
def test_resnet50_input_sizes_whose_stage_maps_are_not_halved_exactly(hw, dev):
    """hw 220: stage 0 is 55x55 and stage 1 28x28; hw 212: stage 1 is 27x27 and stage 2 14x14 — output maps the TAIL chain instances exist for,
    fed by input maps that are NOT twice their size (the planner took them, the kernel read a 55-wide map with stride 56).  The planner
    leaves those joins to the generic dual GEMM; logits == oracle."""
    from f8net_amd.net import build_net
    spec = topology.get('resnet50', normalize=True)
    params = synth.make_params(spec, seed=5, fraclens=topology.R50_NVIDIA_FRACLENS)
    n = 3
    x, x_fl = synth.make_input(spec, params, n, hw, seed=9)
    net = build_net(spec, params, max_batch=n, hw=hw)
    got = net.run(torch.from_numpy(x).to(dev)).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.net_forward(spec, params, x, x_fl), err_msg=f'hw{hw}: {net.describe()}')


@pytest.mark.parametrize('arch', ['resnet18', 'mobilenet_v2'])
def test_op_level_path_equals_fused_path(arch, dev):
    """The reference's control flow over our op-level kernels == the fused plan == the oracle."""
    from f8net_amd import int_model
    spec = topology.get(arch)
    params = synth.make_params(spec, seed=5)
    m = int_model.from_params(spec, params).to(dev)
    x, x_fl = synth.make_input(spec, params, 2, 64, seed=9)
    xt = torch.from_numpy(x).to(dev)
    setattr(xt, 'output_fraclen', x_fl)
    fused = m(xt).cpu().numpy()
    oplevel = m.forward_op_level(xt).cpu().numpy()
    want = oracle.net_forward(spec, params, x, x_fl)
    np.testing.assert_array_equal(fused, want)
    np.testing.assert_array_equal(oplevel, want)


@pytest.mark.parametrize('arch', ARCHS)
def test_integize_mode_matches_reference_float_carried_evaluation(arch, golden_dir, dev):
    """`IntModel.forward_integize` takes the REAL-valued float batch the reference's float-carried branches
    take (fix_resnet.py:384-409) and returns what they return — captured from the reference IntModel in that mode."""
    from f8net_amd import int_model
    g, spec, params = _golden_setup(arch, golden_dir)
    m = int_model.from_params(spec, params).to(dev)
    x, x_fl = synth.make_input(spec, params, 2, 64, seed=7)
    xr = (torch.from_numpy(x).float() / float(2 ** x_fl)).to(dev)
    got = m.forward_integize(xr).cpu().numpy()
    np.testing.assert_array_equal(got, g['s1234_hw64_n2/integize_logits'])


def test_state_dict_keys_match_reference_export():
    """84 / 216 / 112 / 212 keys, 4 per layer."""
    from f8net_amd import int_model
    for arch, nkeys in (('resnet18', 84), ('resnet50', 216), ('mobilenet_v1', 112), ('mobilenet_v2', 212)):
        m = int_model.IntModel(topology.get(arch))
        assert len(m.state_dict()) == nkeys
        assert 'head.0.weight_fraclen' in m.state_dict() and 'classifier.0.input_fraclen' in m.state_dict()


# the single-GPU configurations of BASELINE.json (configs[1..3]) plus the bench workload and MobileNet-V1
FULL_SIZE = [('resnet50', 128), ('resnet50', 256), ('resnet18', 128), ('mobilenet_v2', 128), ('mobilenet_v1', 128)]


@pytest.mark.parametrize('arch,n', FULL_SIZE, ids=lambda v: str(v))
def test_full_size_batch_properties(dev, arch, n):
    """Full-size batches: size-independent properties instead of a CPU forward of the whole batch —
    (1) the batch runs as concurrent sub-batches on internal streams: repeated runs are identical
        (this caught an arena-sharing race between sub-batches during development);
    (2) images are independent: the batch == the same images run in chunks of 32, and permuted;
    (3) images from the first, a middle and the LAST (ragged) chunk of the chunked fused blocks equal the CPU oracle bit for bit."""
    from f8net_amd.net import build_net
    normalize = arch == 'resnet50'
    spec = topology.get(arch, normalize=normalize)
    params = synth.reference_params(spec, seed=1234)
    x, x_fl = synth.make_input(spec, params, n, 224, seed=11)
    net = build_net(spec, params, max_batch=n, hw=224)
    xt = torch.from_numpy(x).to(dev)
    full = net.run(xt).cpu().numpy()
    for _ in range(3):
        np.testing.assert_array_equal(net.run(xt).cpu().numpy(), full)
    parts = np.concatenate([net.run(xt[i:i + 32].contiguous()).cpu().numpy() for i in range(0, n, 32)])
    np.testing.assert_array_equal(full, parts)
    perm = np.array(synth.rand_uniform_int(1, 'perm', (n,), 0, 10**9)).argsort()
    permuted = net.run(xt[torch.from_numpy(perm).to(dev)].contiguous()).cpu().numpy()
    np.testing.assert_array_equal(permuted, full[perm])
    idx = [0, 1, n // 2 - 1, n // 2, n - 3, n - 2, n - 1]      # chunk boundaries of the 56x56 / 28x28 groups fall in between
    np.testing.assert_array_equal(full[idx], oracle.net_forward(spec, params, x[idx], x_fl))
    assert np.count_nonzero(full) > 0.9 * full.size


def test_every_inverted_residual_block_fused_matches_golden(dev, golden_dir):
    """Option fuse_ir = 2 plans EVERY MobileNet-V2 block onto fused_ir_kernel (the default fuses the ones where it wins): all
    eight channel-pair instances, row tiles and whole-image tiles, both strides, with and without residual — logits equal the
    goldens captured from the reference."""
    from f8net_amd.net import build_net
    g, spec, params = _golden_setup('mobilenet_v2', golden_dir)
    for hw, n in ((64, 2), (224, 1)):
        x, _ = synth.make_input(spec, params, n, hw, seed=7)
        net = build_net(spec, params, max_batch=n, hw=hw, options={'fuse_ir': 2})
        assert net.describe().count('fused_ir_') == 16
        np.testing.assert_array_equal(net.run(torch.from_numpy(x).to(dev)).cpu().numpy(), g[f's1234_hw{hw}_n{n}/logits'])
    x, fl = synth.make_input(spec, params, 11, 224, seed=13)          # ragged batch: whole-image tiles of 2 with an odd image count
    net = build_net(spec, params, max_batch=11, hw=224, options={'fuse_ir': 2})
    np.testing.assert_array_equal(net.run(torch.from_numpy(x).to(dev)).cpu().numpy(), oracle.net_forward(spec, params, x, fl))


@pytest.mark.parametrize('arch', ARCHS)
def test_integer_only_requant_plan_matches_golden(arch, golden_dir, dev):
    """Option requant_float = 0 (the default since round 5): every kernel requantises with the integer shift / round-half-even / clamp of
    fix_quant_ops.py:99-112 (no float instruction in any epilogue); requant_float = 1 routes ReLU -> unsigned 8-bit right shifts of bounded
    accumulators through the float converter instead.  Both equal the goldens captured from the reference."""
    from f8net_amd.net import build_net
    g, spec, params = _golden_setup(arch, golden_dir)
    for hw, n in ((64, 2), (224, 1)):
        x, _ = synth.make_input(spec, params, n, hw, seed=7)
        for rq in (0, 1):
            net = build_net(spec, params, max_batch=n, hw=hw, options={'requant_float': rq, 'fuse_ir': 2})
            assert net.get_option('requant_float') == rq
            np.testing.assert_array_equal(net.run(torch.from_numpy(x).to(dev)).cpu().numpy(), g[f's1234_hw{hw}_n{n}/logits'], err_msg=f'{arch} hw{hw} rq{rq}')


def test_mobilenet_v2_fused_launches_where_the_rounding_add_wraps(dev):
    """Biases next to 2^31 in the head conv, the head's depthwise conv and the expand / depthwise convs of two inverted-residual blocks:
    their accumulators cannot be bounded below 2^31 - 2^16, the reference's `v + 2^(n-1)` wraps there (fix_quant_ops.py:100-104: the value
    turns negative, the clamp makes it 0), and the fused launches (stem_rows_kernel<.., true>, fused_ir_kernel) must take their INTEGER
    requantisation by themselves — same plan, oracle's values."""
    from f8net_amd.net import build_net
    spec = topology.get('mobilenet_v2')
    params = synth.make_params(spec, seed=55)
    for key, ch in (('head.0', 5), ('stage_0_layer_0.body.0', 7), ('stage_1_layer_0.body.0', 11), ('stage_1_layer_0.body.2', 3),
                    ('stage_2_layer_1.body.0', 20), ('stage_2_layer_1.body.2', 9)):
        b = params[key + '.bias'].copy()
        b[ch], b[(ch + 13) % b.size] = 2 ** 31 - 50, 2 ** 31 - 2 ** 12
        params[key + '.bias'] = b
    for hw, n in ((64, 3), (224, 2)):
        x, fl = synth.make_input(spec, params, n, hw, seed=3)
        want = oracle.net_forward(spec, params, x, fl)
        for opts in ({'fuse_ir': 2, 'requant_float': 1}, {'fuse_ir': 2, 'requant_float': 0}):
            net = build_net(spec, params, max_batch=n, hw=hw, options=opts)
            plan = net.describe()
            assert 'head3x3s2+dw3x3+1x1' in plan and plan.count('fused_ir_') == 16, plan
            np.testing.assert_array_equal(net.run(torch.from_numpy(x).to(dev)).cpu().numpy(), want, err_msg=f'hw{hw} {opts}')


@pytest.mark.parametrize('mode', [1, 2])
def test_pipelined_runs_overlap_safely(dev, mode):
    """f8_net_set_pipelined (1: lagged sub-batches, 2: whole batches alternating between two streams / arena copies):
    consecutive runs may overlap when the caller double-buffers; every run's result still equals the strictly ordered one
    (different inputs per run, two alternating output buffers, many runs in flight)."""
    from f8net_amd.net import build_net
    spec = topology.get('resnet50', normalize=True)
    params = synth.make_params(spec, seed=1234, fraclens=topology.R50_NVIDIA_FRACLENS)
    n = 64
    net = build_net(spec, params, max_batch=n, hw=224)
    xs = [torch.from_numpy(synth.make_input(spec, params, n, 224, seed=100 + i)[0]).to(dev) for i in range(3)]
    want = [net.run(x).cpu().numpy() for x in xs]
    assert not np.array_equal(want[0], want[1])
    net.set_pipelined(mode)
    if 'F8_SPLIT' not in os.environ:
        assert net.num_parts(n) == (2 if mode == 1 else 1)
    outs = [torch.empty((n, spec.num_classes), dtype=torch.float32, device=dev) for _ in range(2)]
    hist = []
    for rep in range(12):
        o = outs[rep & 1]
        net.run(xs[rep % 3], out=o)
        hist.append(o.clone())        # consumer enqueued right after its run: the buffer is free one call before it is rewritten
    torch.cuda.synchronize(dev)
    net.set_pipelined(False)
    for rep, y in enumerate(hist):
        np.testing.assert_array_equal(y.cpu().numpy(), want[rep % 3])
    # the profiled pass times the launches this mode issues, and is as exact
    net.set_pipelined(mode)
    y, ms = net.run_profiled(xs[1])
    net.set_pipelined(False)
    # every planned step is timed; the input step launches nothing when the stem launch reads the caller's buffer: no duration
    names = [l.split()[1] for l in net.describe().splitlines() if l.strip() and l.split()[0].isdigit()]
    assert len(ms) == net.num_launches == len(names)
    assert all((t == 0) if n.startswith('input(read') else (t > 0) for n, t in zip(names, ms)), list(zip(names, ms))
    np.testing.assert_array_equal(y.cpu().numpy(), want[1])


def test_autotuned_plan_is_bit_identical(dev, golden_dir):
    """f8_net_autotune only re-tiles launches: the logits still equal the reference golden."""
    from f8net_amd.net import build_net
    g = np.load(os.path.join(golden_dir, 'net_resnet50.npz'))
    spec = topology.get('resnet50', normalize=bool(g['normalize']))
    params = synth.make_params(spec, seed=1234, fraclens=topology.R50_NVIDIA_FRACLENS)
    x, _ = synth.make_input(spec, params, 1, 224, seed=7)
    net = build_net(spec, params, max_batch=4, hw=224)
    before = net.describe()
    changed = net.autotune(4, dev)
    assert changed >= 0 and (changed == 0) == (net.describe() == before)
    np.testing.assert_array_equal(net.run(torch.from_numpy(x).to(dev)).cpu().numpy(), g['s1234_hw224_n1/logits'])


def test_late_stage_kernels_equal_the_tile_per_workgroup_plan_at_every_batch_size(dev):
    """The bench plan (128 images per launch, two batches in flight) runs the late stages on the operand-stationary kernels of
    DESIGN 4.9 and the classifier / stem on their fused forms; the same net planned with all of them OFF runs conv_igemm_kernel there.
    Both must give the same logits for ANY image count the 128-image plan is asked to run: one image (fewer pixel tiles than
    workgroups), counts that leave the last workgroup group / pixel tile ragged, the full batch."""
    from f8net_amd.net import build_net
    spec = topology.get('resnet50', normalize=True)
    params = synth.make_params(spec, seed=1234, fraclens=topology.R50_NVIDIA_FRACLENS)
    x, _ = synth.make_input(spec, params, 128, 224, seed=5)
    xd = torch.from_numpy(x).to(dev)
    big = build_net(spec, params, max_batch=128, hw=224, options={'whole_batch_launches': 1})
    big.set_pipelined(2)
    plan = big.describe()
    for name in ('conv1x1_wstat:', 'conv3x3s2_wreg:', 'stage_chain_x3_tail+avgpool:', 'linear_dense:', 'read by the stem launch'):    # stage 3: the cluster chain (f8_cchain.hip)
        assert name in plan, plan
    # ... and the plan of rounds 3 - 5 for the 7x7 stage (option fuse_chain7 = 0): dual GEMM, fused_p12 + residual-carrying 1x1, join + pool
    big7 = build_net(spec, params, max_batch=128, hw=224, options={'whole_batch_launches': 1, 'fuse_chain7': 0})
    big7.set_pipelined(2)
    plan7 = big7.describe()
    for name in ('conv1x1_wstat:', 'conv1x1_wstat_dual:', 'conv1x1_wstat_res:', 'conv3x3s2_wreg:', 'fused_p12:', 'conv1x1_res+avgpool:', 'linear_dense:'):
        assert name in plan7, plan7
    off = {'wstat': 0, 's2wreg': 0, 'fuse_p12': 0, 'wreg': 0, 'fuse_fc': 0, 'fuse_input': 0}
    ref = build_net(spec, params, max_batch=128, hw=224, options=off)
    assert not any(k in ref.describe() for k in ('wstat', 'wreg', 'fused_p12:', 'linear_dense'))
    full = ref.run(xd).cpu().numpy()
    for k in (1, 2, 3, 31, 33, 100, 127, 128):
        got = big.run(xd[:k].contiguous()).cpu().numpy()
        np.testing.assert_array_equal(got, full[:k])
        if k in (1, 33, 128):
            np.testing.assert_array_equal(big7.run(xd[:k].contiguous()).cpu().numpy(), full[:k])
    big.set_pipelined(False)


@pytest.mark.parametrize('rq', [0, 1])
def test_bench_plan_with_three_batches_in_flight_equals_the_oracle(rq, dev):
    """bench.py's EXACT plan and schedule (VERDICT r4 #5c): ResNet-50, 128 images, `whole_batch_launches = 1, arena_copies = 3,
    pipeline_depth = 3`, `set_pipelined(2)`, 12 overlapping runs over 4 rotating input batches and 4 rotating output buffers — 8 images of
    every run (the first 3, the last 3 and two from the middle of the batch: first / last workgroup groups of every launch) against the CPU
    oracle; rq = 0 is the headline's integer-only requantisation, rq = 1 the float-converter plan bench.py times beside it."""
    from f8net_amd.net import build_net
    spec = topology.get('resnet50', normalize=True)
    params = synth.reference_params(spec, seed=1234)
    depth, NX, N = 3, 4, 128
    net = build_net(spec, params, max_batch=N, hw=224, options={'whole_batch_launches': 1, 'arena_copies': depth, 'pipeline_depth': depth, 'requant_float': rq})
    net.upload()
    net.set_pipelined(2)
    pick = [0, 1, 2, 63, 64, 125, 126, 127]
    xs, wants = [], []
    for j in range(NX):
        x, fl = synth.make_input(spec, params, N, 224, seed=100 + j)
        xs.append(torch.from_numpy(x).to(dev))
        wants.append(oracle.net_forward(spec, params, np.ascontiguousarray(x[pick]), fl))
    outs = [torch.empty((N, spec.num_classes), dtype=torch.float32, device=dev) for _ in range(depth + 1)]
    keep = []
    for i in range(12):
        net.run(xs[i % NX], out=outs[i % (depth + 1)])
        keep.append((i % NX, outs[i % (depth + 1)]))
        if (i + 1) % (depth + 1) == 0:                     # the output ring is full: collect it before its buffers are rewritten
            torch.cuda.synchronize()
            for j, o in keep:
                np.testing.assert_array_equal(o[pick].cpu().numpy(), wants[j], err_msg=f'run with input {j}, rq {rq}')
            keep = []
    torch.cuda.synchronize()
    net.check()
    net.set_pipelined(False)
