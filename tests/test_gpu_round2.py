"""Round-2 additions on the GPU: torch.ops.f8net.* (dispatcher over the C ABI), in-place weight edits, the input-ready event of
pipelined callers, device binding, MobileNet-V2 corner formats (unsigned input_fl 8, weight_fl 0 / 1, large shifts), RCCL ranks."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip('torch')

from f8net_amd import _lib, synth, topology
from oracle import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    oracle.build()
    return torch.device('cuda', 0)


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_torch_ops_functional_calls_match_oracle(dev):
    import f8net_amd.torch_ops  # noqa: F401
    x = synth.rand_normal_int(61, 'tx', (3, 64, 12, 12), 3.0e4).astype(np.int32)
    q = torch.ops.f8net.requant(_t(x, dev), 3, 11, False)
    np.testing.assert_array_equal(q.cpu().numpy(), oracle.requant(x, 3, 11, False))
    w = np.clip(synth.rand_normal_int(62, 'tw', (96, 64, 3, 3), 40.0), -127, 127).astype(np.int32)
    b = synth.rand_normal_int(63, 'tb', (96,), 1.0e4).astype(np.int32)
    xq = oracle.requant(x, 3, 11, False)
    wt, bt = torch.from_numpy(w), torch.from_numpy(b)               # parameters may live on the host: they are packed at plan time
    y = torch.ops.f8net.conv2d(_t(xq, dev), wt, bt, 2, 1, 1, 6, 3, False)
    np.testing.assert_array_equal(y.cpu().numpy(), oracle.conv2d(xq, w, b, 2, 1))
    # second call: cached plan; then an IN-PLACE weight edit must re-plan, not run the snapshot
    y2 = torch.ops.f8net.conv2d(_t(xq, dev), wt, bt, 2, 1, 1, 6, 3, False)
    assert torch.equal(y, y2)
    wt[0, 0, 0, 0] += 17
    w2 = wt.numpy().copy()
    y3 = torch.ops.f8net.conv2d(_t(xq, dev), wt, bt, 2, 1, 1, 6, 3, False)
    np.testing.assert_array_equal(y3.cpu().numpy(), oracle.conv2d(xq, w2, b, 2, 1))
    assert not torch.equal(y, y3)
    # relu_ / add_align_ mutate in place and return their first argument
    a = _t(synth.rand_normal_int(64, 'ta', (5, 7, 3), 1.0e9).astype(np.int32), dev)
    c = _t(synth.rand_normal_int(65, 'tc', (5, 7, 3), 1.0e9).astype(np.int32), dev)
    want, _ = oracle.add_align(a.cpu().numpy(), c.cpu().numpy(), 9, 12)
    r = torch.ops.f8net.add_align_(a, c, 9, 12)
    assert r.data_ptr() == a.data_ptr()
    np.testing.assert_array_equal(a.cpu().numpy(), want)
    np.testing.assert_array_equal(torch.ops.f8net.relu_(a).cpu().numpy(), np.maximum(want, 0))


def test_module_weight_edit_replans(dev):
    """F8Conv2d used to snapshot its weights at the first forward; an in-place edit is now seen."""
    from f8net_amd import ops
    m = ops.F8Conv2d(32, 32, 1).to(dev)
    w = np.clip(synth.rand_normal_int(66, 'mw', (32, 32, 1, 1), 40.0), -127, 127).astype(np.int32)
    m.weight.data.copy_(_t(w, dev)); m.input_fraclen.fill_(4); m.weight_fraclen.fill_(6)
    x = synth.rand_uniform_int(67, 'mx', (2, 32, 5, 5), 0, 255).astype(np.int32)
    y1 = m(_t(x, dev)).cpu().numpy()
    np.testing.assert_array_equal(y1, oracle.conv2d(x, w, np.zeros(32, np.int32), 1, 0))
    import f8net_amd.torch_ops as tops
    m.weight.data[3, 5, 0, 0] = -99                     # through `.data`: no version bump; seen after invalidate_plans() ...
    w[3, 5, 0, 0] = -99
    tops.invalidate_plans()
    np.testing.assert_array_equal(m(_t(x, dev)).cpu().numpy(), oracle.conv2d(x, w, np.zeros(32, np.int32), 1, 0))
    tops.set_detect_data_edits(True)                    # ... or by the opt-in content fingerprint
    try:
        m.weight.data[4, 6, 0, 0] = 77
        w[4, 6, 0, 0] = 77
        np.testing.assert_array_equal(m(_t(x, dev)).cpu().numpy(), oracle.conv2d(x, w, np.zeros(32, np.int32), 1, 0))
    finally:
        tops.set_detect_data_edits(False)
    with torch.no_grad():
        m.weight[7, 1, 0, 0] = 55                       # through the parameter: version bump
    w[7, 1, 0, 0] = 55
    np.testing.assert_array_equal(m(_t(x, dev)).cpu().numpy(), oracle.conv2d(x, w, np.zeros(32, np.int32), 1, 0))
    # ... and, with nothing switched on, by the periodic audit of a reused plan (every F8NET_PLAN_AUDIT_PERIOD-th call, default 64)
    m.weight.data[9, 2, 0, 0] = -13
    w[9, 2, 0, 0] = -13
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for _ in range(tops._AUDIT_PERIOD):
            y = m(_t(x, dev))
    np.testing.assert_array_equal(y.cpu().numpy(), oracle.conv2d(x, w, np.zeros(32, np.int32), 1, 0))


def test_net_forward_op_and_device_binding(dev):
    import f8net_amd.torch_ops as tops
    from f8net_amd.net import build_net
    spec = topology.get('resnet18', num_classes=16)
    params = synth.make_params(spec, seed=9)
    x, fl = synth.make_input(spec, params, 3, 64, seed=10)
    net = build_net(spec, params, max_batch=4, hw=64)
    h = tops.register_net(net)
    try:
        got = torch.ops.f8net.net_forward(_t(x, dev), h)
        np.testing.assert_array_equal(got.cpu().numpy(), oracle.net_forward(spec, params, x, fl))
    finally:
        tops.unregister_net(h)
    if torch.cuda.device_count() >= 2:                  # a handle is bound to the device of its upload
        x1 = _t(x, torch.device('cuda', 1))
        with pytest.raises(_lib.F8Error):
            net.run(x1)


@pytest.mark.parametrize('mode', [1, 2])
def test_pipelined_input_from_side_stream_with_ready_event(dev, mode):
    """under f8_net_set_pipelined a run does not wait for work queued on the stream after the previous
    run's entry.  A caller that produces each batch immediately before its run — here on a SIDE stream, like a data loader's
    copy stream — hands the producer's event to the run (f8_net_set_input_ready): every result must equal the ordered one."""
    from f8net_amd.net import build_net
    spec = topology.get('resnet50', normalize=True)
    params = synth.reference_params(spec, seed=1234)
    n = 32
    net = build_net(spec, params, max_batch=n, hw=224)
    host = [torch.from_numpy(synth.make_input(spec, params, n, 224, seed=200 + i)[0]).pin_memory() for i in range(4)]
    want = [net.run(h.to(dev)).cpu().numpy() for h in host]
    net.set_pipelined(mode)
    side = torch.cuda.Stream(dev)
    ins = [torch.empty((n, 3, 224, 224), dtype=torch.int32, device=dev) for _ in range(3)]        # rotating device input buffers
    outs = [torch.empty((n, spec.num_classes), dtype=torch.float32, device=dev) for _ in range(3)]
    done = [None] * 3                                   # consumer-side event per buffer: the previous user of the buffer
    hist = []
    for rep in range(14):
        k = rep % 3
        with torch.cuda.stream(side):
            if done[k] is not None:
                side.wait_event(done[k])                # the run that last read ins[k] has finished
            ins[k].fill_(0)                             # make a missing dependency visible
            ins[k].copy_(host[rep % 4], non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(side)
        net.run(ins[k], out=outs[k], input_ready=ready)
        done[k] = torch.cuda.Event()
        done[k].record(torch.cuda.current_stream(dev))  # behind this run's join on the caller's stream
        hist.append((rep, outs[k].clone()))
    torch.cuda.synchronize(dev)
    net.set_pipelined(False)
    for rep, y in hist:
        np.testing.assert_array_equal(y.cpu().numpy(), want[rep % 4], err_msg=f'run {rep}')


def test_intmodel_refuses_unbuffered_pipelining(dev):
    from f8net_amd import int_model
    spec = topology.get('resnet18', num_classes=8)
    params = synth.make_params(spec, seed=3)
    m = int_model.from_params(spec, params).to(dev)
    x, fl = synth.make_input(spec, params, 2, 64, seed=4)
    xt = _t(x, dev); setattr(xt, 'output_fraclen', fl)
    want = m(xt).cpu().numpy()
    m.set_pipelined(2)
    with pytest.raises(ValueError):
        m(xt)                                            # a fresh output allocation could recycle an in-flight block
    out = torch.empty((2, 8), dtype=torch.float32, device=dev)
    np.testing.assert_array_equal(m(xt, out=out).cpu().numpy(), want)
    m.set_pipelined(0)


MBV2_CORNERS = [
    # (kind, cin, cout, stride, in_fl, w_fl, next_in_fl, next_signed): the formats of fraclen_visual/mbv2_fix_quant.out the seeded draws never hit
    ('dw', 96, 96, 2, 8, 0, 7, False),       # depthwise 8/0 -> project at unsigned fraclen 7 (shift 1)
    ('dw', 576, 576, 1, 8, 1, 8, False),     # depthwise 8/1 -> unsigned fraclen 8 (shift 1)
    ('dw', 960, 960, 1, 8, 6, 8, False),     # depthwise 8/6 (shift 6)
    ('pw', 144, 32, 1, 8, 7, 1, True),       # project 8/7 -> signed fraclen 1: shift 14
    ('pw', 960, 320, 1, 8, 7, 0, True),      # shift 15
    ('pw', 32, 192, 1, 7, 5, 8, False),      # expand 7/5 signed in -> depthwise at unsigned 8: shift 4
]


@pytest.mark.gpu
@pytest.mark.parametrize('normalize', [True, False])
def test_stem_reads_the_raw_input_itself(dev, normalize):
    """fuse_input (default): the fused stem launch builds its patch from the caller's NCHW buffer — int32 (f8_net_run), fp32 quantised
    on the fly (f8_net_run_f32), uint8 through the table (f8_net_run_u8) — and the input launch does nothing; uint8 NHWC keeps the input
    launch.  Every entry equals the plan with the option off, bit for bit: image borders (first row of the first image included: a slot
    there starts before the buffer), ragged batch, both input formats (signed = normalize, unsigned)."""
    import torch
    from f8net_amd.net import build_net
    spec = topology.get('resnet18', normalize=normalize)
    params = synth.reference_params(spec, seed=77)
    n, hw = 3, 224                                      # the fused stem needs whole 7 x 8 pooled tiles
    x, x_fl = synth.make_input(spec, params, n, hw, seed=3)
    nets = {v: build_net(spec, params, max_batch=4, hw=hw, options={'fuse_input': v}) for v in (0, 1)}
    assert 'read by the stem launch' in nets[1].describe() and 'read by the stem launch' not in nets[0].describe()
    xt = torch.from_numpy(x).to(dev)
    ref = nets[0].run(xt).cpu().numpy()
    np.testing.assert_array_equal(nets[1].run(xt).cpu().numpy(), ref)
    np.testing.assert_array_equal(ref, oracle.net_forward(spec, params, x, x_fl))
    img = torch.from_numpy(synth.rand_uniform_int(9, 'u8img', (n, 3, hw, hw), 0, 255).astype(np.uint8)).to(dev)
    f32 = img.to(torch.float32) / 255.0
    if normalize:
        mean = torch.tensor([0.485, 0.456, 0.406], device=dev).view(1, 3, 1, 1); std = torch.tensor([0.229, 0.224, 0.225], device=dev).view(1, 3, 1, 1)
        f32 = ((f32 - mean) / std).contiguous()
    np.testing.assert_array_equal(nets[1].run_f32(f32, normalize).cpu().numpy(), nets[0].run_f32(f32, normalize).cpu().numpy())
    kw = dict(normalize=normalize, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)) if normalize else dict(normalize=False)
    u0 = nets[0].run_u8(img, **kw).cpu().numpy()
    np.testing.assert_array_equal(nets[1].run_u8(img, **kw).cpu().numpy(), u0)
    np.testing.assert_array_equal(nets[1].run_u8(img.permute(0, 2, 3, 1).contiguous(), nhwc=True, **kw).cpu().numpy(), u0)


@pytest.mark.parametrize('case', MBV2_CORNERS, ids=lambda c: f'{c[0]}{c[1]}x{c[2]}s{c[3]}_{c[4]}_{c[5]}to{c[6]}')
def test_mobilenet_v2_corner_formats(dev, case):
    """Unsigned `input_fl = 8` on non-head layers, weight_fl 0 / 1 depthwise, requant shifts >= 12: one conv
    in the corner format followed by a 1x1 that fixes the int8 format of its output, vs the oracle."""
    from f8net_amd.net import F8Net
    kind, cin, cout, stride, in_fl, w_fl, nfl, nsgn = case
    signed_in = (kind == 'pw' and in_fl == 7)
    N, H = 3, 14
    lo, hi = (-127, 127) if signed_in else (0, 255)
    x = synth.rand_uniform_int(71, f'cx{case}', (N, cin, H, H), lo, hi).astype(np.int32)
    k, groups, pad = (3, cin, 1) if kind == 'dw' else (1, 1, 0)
    sig = 40.0 if kind == 'dw' else 2.0 ** (in_fl + w_fl - nfl) * 48 / (60.0 * cin ** 0.5)
    w = np.clip(synth.rand_normal_int(72, f'cw{case}', (cout, cin // groups, k, k), min(45.0, max(1.5, sig))), -127, 127).astype(np.int32)
    b = synth.rand_normal_int(73, f'cb{case}', (cout,), 2.0 ** (in_fl + w_fl - 2)).astype(np.int32)
    w2 = np.clip(synth.rand_normal_int(74, 'cw2', (32, cout, 1, 1), 30.0), -127, 127).astype(np.int32)
    net = F8Net()
    t = net.input(cin, H, H, in_fl)
    c = net.conv(t, w, b, stride=stride, pad=pad, groups=groups, weight_fl=w_fl, input_fl=in_fl, input_signed=signed_in, quant_input=False, relu=(kind == 'dw'))
    o = net.conv(c, w2, None, stride=1, pad=0, groups=1, weight_fl=6, input_fl=nfl, input_signed=nsgn, quant_input=True, relu=False)
    net.output(o, as_float=False)
    net.finalize(N)
    y = oracle.conv2d(x, w, b, stride, pad, groups)
    if kind == 'dw':
        y = oracle.relu(y)
    q = oracle.requant(y, nfl, in_fl + w_fl, nsgn)
    assert np.unique(q).size > 4 and (np.abs(q) >= (127 if nsgn else 255)).mean() < 0.9     # a live signal, not everything saturated
    want = oracle.conv2d(q, w2, np.zeros(32, np.int32), 1, 0)
    P = (H + 2 * pad - k) // stride + 1
    got = net.run(_t(x, dev)).cpu().numpy().reshape(N, 32, P, P)
    np.testing.assert_array_equal(got, want)


def _nccl_worker(rank, world, port, q, force=False):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch as th
    from f8net_amd import dist as f8dist
    from f8net_amd import synth as sy, topology as tp
    from f8net_amd.net import build_net
    r, w, lr = f8dist.init_from_env(backend='nccl')
    if force and w == 1:                                 # a group of ONE rank: init_from_env needs none, the forced collective does
        th.cuda.set_device(lr)
        th.distributed.init_process_group(backend='nccl', rank=0, world_size=1)
    assert th.distributed.get_backend() == 'nccl'        # RCCL, not a silent gloo fallback
    dev = th.device('cuda', lr)
    th.cuda.set_device(dev)
    spec = tp.get('resnet50', normalize=True)
    params = sy.reference_params(spec, seed=1234)
    n_local = 8
    net = build_net(spec, params, max_batch=n_local, hw=224)
    net.set_pipelined(2)
    pf = f8dist.PipelinedShardedForward(lambda t, out: net.run(t, out=out), spec.num_classes, n_local, dev, lagged=True, force_collective=force)
    xs = [sy.make_input(spec, params, n_local * w, 224, seed=300 + i)[0] for i in range(3)]
    outs = []
    for rep in range(6):
        full = pf(th.from_numpy(xs[rep % 3][r * n_local:(r + 1) * n_local]).to(dev))
        outs.append((rep, full))
        if rep >= 1:                                    # the previous batch's gathered logits are complete one call later
            pass
    pf.finish()
    th.cuda.synchronize(dev)
    res = {rep: full.cpu().numpy() for rep, full in outs[-2:]}       # the buffers still holding their last results
    res['device'] = th.cuda.current_device()
    q.put((rank, res))
    th.distributed.barrier()
    th.distributed.destroy_process_group()


def test_two_ranks_over_rccl_match_oracle(dev):
    """§8e on real hardware when >= 2 GPUs are visible: two processes, one per GPU, `PipelinedShardedForward` over nccl
    (= RCCL) with the pipelined schedule bench.py uses; the gathered logits equal the oracle's for the whole batch."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs >= 2 GPUs (the round-end driver boxes have 1; an 8-GPU node runs this)')
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + os.getpid() % 200
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0]['device'] != got[1]['device']          # one GPU per rank
    spec = topology.get('resnet50', normalize=True)
    params = synth.reference_params(spec, seed=1234)
    for rep in (4, 5):
        x, fl = synth.make_input(spec, params, 16, 224, seed=300 + rep % 3)
        want = oracle.net_forward(spec, params, x, fl)
        for rank in (0, 1):
            np.testing.assert_array_equal(got[rank][rep], want)


def test_one_rank_over_rccl_issues_the_collective(dev):
    """What a 1-GPU box can say about §8e: a process group of ONE rank over nccl (= RCCL) with the collective forced — the asynchronous
    all-gather on RCCL's stream, ordered against the net's own streams under the pipelined schedule bench.py uses, waited for before a buffer
    pair is reused; the gathered logits equal the oracle's.  (Rendezvous of several ranks: tests/test_dist_gloo.py and, with >= 2 GPUs, the
    test above.)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29400 + os.getpid() % 200
    p = ctx.Process(target=_nccl_worker, args=(0, 1, port, q, True))
    p.start()
    rank, got = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    spec = topology.get('resnet50', normalize=True)
    params = synth.reference_params(spec, seed=1234)
    for rep in (4, 5):
        x, fl = synth.make_input(spec, params, 8, 224, seed=300 + rep % 3)
        np.testing.assert_array_equal(got[rep], oracle.net_forward(spec, params, x, fl))


def test_output_conv_with_relu_is_rectified(dev):
    """A graph may end in a 1x1 conv + ReLU on a 1x1 map with classifier-sized K: the dense classifier kernel applies no ReLU, so the
    planner must keep such a node on the conv + output path.  The same node without ReLU is the dense launch."""
    from f8net_amd.net import F8Net
    K, CO, N = 512, 40, 5
    w = np.clip(synth.rand_normal_int(91, 'ow', (CO, K, 1, 1), 30.0), -127, 127).astype(np.int32)
    b = synth.rand_normal_int(92, 'ob', (CO,), 2.0 ** 12).astype(np.int32)
    x = synth.rand_normal_int(93, 'ox', (N, K, 1, 1), 3.0e3).astype(np.int32)
    for relu in (True, False):
        net = F8Net()
        t = net.input(K, 1, 1, 9)
        r = net.conv(t, w, b, stride=1, pad=0, groups=1, weight_fl=7, input_fl=4, input_signed=False, quant_input=True, relu=relu)
        net.output(r, as_float=True)
        net.finalize(N)
        assert ('linear_dense' in net.describe()) == (not relu), net.describe()
        got = net.run(_t(x, dev)).cpu().numpy().reshape(N, CO)
        xq = oracle.requant(x, 4, 9, False)
        want = oracle.conv2d(xq, w, b, 1, 0).reshape(N, CO)
        if relu:
            want = np.maximum(want, 0)
        assert (want < 0).any() or relu
        np.testing.assert_array_equal(got, want.astype(np.float32))


def test_out_of_format_int32_input_is_reported(dev):
    """f8_net_run narrows an int32 head input to its 8-bit format; a value outside it (legal for the reference, which feeds int32 to the
    head conv) cannot be honoured and must not pass silently: the input / stem kernel flags it and f8_net_check raises."""
    from f8net_amd.net import build_net
    for arch, hw in (('resnet18', 64), ('mobilenet_v2', 32)):       # fused stem (raw int32 planes) / generic input kernel
        spec = topology.get(arch, num_classes=16)
        params = synth.make_params(spec, seed=5)
        x, _ = synth.make_input(spec, params, 2, hw, seed=9)
        net = build_net(spec, params, max_batch=2, hw=hw)
        net.run(_t(x, dev)); net.check()                              # in-format input: nothing to report
        bad = x.copy(); bad[1, 2, 7, 5] = 300
        net.run(_t(bad, dev))
        with pytest.raises(_lib.F8Error, match='8-bit format'):
            net.check()
        net.check()                                                   # the word is cleared by the report
        quiet = build_net(spec, params, max_batch=2, hw=hw, options={'check_input_range': 0})
        quiet.run(_t(bad, dev)); quiet.check()


def test_handles_share_the_internal_streams_and_stay_independent(dev):
    """The internal streams are one set per device for every handle of the process (a later handle's own streams shared hardware
    queues with the first one's and its batches in flight serialised: DESIGN.md round-3 page).  Two nets, three batches in flight
    each, their runs interleaved call by call on one caller stream: every output equals its oracle."""
    from f8net_amd.net import build_net
    cases = []
    for arch, seed in (('resnet18', 3), ('mobilenet_v2', 4)):
        spec = topology.get(arch, num_classes=24)
        params = synth.make_params(spec, seed=seed)
        x, x_fl = synth.make_input(spec, params, 4, 64, seed=seed + 10)
        net = build_net(spec, params, max_batch=4, hw=64, options={'whole_batch_launches': 1, 'arena_copies': 3, 'pipeline_depth': 3})
        net.upload(); net.set_pipelined(2)
        want = oracle.net_forward(spec, params, x, x_fl)
        outs = [torch.empty((4, 24), dtype=torch.float32, device=dev) for _ in range(4)]
        cases.append((net, _t(x, dev), want, outs))
    for rep in range(9):
        for net, xd, want, outs in cases:
            net.run(xd, out=outs[rep % 4])
    torch.cuda.synchronize()
    for net, xd, want, outs in cases:
        net.check()
        for o in outs:
            assert np.array_equal(o.cpu().numpy(), want)


@pytest.mark.parametrize('hw', [56, 70, 100, 120])
def test_resnet_head_at_odd_sizes(dev, hw):
    """The row-walking head takes input sides that are multiples of 4 (ragged last band, one strip, lanes past the image); other sizes
    keep the tile kernel or the unfused pair: 56 -> pooled 14x14 (two bands), 100 -> 25x25 (3 bands + 4 rows), 120 -> 30x30 (two strips,
    the second 2 columns wide), 70 -> conv 35x35, pool 18x18 (no fused head).  ResNet-18, bs 3, against the oracle; uint8 entry too."""
    from f8net_amd.net import build_net
    spec = topology.get('resnet18', num_classes=16)
    params = synth.make_params(spec, seed=21)
    x, x_fl = synth.make_input(spec, params, 3, hw, seed=22)
    net = build_net(spec, params, max_batch=3, hw=hw)
    plan = net.describe()
    assert ('stem7x7s2+maxpool3x3s2' in plan) == (hw % 4 == 0), plan
    want = oracle.net_forward(spec, params, x, x_fl)
    got = net.run(_t(x, dev)).cpu().numpy()
    net.check()
    assert np.array_equal(got, want)
    got1 = net.run(_t(x[:1], dev)).cpu().numpy()
    assert np.array_equal(got1, want[:1])


def test_graph_switched_on_after_the_first_runs(dev):
    """`graph` is a scheduling key: switched on after a handle already took the device's shared internal streams, the handle moves to
    streams of its own (a capture must not see another handle's launches) and keeps its results."""
    from f8net_amd.net import build_net
    spec = topology.get('resnet18', num_classes=16)
    params = synth.make_params(spec, seed=31)
    x, x_fl = synth.make_input(spec, params, 4, 64, seed=32)
    want = oracle.net_forward(spec, params, x, x_fl)
    net = build_net(spec, params, max_batch=4, hw=64)
    other = build_net(spec, params, max_batch=4, hw=64)
    xd = _t(x, dev)
    assert np.array_equal(net.run(xd).cpu().numpy(), want)
    assert np.array_equal(other.run(xd).cpu().numpy(), want)
    net.set_option('graph', 1)
    out = torch.empty((4, 16), dtype=torch.float32, device=dev)
    for _ in range(4):                                   # eager, capture, replay, replay
        net.run(xd, out=out)
        assert np.array_equal(other.run(xd).cpu().numpy(), want)
        assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize('big', [False, True], ids=['bounded', 'bias_near_2^31'])
def test_depthwise_requant_where_the_rounding_add_wraps(dev, big):
    """The float requantisation (requant_u8x4) is only selected for convs whose accumulators are provably below 2^31 - 2^16
    (f8_net.cpp conv_acc_bounded).  With a bias next to 2^31 the reference's `v + 2^(n-1)` wraps in int32 (fix_quant_ops.py:100-104:
    the value turns negative, the clamp makes it 0); the library must keep the integer form there — and give the oracle's values."""
    from f8net_amd.net import F8Net
    N, C, H = 2, 32, 28
    x = synth.rand_uniform_int(81, 'wrapx', (N, C, H, H), 0, 255).astype(np.int32)
    w = np.clip(synth.rand_normal_int(82, 'wrapw', (C, 1, 3, 3), 40.0), -127, 127).astype(np.int32)
    b = synth.rand_normal_int(83, 'wrapb', (C,), 2.0 ** 9).astype(np.int32)
    if big:
        b[3], b[17] = 2 ** 31 - 50, 2 ** 31 - 2 ** 14        # accumulators within 2^(n-1) of 2^31: the rounding add wraps
    w2 = np.clip(synth.rand_normal_int(84, 'wrapw2', (32, C, 1, 1), 30.0), -127, 127).astype(np.int32)
    in_fl, w_fl, nfl = 8, 5, 4                                # shift 9
    net = F8Net()
    t = net.input(C, H, H, in_fl)
    c = net.conv(t, w, b, stride=1, pad=1, groups=C, weight_fl=w_fl, input_fl=in_fl, input_signed=False, quant_input=False, relu=True)
    o = net.conv(c, w2, None, stride=1, pad=0, groups=1, weight_fl=6, input_fl=nfl, input_signed=False, quant_input=True, relu=False)
    net.output(o, as_float=False)
    net.finalize(N)
    y = oracle.relu(oracle.conv2d(x, w, b, 1, 1, C))
    q = oracle.requant(y, nfl, in_fl + w_fl, False)
    if big:
        assert (y[:, 3] > 2 ** 31 - 2 ** 8).any() and (q[:, 3] == 0).any()       # the wrap really happens in the oracle
    want = oracle.conv2d(q, w2, np.zeros(32, np.int32), 1, 0)
    got = net.run(_t(x, dev)).cpu().numpy().reshape(N, 32, H, H)
    np.testing.assert_array_equal(got, want)


def test_chain_halo_timeout_poisons_one_run_is_reported_and_the_handle_recovers():
    """ADVICE r4 (medium) / VERDICT r4 #5d.  `chain_timeout_ms = 0` makes a stage-chain launch give up at its first unsuccessful poll of a neighbour's
    flag (a stand-in for "another process held the CUs"): the launch runs on without waiting, so the run's logits are garbage and must be POISONED
    (NaN); the next f8_net_run must REFUSE (host-visible mirror of the error word, no synchronisation needed); f8_net_check reports and re-arms; with
    the time-out restored the very same handle equals the oracle again — the error word of the failed run neither cuts later waits short nor poisons
    later logits (it carries the failed run's tag)."""
    import numpy as np
    import torch
    from f8net_amd import synth, topology
    from f8net_amd._lib import F8Error
    from f8net_amd.net import build_net
    from oracle import oracle
    spec = topology.get('resnet50', normalize=True)
    params = synth.make_params(spec, seed=21, fraclens=topology.R50_NVIDIA_FRACLENS)
    n = 8
    x, fl = synth.make_input(spec, params, n, 224, seed=2)
    want = oracle.net_forward(spec, params, x, fl)
    # one sub-batch per run: a chain error word (and the poison the classifier derives from it) is per arena copy, i.e. per sub-batch — with the
    # default split of 2 only the half whose launch gave up would be poisoned
    net = build_net(spec, params, max_batch=n, hw=224, options={'split': 1})
    assert 'stage_chain' in net.describe()
    assert net.get_option('err_mirror') in (0, 1)
    xd = torch.from_numpy(x).cuda()
    assert np.array_equal(net.run(xd).cpu().numpy(), want)
    net.set_option('chain_timeout_ms', 0)
    failed = None
    for _ in range(20):                       # 14 tiles per image at 56x56: some neighbour is late in (practically) every launch
        try:
            y = net.run(xd)
        except F8Error as e:                  # the mirror already shows an earlier iteration's time-out
            failed = str(e)
            break
        torch.cuda.synchronize()
        if torch.isnan(y).any():
            assert torch.isnan(y).all(), 'a run whose chain launch gave up must poison ALL its logits'
            with pytest.raises(F8Error, match='gave up waiting'):
                net.run(xd)                   # refused without a synchronisation: the host-visible mirror is set
            failed = 'nan'
            break
    if failed is None:
        pytest.skip('no halo wait was ever unsuccessful on this box (nothing to test)')
    with pytest.raises(F8Error, match='gave up waiting'):
        net.check()
    net.set_option('chain_timeout_ms', 10000)
    for _ in range(3):
        got = net.run(xd).cpu().numpy()
        assert np.array_equal(got, want), 'the handle must recover after f8_net_check'
    net.check()
