"""GPU parity of the caller-side slices of forward_loss (fix_train.py:676-718): input quantisation (stand-alone and fused
into the net's input kernel), top-k scoring, and the whole evaluation step — against the golden vectors captured from the
reference and against the CPU oracle."""
import os

import numpy as np
import pytest

from f8net_amd import synth, topology
from oracle import oracle

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    oracle.build()
    return torch.device('cuda', 0)


@pytest.fixture(scope='module')
def ops():
    return np.load(os.path.join(GOLD, 'ops.npz'))


def test_quantize_input_golden_and_ties(dev, ops):
    from f8net_amd import pipeline
    q = pipeline.quantize_input(torch.from_numpy(ops['inq/img']).to(dev), normalize=False)
    assert q.output_fraclen == 8 and q.dtype == torch.int32
    np.testing.assert_array_equal(q.cpu().numpy(), ops['inq/u8'])
    q = pipeline.quantize_input(torch.from_numpy(ops['inq/xn']).to(dev), head_input_fraclen=5, input_symmetric=True, normalize=True)
    assert q.output_fraclen == 5
    np.testing.assert_array_equal(q.cpu().numpy(), ops['inq/s8_fl5'])
    # exact ties (k + 0.5) / 2^fl round to even; values beyond the 8-bit range clamp; both signednesses; every fl
    for fl in range(0, 9):
        for signed in (True, False):
            if signed and fl > 7:
                continue
            k = np.arange(-300, 300, dtype=np.float32)
            x = np.concatenate([(k + 0.5) / 2.0 ** fl, k / 2.0 ** fl, synth.rand_normal_int(3, f'qi{fl}', (257,), 500.0) / np.float32(2.0 ** fl) / 3]).astype(np.float32)
            got = pipeline.quantize_input(torch.from_numpy(x).to(dev), head_input_fraclen=fl, input_symmetric=signed, normalize=True)
            v = np.rint(x * np.float32(2.0 ** fl))
            want = np.clip(v, -127, 127) if signed else np.clip(v, 0, 255)
            np.testing.assert_array_equal(got.cpu().numpy(), want.astype(np.int32))
    # 255 * x path on a dense grid of [0, 1] (single IEEE multiply, then round half to even)
    x = (np.arange(0, 100001, dtype=np.float64) / 100000.0).astype(np.float32)
    got = pipeline.quantize_input(torch.from_numpy(x).to(dev), normalize=False)
    np.testing.assert_array_equal(got.cpu().numpy(), oracle.quantize_input_u8(x)[0])
    with pytest.raises(Exception):
        pipeline.quantize_input(torch.zeros(4, device=dev), head_input_fraclen=8, input_symmetric=True, normalize=True)   # fl > 7 signed


@pytest.mark.parametrize('head', ['stem7x7', 'conv3x3'])
@pytest.mark.parametrize('normalize,fl', [(False, 8), (True, 4), (True, 5), (True, 6)])
@pytest.mark.parametrize('nhwc', [False, True])
def test_uint8_input_entry_matches_reference_pipeline(dev, ops, head, normalize, fl, nhwc):
    """f8_net_run_u8: uint8 pixels in, ToTensor / Normalize / forward_loss's quantisation as a table lookup
    inside the input kernel.  The integers the head conv sees must be the ones torch computes from the same pixels (golden
    `inq8/*`, every pixel value in every channel): the logits equal those of f8_net_run on the golden int32 tensor."""
    from f8net_amd.net import F8Net
    u8 = ops['inq8/u8']                                         # [2, 3, 16, 16]
    ints = ops['inq8/plain'] if not normalize else ops[f'inq8/s8_fl{fl}']
    k, stride, pad, cout = (7, 2, 3, 64) if head == 'stem7x7' else (3, 2, 1, 32)
    w = np.clip(synth.rand_normal_int(81, 'u8w' + head, (cout, 3, k, k), 40.0), -127, 127).astype(np.int32)
    b = synth.rand_normal_int(82, 'u8b', (cout,), 1.0e3).astype(np.int32)
    net = F8Net()
    t = net.input(3, 16, 16, fl)
    t = net.conv(t, w, b, stride=stride, pad=pad, groups=1, weight_fl=6, input_fl=fl, input_signed=normalize, quant_input=False, relu=True)
    net.output(t, as_float=False)
    net.finalize(2)
    want = net.run(torch.from_numpy(ints).to(dev)).cpu().numpy()
    np.testing.assert_array_equal(want.reshape(2, cout, 8, 8), oracle.relu(oracle.conv2d(ints, w, b, stride, pad)))
    img = torch.from_numpy(np.ascontiguousarray(u8.transpose(0, 2, 3, 1)) if nhwc else u8).to(dev)
    got = net.run_u8(img, normalize=normalize, mean=ops['inq8/mean'], std=ops['inq8/std'], nhwc=nhwc).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    if not normalize:                                           # a signed head cannot take raw pixels; an unsigned fraclen-8 head needs no mean / std
        got = net.run_u8(img, nhwc=nhwc).cpu().numpy()
        np.testing.assert_array_equal(got, want)
    else:
        from f8net_amd import _lib
        with pytest.raises(_lib.F8Error):
            net.run_u8(img, normalize=False, nhwc=nhwc)


@pytest.mark.parametrize('arch,normalize', [('resnet18', False), ('resnet50', True), ('mobilenet_v2', False)])
def test_fused_input_quantisation_equals_two_step(dev, arch, normalize):
    """model.forward_f32(images) == model(quantize_input(images)) == oracle on the quantised images, bit for bit."""
    from f8net_amd import int_model, pipeline
    spec = topology.get(arch, normalize=normalize)
    fr = topology.R50_NVIDIA_FRACLENS if arch == 'resnet50' else None
    params = synth.make_params(spec, seed=77, fraclens=fr)
    model = int_model.from_params(spec, params).to(dev)
    N, hw = 5, 64
    u = synth.rand_uniform_int(9, f'img{arch}', (N, 3, hw, hw), 0, 255).astype(np.float32) / np.float32(255.0)
    if normalize:
        mean = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(1, 3, 1, 1)
        std = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(1, 3, 1, 1)
        u = ((u - mean) / std).astype(np.float32)
    images = torch.from_numpy(u).to(dev)
    head = model.head[0]
    xq = pipeline.quantize_input(images, head_input_fraclen=int(head.input_fraclen.item()), input_symmetric=bool(head.input_symmetric),
                                 normalize=normalize)
    two_step = model(xq)
    fused = model.forward_f32(images, normalize=normalize)
    assert torch.equal(two_step, fused)
    if normalize:
        x_ref, fl = oracle.quantize_input_normalized(u, int(head.input_fraclen.item()))
    else:
        x_ref, fl = oracle.quantize_input_u8(u)
    np.testing.assert_array_equal(xq.cpu().numpy(), x_ref)
    want = oracle.net_forward(spec, params, x_ref, fl)
    np.testing.assert_array_equal(fused.cpu().numpy(), want)


def test_fused_input_rejects_wrong_mode(dev):
    from f8net_amd import int_model
    spec = topology.get('resnet50', normalize=True)
    params = synth.make_params(spec, seed=1, fraclens=topology.R50_NVIDIA_FRACLENS)
    model = int_model.from_params(spec, params).to(dev)
    with pytest.raises(Exception):          # signed head at fraclen 5 cannot take the 255 * x path
        model.forward_f32(torch.zeros((1, 3, 64, 64), device=dev), normalize=False)


def test_topk_correct(dev, ops):
    from f8net_amd import pipeline
    got = pipeline.topk_correct(torch.from_numpy(ops['topk/logits']).to(dev), torch.from_numpy(ops['topk/target']).to(dev), (1, 5))
    np.testing.assert_array_equal(got.cpu().numpy(), ops['topk/correct'])
    # larger, tie-free: against torch.topk run as the reference does; 1000 classes, k up to 10
    g = torch.Generator().manual_seed(5)
    logits = torch.randperm(1000 * 64, generator=g).reshape(64, 1000).float()
    target = torch.randint(0, 1000, (64,), generator=g)
    target[:8] = logits[:8].argmax(1)
    _, pred = logits.topk(10)
    correct = pred.t().eq(target.view(1, -1).expand(10, -1))
    want = torch.stack([correct[:k].float().sum(0) for k in (1, 5, 10)], 0)
    got = pipeline.topk_correct(logits.to(dev), target.to(dev), (1, 5, 10))
    assert torch.equal(got.cpu(), want)
    # ties: all-equal logits -> the k lowest class indices win
    got = pipeline.topk_correct(torch.zeros((3, 16), device=dev), torch.tensor([0, 4, 5], device=dev), (1, 5))
    np.testing.assert_array_equal(got.cpu().numpy(), [[1, 0, 0], [1, 1, 0]])
    np.testing.assert_array_equal(got.cpu().numpy(), oracle.topk_correct(np.zeros((3, 16), np.float32), [0, 4, 5], (1, 5)))


def test_forward_loss_end_to_end(dev):
    from f8net_amd import int_model, pipeline
    spec = topology.get('resnet18')
    params = synth.make_params(spec, seed=3)
    model = int_model.from_params(spec, params).to(dev)
    N = 6
    u = synth.rand_uniform_int(10, 'fl', (N, 3, 64, 64), 0, 255).astype(np.float32) / np.float32(255.0)
    x_ref, fl = oracle.quantize_input_u8(u)
    logits = oracle.net_forward(spec, params, x_ref, fl)
    target = np.argsort(-logits, axis=1, kind='stable')[np.arange(N), [0, 1, 4, 5, 9, 0]]
    out, errors = pipeline.forward_loss(model, torch.from_numpy(u).to(dev), torch.from_numpy(target).to(dev), topk=(1, 5))
    np.testing.assert_array_equal(out.cpu().numpy(), logits)
    want = oracle.topk_correct(logits, target, (1, 5))
    assert errors[1] == list(1.0 - want[0]) and errors[5] == list(1.0 - want[1])
    assert errors[1] == [0.0, 1.0, 1.0, 1.0, 1.0, 0.0] and errors[5] == [0.0, 0.0, 0.0, 1.0, 1.0, 0.0]


def test_float_checkpoint_to_gpu_logits(dev, tmp_path):
    """best_model.pt-shaped float checkpoint -> exporter -> IntModel on the GPU == the oracle on the exported integers."""
    from f8net_amd import export
    spec = topology.get('resnet18')
    fstate = synth.make_float_state(spec, seed=9)
    ck = tmp_path / 'best_model.pt'
    torch.save({'model': {'module.' + k: torch.from_numpy(np.asarray(v)) for k, v in fstate.items()}}, ck)
    cfg = export.ExportConfig()
    model = export.int_model_from_float('resnet18', str(ck), cfg).to(dev)
    params = {k: v.cpu().numpy() for k, v in model.state_dict().items()}
    u = synth.rand_uniform_int(4, 'ck', (3, 3, 64, 64), 0, 255).astype(np.float32) / np.float32(255.0)
    got = model.forward_f32(torch.from_numpy(u).to(dev), normalize=False)
    x_ref, fl = oracle.quantize_input_u8(u)
    np.testing.assert_array_equal(got.cpu().numpy(), oracle.net_forward(spec, params, x_ref, fl))


def test_stream_evaluator_matches_oracle_on_host_fed_batches():
    """f8net_amd/stream_eval.py: host-resident uint8 NHWC batches (pageable numpy AND page-locked tensors, a ragged last batch) -> copy
    stream -> f8_net_run_u8 under pipelining mode 2 with the input handed over by event -> top-k on the device.  Logits equal the
    oracle's on the same pixels (fix_train.py:689-692: x_int = the pixel value at fraclen 8), the top-k rates equal numpy's."""
    import torch
    from f8net_amd import stream_eval
    from f8net_amd.net import build_net
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    oracle.build()
    spec = topology.get('resnet18', num_classes=16)
    params = synth.make_params(spec, seed=21)
    B, hw = 4, 64
    net = build_net(spec, params, max_batch=B, hw=hw)
    ev = stream_eval.StreamEvaluator(net, normalize=False, topk=(1, 3))
    rng = np.random.default_rng(5)
    sizes = [4, 4, 4, 4, 4, 3]
    imgs = [rng.integers(0, 256, (n, hw, hw, 3), dtype=np.uint8) for n in sizes]
    labs = [rng.integers(0, 16, (n,)).astype(np.int64) for n in sizes]
    feed = [(torch.from_numpy(im).pin_memory() if i % 2 else im, lb) for i, (im, lb) in enumerate(zip(imgs, labs))]
    r = ev.run(feed, keep_logits=True)
    net.check()
    assert r['images'] == sum(sizes) and r['img_per_s'] > 0
    hit1 = hit3 = 0
    for im, lb, got in zip(imgs, labs, r['logits']):
        want = oracle.net_forward(spec, params, np.ascontiguousarray(im.transpose(0, 3, 1, 2)).astype(np.int32), 8)
        np.testing.assert_array_equal(got, want)
        order = np.argsort(-want, axis=1, kind='stable')
        hit1 += int((order[:, 0] == lb).sum())
        hit3 += int((order[:, :3] == lb[:, None]).any(axis=1).sum())
    assert abs(r['top1'] - hit1 / sum(sizes)) < 1e-12 and abs(r['top3'] - hit3 / sum(sizes)) < 1e-12
    r2 = ev.run(feed[:2])                       # a second epoch on the same evaluator
    assert r2['images'] == 8


@pytest.mark.parametrize('arch', ['mobilenet_v2', 'resnet18'])
@pytest.mark.parametrize('nhwc', [False, True])
def test_uint8_entry_through_the_fused_heads(dev, arch, nhwc):
    """f8_net_run_u8 on whole nets whose head is a row-walking launch (f8_stem.hip): uint8 NCHW planes are read by its loader waves
    through the 3 x 256 table, uint8 NHWC goes through the input launch and the haloed form (the launch's KIND -1 loader).  An unsigned
    fraclen-8 head sees the pixel values themselves: the logits equal those of f8_net_run on the same integers, at 224 (bands, strips)
    and 64 pixels."""
    from f8net_amd.net import build_net
    spec = topology.get(arch, num_classes=20)
    params = synth.make_params(spec, seed=41)
    for hw, n in ((224, 3), (64, 2)):
        u8 = synth.rand_uniform_int(43, f'u8{arch}{hw}', (n, 3, hw, hw), 0, 255).astype(np.uint8)
        net = build_net(spec, params, max_batch=n, hw=hw)
        want = net.run(torch.from_numpy(u8.astype(np.int32)).to(dev)).cpu().numpy()
        img = torch.from_numpy(np.ascontiguousarray(u8.transpose(0, 2, 3, 1)) if nhwc else u8).to(dev)
        got = net.run_u8(img, nhwc=nhwc).cpu().numpy()
        np.testing.assert_array_equal(got, want)
