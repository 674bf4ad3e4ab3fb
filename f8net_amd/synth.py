"""Deterministic synthetic integer parameters and inputs.

There is no network here (no Model-Zoo checkpoints, no ImageNet), and a randomly initialised
float model exports to degenerate integer nets.  So parameters are synthesised
directly in the exported format (`int_conv` / `int_fc`, fix_quant_ops.py:680-714, 1165-1195):
int32 weights in [-127,127], int32 bias at fraclen in_fl+w_fl, per-layer scalar fraclens.

The generator is a counter-based splitmix64 hash evaluated with numpy uint64 arithmetic only, so
the same (seed, key) gives the same integers on every machine — the golden fixtures captured from
the reference in the build container and the GPU-box tests regenerate identical nets.
"""
import zlib

import numpy as np

from . import topology

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _splitmix(z):
    with np.errstate(over='ignore'):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def rand_u64(seed: int, n: int, stream: int = 0) -> np.ndarray:
    with np.errstate(over='ignore'):
        base = np.uint64((seed * 0x632BE59BD9B4E019 + stream * 0xD6E8FEB86659FD93) & (2**64 - 1))
        idx = np.arange(1, n + 1, dtype=np.uint64)
        return _splitmix(base + idx * _GOLD)


def _stream(key: str) -> int:
    return zlib.crc32(key.encode())


def rand_uniform_int(seed: int, key: str, shape, lo: int, hi: int) -> np.ndarray:
    """Integers uniform in [lo, hi] (inclusive)."""
    n = int(np.prod(shape)) if len(shape) else 1
    r = rand_u64(seed, n, _stream(key))
    span = np.uint64(hi - lo + 1)
    return ((r >> np.uint64(11)) % span).astype(np.int64).reshape(shape) + lo


def rand_normal_int(seed: int, key: str, shape, sigma: float) -> np.ndarray:
    """round(N(0, sigma)) approximated by an Irwin-Hall sum of 8 uniform 16-bit draws (int64)."""
    n = int(np.prod(shape)) if len(shape) else 1
    acc = np.zeros(n, dtype=np.float64)
    for j in range(2):
        r = rand_u64(seed, n, _stream(key) + 0x1000 * (j + 1))
        for b in range(4):
            acc += ((r >> np.uint64(16 * b)) & np.uint64(0xFFFF)).astype(np.float64)
    # mean 8*32767.5, variance 8*(65536^2-1)/12
    acc -= 8 * 32767.5
    std = np.sqrt(8 * (65536.0 ** 2 - 1) / 12.0)
    return np.rint(acc * (sigma / std)).astype(np.int64).reshape(shape)


def make_params(spec: topology.NetSpec, seed: int = 1234, fraclens=None,
                w_sigma: float = 24.0, auto_range: bool = False) -> dict:
    """Exported-IntModel-shaped parameter dict (numpy int32), keys as `state_dict()`.

    fraclens: optional {key: (input_fl, weight_fl)}; missing keys get seeded draws
    (w_fl in {5,6,7}; in_fl in {2..6}, or {1..5} for signed inputs).  The head of a
    non-normalised net reads unsigned 0..255 at fraclen 8 (fix_quant_ops.py:486-488).
    """
    fraclens = fraclens or {}
    p = {}
    layers = [(c.key, (c.cout, c.cin // c.groups, c.k, c.k), c.signed_in) for c in spec.convs()]
    layers.append((spec.fc_key, (spec.num_classes, spec.fc_in), spec.fc_signed_in))
    # auto_range (real fraclen tables whose formats assume TRAINED weight magnitudes, e.g. MobileNet-V2's unsigned fraclen-8
    # project inputs): the weight spread of a layer is chosen so that its accumulators, requantised into the NEXT layer's
    # format, land mid-range instead of saturating every value: sigma = 48 * 2^n / (x_rms * sqrt(fan)), n = the next requant shift
    nxt = {}
    if auto_range:
        seq = [spec.head.key]
        for b in spec.blocks:
            seq.extend(c.key for c in b.body)
        if spec.tail is not None:
            seq.append(spec.tail.key)
        seq.append(spec.fc_key)
        nxt = {a: b for a, b in zip(seq, seq[1:])}
    for key, wshape, signed in layers:
        if key in fraclens:
            in_fl, w_fl = fraclens[key]
        else:
            w_fl = int(rand_uniform_int(seed, key + '/wfl', (), 5, 7))
            in_fl = int(rand_uniform_int(seed, key + '/ifl', (), 1, 5) if signed
                        else rand_uniform_int(seed, key + '/ifl', (), 2, 6))
            if key == 'head.0' and not spec.normalize:
                in_fl = 8
        fan = int(np.prod(wshape[1:]))
        # keep accumulators well inside int32 and outputs in a useful dynamic range:
        # depthwise layers (fan 9) get wider weights, as in the reference's logs (w_fl 0..1 there)
        sig = w_sigma if fan > 16 else 40.0
        b_sigma = 0.5 * 2.0 ** (in_fl + w_fl)
        if auto_range and key in nxt and nxt[key] in fraclens:
            n = in_fl + w_fl - fraclens[nxt[key]][0] + (6 if nxt[key] == spec.fc_key else 0)
            acc = 48.0 * 2.0 ** n / (30.0 if nxt[key] == spec.fc_key else 1.0)      # avg-pool: sum of 49 mostly positive values
            sig = float(min(45.0, max(1.5, acc / (56.0 * fan ** 0.5))))
            b_sigma = 0.25 * sig * 56.0 * fan ** 0.5
        w = np.clip(rand_normal_int(seed, key + '/w', wshape, sig), -127, 127)
        b = rand_normal_int(seed, key + '/b', (wshape[0],), b_sigma)
        p[key + '.weight'] = w.astype(np.int32)
        p[key + '.bias'] = b.astype(np.int32)
        p[key + '.weight_fraclen'] = np.array(w_fl, dtype=np.int32)
        p[key + '.input_fraclen'] = np.array([in_fl], dtype=np.int32)
    return p


def reference_params(spec: topology.NetSpec, seed: int = 1234) -> dict:
    """Parameters on the reference's own learned fraclen table where its logs hold one (ResNet-50: NVIDIA-pretrained run;
    MobileNet-V2: the mbv2_fix_quant log, with range-matched weights), seeded draws otherwise.  What the net goldens,
    `bench.py` and the full-size tests use."""
    fr = topology.real_fraclens(spec.arch)
    if fr is not None and not spec.normalize:
        # the exporter pins the head of a non-normalised net to unsigned 0..255 at fraclen 8 (fix_quant_ops.py:486-488; the
        # MobileNet-V2 log predates that code and prints 6): IntModel.forward feeds the head conv without requantising
        fr = dict(fr)
        fr[spec.head.key] = (8, fr[spec.head.key][1])
    return make_params(spec, seed=seed, fraclens=fr, auto_range=(spec.arch == 'mobilenet_v2'))


def make_input(spec: topology.NetSpec, params: dict, n: int, hw: int = 224, seed: int = 1) -> tuple:
    """int32 NCHW network input + its fraclen, as `forward_loss` hands it to IntModel
    (fix_train.py:683-692): u8 0..255 at fraclen 8, or signed [-127,127] at head.input_fraclen."""
    if spec.normalize:
        x = rand_uniform_int(seed, 'input', (n, 3, hw, hw), -127, 127)
        fl = int(params['head.0.input_fraclen'][0])
    else:
        x = rand_uniform_int(seed, 'input', (n, 3, hw, hw), 0, 255)
        fl = 8
    return x.astype(np.int32), fl


def make_float_state(spec: topology.NetSpec, seed: int = 77) -> dict:
    """Synthetic FLOAT model state (the keys of a trained F8Net ResNet `state_dict()`: `….conv.weight`, `….bn.*`,
    `….alpha`, `….input_fraclen`, `classifier.0.weight/bias`) for the exporter parity tests.  Every value is an integer
    from the counter PRNG divided by a power of two, so the float32 arrays are bit-identical wherever they are rebuilt."""
    from .export import float_key
    sd = {}

    def f(key, name, shape, sigma_q, shift, offset_q=0):
        v = rand_normal_int(seed, f'{key}.{name}', shape, sigma_q) + offset_q
        return (v.astype(np.float64) / float(1 << shift)).astype(np.float32)

    def u(key, name, shape, lo_q, hi_q, shift):
        v = rand_uniform_int(seed, f'{key}.{name}', shape, lo_q, hi_q)
        return (v.astype(np.float64) / float(1 << shift)).astype(np.float32)

    for c in spec.convs():
        fk = float_key(c.key)
        fan = c.k * c.k * c.cout
        sig = (2.0 / fan) ** 0.5
        sd[f'{fk}.conv.weight'] = f(c.key, 'w', (c.cout, c.cin // c.groups, c.k, c.k), sig * (1 << 16), 16)
        sd[f'{fk}.bn.weight'] = u(c.key, 'bnw', (c.cout,), 512, 1536, 10)            # 0.5 .. 1.5
        sd[f'{fk}.bn.bias'] = f(c.key, 'bnb', (c.cout,), 0.2 * 1024, 10)
        sd[f'{fk}.bn.running_mean'] = f(c.key, 'bnm', (c.cout,), 0.3 * 1024, 10)
        sd[f'{fk}.bn.running_var'] = u(c.key, 'bnv', (c.cout,), 256, 2048, 10)       # 0.25 .. 2.0
        sd[f'{fk}.alpha'] = u(c.key, 'alpha', (), 3 * 256, 9 * 256, 8)               # 3 .. 9
        sd[f'{fk}.input_fraclen'] = u(c.key, 'infl', (1,), 67, 109, 4)               # 4.19 .. 6.81 in steps of 1/16: exact .5 ties occur
    k = spec.fc_key
    sd[f'{k}.weight'] = f(k, 'w', (spec.num_classes, spec.fc_in), 0.05 * (1 << 16), 16)
    sd[f'{k}.bias'] = f(k, 'b', (spec.num_classes,), 0.1 * 1024, 10)
    sd[f'{k}.alpha'] = u(k, 'alpha', (), 3 * 256, 9 * 256, 8)
    sd[f'{k}.input_fraclen'] = u(k, 'infl', (1,), 67, 109, 4)
    return sd
