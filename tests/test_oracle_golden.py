"""Pin the CPU oracle (oracle/) against fixtures captured from the imported reference
(oracle/gen_golden.py, build container).  Bit-exact: integer arithmetic."""
import os

import numpy as np
import pytest

from f8net_amd import synth, topology
from oracle import oracle
from oracle.gen_golden import checksum


@pytest.fixture(scope='module')
def ops(golden_dir):
    return np.load(os.path.join(golden_dir, 'ops.npz'))


def test_requant_known_answers(ops):
    vec = ops['requant/in']
    cases = ops['requant/cases']
    assert len(cases) > 100
    for dst_fl, src_fl, signed in cases.tolist():
        want = ops[f'requant/out_{dst_fl}_{src_fl}_{signed}']
        got = oracle.requant(vec, dst_fl, src_fl, bool(signed))
        np.testing.assert_array_equal(got, want, err_msg=f'{dst_fl=} {src_fl=} {signed=}')
        # independent big-int statement on a slice
        np.testing.assert_array_equal(oracle.requant_py(vec[:80], dst_fl, src_fl, bool(signed)), want[:80])


def test_requant_survey_table():
    """SURVEY.md App. B rows (outputs produced by the imported reference)."""
    v = np.array([-13, -12, -11, -10, -9, -8, -7, -6, -5, -4, -3, -2, -1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10,
                  11, 12, 13, 2000, -2000, 1020, 1022, 1018], dtype=np.int32)
    r = oracle.requant(v, 4, 6, True)
    assert r.tolist() == [-3, -3, -3, -2, -2, -2, -2, -2, -1, -1, -1, 0, 0, 0, 0, 0, 1, 1, 1, 2, 2, 2, 2, 2,
                          3, 3, 3, 127, -127, 127, 127, 127]
    r = oracle.requant(v, 3, 6, False)
    assert r.tolist() == [0] * 18 + [1, 1, 1, 1, 1, 1, 1, 2, 2, 250, 0, 128, 128, 127]
    r = oracle.requant(v, 7, 5, True)
    assert r[:27].tolist() == [4 * i for i in range(-13, 14)] and r[27:].tolist() == [127, -127, 127, 127, 127]


def test_requant_rejects_what_reference_asserts():
    v = np.zeros(4, dtype=np.int32)
    with pytest.raises(AssertionError):
        oracle.requant(v, 8, 10, True)     # fl <= wl-1 for signed (fix_quant_ops.py:93-94)
    with pytest.raises(AssertionError):
        oracle.requant(v, 9, 10, False)
    with pytest.raises(AssertionError):
        oracle.requant(v, -1, 10, False)


def test_conv_linear(ops):
    gi = 0
    while f'conv/{gi}/geom' in ops:
        N, C, H, W, K, k, s, p, g = ops[f'conv/{gi}/geom'].tolist()
        y = oracle.conv2d(ops[f'conv/{gi}/x'], ops[f'conv/{gi}/w'], ops[f'conv/{gi}/b'], s, p, g)
        np.testing.assert_array_equal(y, ops[f'conv/{gi}/y'], err_msg=f'geom {gi}')
        gi += 1
    assert gi >= 8
    y = oracle.conv2d(np.full((1, 1, 2, 2), 65537, np.int32), np.full((1, 1, 1, 1), 65536, np.int32),
                      np.array([7], np.int32), 1, 0)
    np.testing.assert_array_equal(y, ops['conv/wrap_y'])
    y = oracle.linear(ops['linear/x'], ops['linear/w'], ops['linear/b'])
    np.testing.assert_array_equal(y, ops['linear/y'])
    np.testing.assert_array_equal(y.astype(np.float32), ops['linear/y_float'])


def test_pools_add_input(ops):
    np.testing.assert_array_equal(oracle.avgpool_sum(ops['avgpool/x']), ops['avgpool/y'])
    assert ops['avgpool/fl'].tolist() == [13, 13 + oracle.AVGPOOL_SHIFT]
    y = oracle.maxpool(ops['maxpool/x'])
    np.testing.assert_array_equal(y, ops['maxpool/y'])
    np.testing.assert_array_equal(y, ops['maxpool/y_fxq'])
    for res_fl, x_fl in ((11, 9), (9, 12), (10, 10)):
        got, fl = oracle.add_align(ops['add/res'], ops['add/x'], res_fl, x_fl)
        np.testing.assert_array_equal(got, ops[f'add/out_{res_fl}_{x_fl}'])
        assert fl == max(res_fl, x_fl)
    q, fl = oracle.quantize_input_u8(ops['inq/img'])
    np.testing.assert_array_equal(q, ops['inq/u8'])
    q, fl = oracle.quantize_input_normalized(ops['inq/xn'], 5)
    np.testing.assert_array_equal(q, ops['inq/s8_fl5'])
    # the same from uint8 pixels (every pixel value in every channel): ToTensor / Normalize / fix_quant executed by torch
    q, fl = oracle.quantize_input_pixels(ops['inq8/u8'])
    np.testing.assert_array_equal(q, ops['inq8/plain'])
    assert fl == 8 and np.array_equal(q, ops['inq8/u8'].astype(np.int32))
    for f in (4, 5, 6):
        q, fl = oracle.quantize_input_pixels(ops['inq8/u8'], True, ops['inq8/mean'], ops['inq8/std'], f, True)
        np.testing.assert_array_equal(q, ops[f'inq8/s8_fl{f}'])


@pytest.mark.parametrize('arch', ['resnet18', 'resnet50', 'mobilenet_v1', 'mobilenet_v2'])
def test_whole_net_vs_reference(arch, golden_dir):
    g = np.load(os.path.join(golden_dir, f'net_{arch}.npz'))
    spec = topology.get(arch, normalize=bool(g['normalize']))
    params = synth.reference_params(spec, seed=1234)
    for hw, n in ((64, 2), (224, 1)):
        tag = f's1234_hw{hw}_n{n}'
        x, x_fl = synth.make_input(spec, params, n, hw, seed=7)
        caps = {}
        logits = oracle.net_forward(spec, params, x, x_fl, tap=lambda k, t, fl: caps.__setitem__(k, checksum(t)))
        names = [str(s) for s in g[f'{tag}/cap_names']]
        sums = g[f'{tag}/cap_sums']
        for name, want in zip(names, sums):
            np.testing.assert_array_equal(caps[name], want, err_msg=f'{arch} {tag} layer {name}')
        np.testing.assert_array_equal(logits, g[f'{tag}/logits'])
        if f'{tag}/integize_logits' in g.files:
            # the reference's float-carried "integize" evaluation of the same IntModel, fed the real value of
            # the same input integers — float32 carries these accumulators exactly, so it returns the very same logits
            assert bool(g[f'{tag}/integize_equal'])
            np.testing.assert_array_equal(logits, g[f'{tag}/integize_logits'])


def test_topk_scoring_matches_reference(ops):
    """fix_train.py:697-704 executed by torch (gen_golden.py) vs the oracle's rank formulation."""
    got = oracle.topk_correct(ops['topk/logits'], ops['topk/target'], (1, 5))
    np.testing.assert_array_equal(got, ops['topk/correct'])
    assert got[0].sum() >= 1 and got[1].sum() > got[0].sum()      # hits at k = 1 and more at k = 5
