"""Soak test of the 7x7 cluster chain (f8_cchain.hip): a stage of identity blocks as a net of its own, many runs over several batch sizes, every
result compared on the GPU with the first (oracle-checked by tests/test_gpu_chain.py) result.  `python tools/soak_chain7.py [runs]`"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from f8net_amd import synth          # noqa: E402
from f8net_amd.net import F8Net      # noqa: E402
import test_gpu_chain as tc          # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
C, MID, HW, nblk = 2048, 512, 7, 2
blocks, fls = tc._stage(C, MID, nblk, C, 'acc_shifts_left')
convs = [c for b in blocks for c in b.body]
params = tc._params(convs, fls, 11, 'soak')
from f8net_amd import topology       # noqa: E402


def tail_net(N, nid=2, rq=0):
    """The stride-2 opening block + `nid` identity blocks behind a 1x1 conv (tests/test_gpu_chain.py, TAIL form): conv1x1, conv3x3 / 2, cluster chain with the join first."""
    CIN0 = 1024
    name = 'o.0'
    body = [topology.ConvSpec(name + '.body.0', CIN0, MID, 1, 1, 0, relu=True), topology.ConvSpec(name + '.body.2', MID, MID, 3, 2, 1, relu=True),
            topology.ConvSpec(name + '.body.4', MID, C, 1, 1, 0)]
    sc = topology.ConvSpec(name + '.shortcut.0', CIN0, C, 1, 2, 0)
    opener = topology.BlockSpec(name, body, sc, residual=True, post_relu=True)
    idb, f = tc._stage(C, MID, nid, C, 'acc_shifts_left')
    f[name + '.body.0'], f[name + '.body.2'], f[name + '.body.4'], f[name + '.shortcut.0'] = (4, 7), (3, 6), (3, 6), (4, 7)
    blks = [opener] + idb
    cv = [c for b in blks for c in b.body] + [sc]
    pre = topology.ConvSpec('pre.0', CIN0, CIN0, 1, 1, 0)
    f['pre.0'] = (4, 7)
    p = tc._params(cv + [pre], f, 51, 'soaktail')
    x = synth.rand_normal_int(23, 'soaktailx', (N, CIN0, 2 * HW, 2 * HW), 3.0e3).astype(np.int32)
    net = F8Net()
    net.set_option('requant_float', rq)
    t = net.input(CIN0, 2 * HW, 2 * HW, 9)
    r = net.conv(t, p['pre.0.weight'], p['pre.0.bias'], stride=1, pad=0, groups=1, weight_fl=7, input_fl=4, input_signed=False, quant_input=True, relu=True)
    for b in blks:
        xin = r
        for c in b.body:
            r = net.conv(r, p[c.key + '.weight'], p[c.key + '.bias'], stride=c.stride, pad=c.pad, groups=1, weight_fl=f[c.key][1], input_fl=f[c.key][0], input_signed=False,
                         quant_input=True, relu=c.relu)
        if b.shortcut is not None:
            c = b.shortcut
            xin = net.conv(xin, p[c.key + '.weight'], p[c.key + '.bias'], stride=2, pad=0, groups=1, weight_fl=f[c.key][1], input_fl=f[c.key][0], input_signed=False,
                           quant_input=True, relu=False)
        r = net.add(r, xin, relu=True)
    net.output(r, as_float=False)
    net.finalize(N)
    assert f'stage_chain_x{nid + 1}_tail' in net.describe()
    return net, x


total_bad = 0
for N, rq in ((5, 0), (130, 0), (130, 1), (33, 1)):
    net, x = tail_net(N, rq=rq)
    xd = torch.from_numpy(x).cuda()
    want = net.run(xd).clone()
    bad = torch.zeros((), dtype=torch.int64, device='cuda')
    for i in range(runs):
        bad += (net.run(xd) != want).any().to(torch.int64)
    torch.cuda.synchronize()
    net.check()
    print(f'TAIL form, N={N}, requant_float={rq}: {runs} runs, {int(bad.item())} with a result different from the first run')
    total_bad += int(bad.item())
for N in (1, 2, 5, 33, 128):
    x = synth.rand_normal_int(7, 'soakx', (N, C, HW, HW), 3.0e3).astype(np.int32)
    net = F8Net()
    t = net.input(C, HW, HW, 9)
    r = t
    for b in blocks:
        xin = r
        for c in b.body:
            r = net.conv(r, params[c.key + '.weight'], params[c.key + '.bias'], stride=1, pad=c.pad, groups=1, weight_fl=fls[c.key][1], input_fl=fls[c.key][0],
                         input_signed=c.signed_in, quant_input=True, relu=c.relu)
        r = net.add(r, xin, relu=True)
    net.output(r, as_float=False)
    net.finalize(N)
    assert f'stage_chain_x{nblk}' in net.describe()
    xd = torch.from_numpy(x).cuda()
    want = net.run(xd).clone()
    bad = torch.zeros((), dtype=torch.int64, device='cuda')
    nbad_el = torch.zeros((), dtype=torch.int64, device='cuda')
    for i in range(runs):
        o = net.run(xd)
        d = (o != want)
        bad += d.any().to(torch.int64)
        nbad_el += d.sum()
    torch.cuda.synchronize()
    net.check()
    print(f'N={N}: {runs} runs, {int(bad.item())} with a wrong result ({int(nbad_el.item())} elements)')
    total_bad += int(bad.item())
print('OK' if total_bad == 0 else 'FAILED')
sys.exit(1 if total_bad else 0)
