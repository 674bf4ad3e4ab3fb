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
total_bad = 0
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
