"""Soak test of the pipelined schedules: thousands of overlapping runs on varying inputs, every result compared on the GPU
with the strictly ordered result of the same input.  `python tools/soak.py [runs]` (F8_SOAK_ARCH, F8_SOAK_BS: another net / batch)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f8net_amd import synth, topology
from f8net_amd.net import build_net

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
arch = os.environ.get('F8_SOAK_ARCH', 'resnet50')              # resnet50 (default) | resnet18 | mobilenet_v2 | mobilenet_v1
spec = topology.get(arch, normalize=(arch == 'resnet50'))
params = synth.reference_params(spec, seed=1234)
n = int(os.environ.get('F8_SOAK_BS', '128'))
whole = os.environ.get('F8_SOAK_WHOLE', '1') == '1'      # plan as bench.py does: a launch covers the whole batch under mode 2
depth = int(os.environ.get('F8_SOAK_DEPTH', '3'))             # batches in flight under mode 2 (bench.py: 3)
net = build_net(spec, params, max_batch=n, hw=224, options={'whole_batch_launches': 1, 'arena_copies': depth, 'pipeline_depth': depth} if whole else None)
xs = [torch.from_numpy(synth.make_input(spec, params, n, 224, seed=50 + i)[0]).cuda() for i in range(3)]
want = [net.run(x).clone() for x in xs]
torch.cuda.synchronize()
for mode in (2, 1):
    net.set_pipelined(mode)
    nb = depth + 1 if mode == 2 else 2
    outs = [torch.empty((n, 1000), dtype=torch.float32, device='cuda') for _ in range(nb)]
    bad = torch.zeros((), dtype=torch.int64, device='cuda')
    for r in range(runs):
        o = outs[r % nb]
        net.run(xs[r % 3], out=o)
        bad += (o != want[r % 3]).any().to(torch.int64)        # consumer enqueued right after its run
    torch.cuda.synchronize()
    net.check()                                                   # no chain launch timed out, no input out of format
    net.set_pipelined(False)
    print(f'mode {mode}: {runs} overlapping runs, {int(bad.item())} with a wrong result')
    assert int(bad.item()) == 0
print('OK')
