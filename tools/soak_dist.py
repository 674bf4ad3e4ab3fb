"""Soak of the N > 1 exchange on ONE GPU: a one-rank nccl (= RCCL) group with the collective forced (f8net_amd/dist.py), the pipelined schedule bench.py
uses (three batches in flight), thousands of runs on alternating inputs; every gathered result is compared on the GPU with the strictly ordered result of
the same input.  `python tools/soak_dist.py [runs]`"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f8net_amd import dist as f8dist
from f8net_amd import synth, topology
from f8net_amd.net import build_net

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', str(29300 + os.getpid() % 200))
torch.cuda.set_device(0)
dist.init_process_group(backend='nccl', rank=0, world_size=1)
dev = torch.device('cuda', 0)
spec = topology.get('resnet50', normalize=True)
params = synth.reference_params(spec, seed=1234)
n, depth = 128, 3
net = build_net(spec, params, max_batch=n, hw=224, options={'whole_batch_launches': 1, 'arena_copies': depth, 'pipeline_depth': depth})
xs = [torch.from_numpy(synth.make_input(spec, params, n, 224, seed=70 + i)[0]).to(dev) for i in range(3)]
want = [net.run(x).clone() for x in xs]
torch.cuda.synchronize()
net.set_pipelined(2)
pf = f8dist.PipelinedShardedForward(lambda t, out: net.run(t, out=out), spec.num_classes, n, dev, lagged=True, depth=depth, force_collective=True)
bad = torch.zeros((), dtype=torch.int64, device=dev)
pending = []
for r in range(runs):
    full = pf(xs[r % 3])
    pending.append((r, full))
    if len(pending) == depth:             # the oldest pair is about to be reused: its collective is waited for inside the next call; check it now
        r0, f0 = pending.pop(0)
        pf.work[r0 % depth].wait() if pf.work[r0 % depth] is not None else None
        bad += (f0 != want[r0 % 3]).any().to(torch.int64)
pf.finish()
for r0, f0 in pending:
    bad += (f0 != want[r0 % 3]).any().to(torch.int64)
torch.cuda.synchronize()
net.check()
print(f'{runs} pipelined runs through a forced one-rank RCCL all-gather, {int(bad.item())} with a wrong result')
dist.destroy_process_group()
sys.exit(1 if int(bad.item()) else 0)
