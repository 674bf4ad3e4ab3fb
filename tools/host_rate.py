"""How fast can the host enqueue runs?  (Measured: 105 us per run in pipelining mode 2, 810 us in mode 1, against 1.65 ms of
GPU time: the host is never the limit; the ~0.5 ms a stream idles between its runs in a rocprofv3 timeline — tools/overlap.py —
is not a dependency either: removing every inter-stream event changed nothing.  The two in-flight runs simply saturate the
chip; the stream whose next kernel cannot get CUs waits.)  Times the Python loop that submits `n` pipelined runs (no synchronisation inside)
and the GPU's completion of the same runs."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f8net_amd import synth, topology
from f8net_amd.net import build_net

spec = topology.get('resnet50', normalize=True)
params = synth.make_params(spec, seed=1234, fraclens=topology.R50_NVIDIA_FRACLENS)
net = build_net(spec, params, max_batch=128, hw=224)
x = torch.from_numpy(synth.make_input(spec, params, 128, 224, seed=7)[0]).cuda()
outs = [torch.empty((128, 1000), dtype=torch.float32, device='cuda') for _ in range(2)]
side = torch.cuda.Stream()
for mode, strm in ((2, None), (2, side), (1, side)):
    ctx = torch.cuda.stream(strm) if strm is not None else torch.cuda.stream(torch.cuda.current_stream())
    ctx.__enter__()
    net.set_pipelined(mode)
    for i in range(10):
        net.run(x, out=outs[i & 1])
    torch.cuda.synchronize()
    n = 100
    t0 = time.perf_counter()
    for i in range(n):
        net.run(x, out=outs[i & 1])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ctx.__exit__(None, None, None)
    print(f'mode {mode} on {"a side stream" if strm is not None else "the default stream"}: host submits a run in {(t1 - t0) / n * 1e6:.0f} us; GPU finishes one every {(t2 - t0) / n * 1e6:.0f} us')
    net.set_pipelined(False)
