"""Runs a few ResNet-50 bs-128 forwards so that a -DF8_TRACE build prints its per-workgroup phase
timings: F8_TRACE_FUSED=<k> / F8_TRACE_PATCH=<k> / F8_TRACE_LAUNCH=<k> select the k-th call of each kernel instance."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from f8net_amd import synth, topology
from f8net_amd.net import build_net

spec = topology.get('resnet50', normalize=True)
params = synth.make_params(spec, seed=1234, fraclens=topology.R50_NVIDIA_FRACLENS)
whole = os.environ.get('F8_TRACE_WHOLE', '1') == '1'      # 1: one launch per step covers the batch, as bench.py runs (mode 2); 0: two 64-image sub-batches
net = build_net(spec, params, max_batch=128, hw=224, options={'whole_batch_launches': 1} if whole else None)
if whole:
    net.set_pipelined(2)
x, fl = synth.make_input(spec, params, 128, 224, seed=7)
xd = torch.from_numpy(x).cuda()
for i in range(6):
    net.run(xd)
    torch.cuda.synchronize()
