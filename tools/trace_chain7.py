"""Phase cycle counters of the 7x7 cluster chain inside ResNet-50 (bs 128): needs a -DF8_TRACE build of f8_cchain.hip
   tools/variant.sh trace f8_cchain -DF8_TRACE;  F8NET_LIB=f8net_amd/libf8net_trace.so F8_TRACE_CHAIN7=3 python tools/trace_chain7.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f8net_amd import synth, topology      # noqa: E402
from f8net_amd.net import build_net        # noqa: E402

n = int(os.environ.get('F8_TRACE_BS', '128'))
spec = topology.get('resnet50', normalize=True)
params = synth.reference_params(spec, seed=1234)
net = build_net(spec, params, max_batch=n, hw=224, options={'whole_batch_launches': 1, 'split': 1})
x = torch.from_numpy(synth.make_input(spec, params, n, 224, seed=50)[0]).cuda()
for _ in range(6):
    net.run(x)
torch.cuda.synchronize()
net.check()
