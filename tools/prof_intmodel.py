"""Tuning: does the ORDER in which handles are created (device allocations) change their speed?  (GPU box)"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from f8net_amd import topology, synth
from f8net_amd.net import build_net
spec = topology.get('resnet18')
params = synth.reference_params(spec, seed=1234)
dev = torch.device('cuda:0')
x_np, x_fl = synth.make_input(spec, params, 128, 224, seed=1)
x = torch.from_numpy(x_np).to(dev)
outs = [torch.empty((128, 1000), dtype=torch.float32, device=dev) for _ in range(4)]
opts = {'whole_batch_launches': 1, 'arena_copies': 3, 'pipeline_depth': 3}

def gpu_rate(fn, n=800):
    for i in range(30): fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    return round(128 * n / (time.perf_counter() - t0))

nets = []
for k in range(4):
    n = build_net(spec, params, max_batch=128, hw=224, options=opts); n.upload(); n.set_pipelined(2)
    nets.append(n)
    print(f'net {k} (created {k}-th)  :', gpu_rate(lambda i: n.run(x, out=outs[i % 4])))
for k in range(4):
    print(f'net {k} again           :', gpu_rate(lambda i: nets[k].run(x, out=outs[i % 4])))
del nets[0]; torch.cuda.synchronize()
n = build_net(spec, params, max_batch=128, hw=224, options=opts); n.upload(); n.set_pipelined(2)
print('net created after freeing net 0:', gpu_rate(lambda i: n.run(x, out=outs[i % 4])))
