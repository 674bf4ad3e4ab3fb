# Host time to SUBMIT one batch in the headline schedule of bench.py (whole-batch launches, three batches in flight) against the time the
# device takes for it: is the net host-bound?     gpurun -- python tools/submit_probe.py mobilenet_v2
import sys, time, torch
sys.path.insert(0, '.')
from f8net_amd import synth, topology
from f8net_amd.net import build_net
arch = sys.argv[1] if len(sys.argv) > 1 else 'mobilenet_v2'
depth = 3
spec = topology.get(arch, normalize=True)
params = synth.reference_params(spec, seed=1234)
net = build_net(spec, params, max_batch=128, hw=224, options={'whole_batch_launches': 1, 'arena_copies': depth, 'pipeline_depth': depth})
net.set_pipelined(2)
x = torch.from_numpy(synth.make_input(spec, params, 128, 224, seed=3)[0]).cuda()
outs = [torch.empty((128, spec.num_classes), dtype=torch.float32, device='cuda') for _ in range(depth + 1)]
for i in range(30): net.run(x, out=outs[i % (depth + 1)])
torch.cuda.synchronize()
K = 300
t = time.perf_counter()
for i in range(K): net.run(x, out=outs[i % (depth + 1)])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'{arch}: host submit {1e6 * (t1 - t) / K:.1f} us per batch, device {1e6 * (t2 - t) / K:.1f} us per batch ({128 * K / (t2 - t):.0f} img/s)')
