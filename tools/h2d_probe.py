# Page-locked host -> device copy rate of one uint8 batch (128 x 224 x 224 x 3), one and two copy streams: the bound of bench.py value_host_fed.
#   gpurun -- python tools/h2d_probe.py
import torch, time
n = 128*224*224*3
hs = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(4)]
ds = [torch.empty(n, dtype=torch.uint8, device='cuda') for _ in range(4)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(streams, reps=50):
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(reps):
        with torch.cuda.stream(streams[i % len(streams)]):
            ds[i % 4].copy_(hs[i % 4], non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    return n * reps / dt / 1e9
run([s1], 10)
print('one stream GB/s', run([s1]))
print('two streams GB/s', run([s1, s2]))
big = torch.empty(n*8, dtype=torch.uint8).pin_memory(); dbig = torch.empty(n*8, dtype=torch.uint8, device='cuda')
torch.cuda.synchronize(); t=time.perf_counter(); dbig.copy_(big, non_blocking=True); torch.cuda.synchronize(); print('154 MB copy GB/s', n*8/(time.perf_counter()-t)/1e9)
