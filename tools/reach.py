#!/usr/bin/env python3
"""Which kernels do the DEFAULT plans of the four nets reach?  Planning needs no GPU:
    python tools/reach.py            -> markdown table (kernel symbol x net / batch), used by DESIGN.md §4
Batches 1 / 8 / 128 / 256 at 224x224; 128 and 256 also with `whole_batch_launches` (what bench.py plans under pipelining mode 2)."""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f8net_amd import synth, topology        # noqa: E402
from f8net_amd.net import build_net          # noqa: E402

NETS = ['resnet50', 'resnet18', 'mobilenet_v2', 'mobilenet_v1']
CASES = [(1, 0), (8, 0), (128, 0), (128, 1), (256, 1)]


def family(sym):
    return re.sub(r'<.*', '', sym.replace('f8::', ''))


def main():
    reach = {}
    for arch in NETS:
        spec = topology.get(arch, normalize=(arch == 'resnet50'))
        params = synth.reference_params(spec, seed=1234)
        for bs, whole in CASES:
            net = build_net(spec, params, max_batch=bs, hw=224, options={'whole_batch_launches': 1} if whole else None)
            for i in range(net.num_launches):
                k = net.launch_kernel(i)
                if k:
                    reach.setdefault(k, set()).add((arch, bs, whole))
    fams = {}
    for k, v in reach.items():
        fams.setdefault(family(k), []).append((k, v))
    print('| kernel family | instances reached | nets (batch sizes) |\n|---|---:|---|')
    for f in sorted(fams):
        inst = fams[f]
        where = {}
        for _, v in inst:
            for arch, bs, whole in v:
                where.setdefault(arch, set()).add(f'{bs}{"w" if whole else ""}')
        txt = '; '.join(f'{a}: {", ".join(sorted(b, key=lambda x: (int(x.rstrip("w")), x)))}' for a, b in sorted(where.items()))
        print(f'| `{f}` | {len(inst)} | {txt} |')
    if '-v' in sys.argv:
        for f in sorted(fams):
            for k, v in sorted(fams[f]):
                print(' ', k, sorted(v))


if __name__ == '__main__':
    main()
