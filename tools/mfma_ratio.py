#!/usr/bin/env python3
"""Fill the `algorithmic MFMA` / `issued / algorithmic` columns of profiles/rocprof_<tag>_mfma.md from a bench line of the same build
(profiles/bench_<tag>*.json: per_kernel) and the stamped counter file (profiles/pmc_mfma_<arch>_bs<N>.json) — what tools/summarize_prof.py does on the
GPU box when the bench line of the profiled run is at hand.   python tools/mfma_ratio.py profiles/rocprof_r05_mfma.md profiles/bench_r05.json profiles/pmc_mfma_resnet50_bs128.json"""
import json
import re
import sys

md, bench, pmc = sys.argv[1:4]
b = json.load(open(bench))
m = json.load(open(pmc))
assert b['build']['csrc_sha256'] == m['csrc_sha256'][:16], 'bench line and counters come from different kernel sources'
alg = {k: v['alg_ops'] / v['launches'] / 65536.0 for k, v in b['per_kernel'].items() if v.get('launches')}
# a name without template arguments in the bench line stands for the one instance the counters saw
for k in list(alg):
    if '<' not in k:
        for kk in m['kernels']:
            if kk.startswith(k + '<'):
                alg[kk] = alg[k]
out = []
for line in open(md):
    mm = re.match(r'^\| `([^`]+)` \| (\d+) \| ([0-9.e+]+) \| - \| - \|(.*)$', line)
    if mm and mm.group(1) in alg and alg[mm.group(1)] > 0:
        k = mm.group(1)
        issued = m['kernels'][k]['SQ_INSTS_VALU_MFMA_I8']
        line = f"| `{k}` | {mm.group(2)} | {mm.group(3)} | {alg[k]:.4g} | {issued / alg[k]:.3f} |{mm.group(4)}\n"
        m['kernels'][k]['alg_mfma_insts'] = round(alg[k], 1)
        m['kernels'][k]['issued_over_algorithmic'] = round(issued / alg[k], 4)
    out.append(line)
open(md, 'w').writelines(out)
json.dump(m, open(pmc, 'w'), indent=1, sort_keys=True)
print('patched', md)
