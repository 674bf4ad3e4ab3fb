#!/usr/bin/env python3
"""Instruction mix of a kernel between consecutive s_barrier instructions, from hipcc -S output (static counts, in program order).

    python tools/asm_phases.py build/asm/f8_chain.s '_ZN2f812chain_kernelILi256ELi64ELi56ELi56ELi4ELi64ELi2ELi2ELi1ELb0ELb0EEE'

rocprofv3 --att is not usable on this image (no rocprof-trace-decoder library: profiles/att_probe_r05.log), so the instruction-level picture of
the chain launches is assembled from (a) this static per-segment mix and (b) the per-wave cycle stamps of the -DF8_TRACE build.
"""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_'): return 'salu'
    if op.startswith('v_'): return 'valu'
    return 'other'


def main():
    path, sym = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    lines = open(path, errors='replace').read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith(sym) and l.rstrip().endswith(('EEvNS_9ChainArgsE', ':')) or (l.startswith(sym) and ':' in l))
    seg, segs, label = Counter(), [], None
    ops = Counter()
    blocks = []
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith('.Lfunc_end') or s.startswith('.section'):
            break
        m = re.match(r'^(\.LBB\d+_\d+):', s)
        if m:
            blocks.append((m.group(1), len(segs)))
            continue
        if not s or s.startswith((';', '.', '//')):
            continue
        op = s.split()[0]
        c = classify(op)
        seg[c] += 1
        if c == 'valu':
            ops[op] += 1
        if c == 'barrier':
            segs.append((seg, ops)); seg, ops = Counter(), Counter()
    segs.append((seg, ops))
    print(f'{len(segs)} segments (split at s_barrier), {len(blocks)} basic blocks')
    print('seg  valu mfma  lds vmem salu wait  nop | top VALU ops')
    for i, (c, o) in enumerate(segs):
        t = ' '.join(f'{k}:{v}' for k, v in o.most_common(top))
        print(f'{i:3d} {c["valu"]:5d} {c["mfma"]:4d} {c["lds"]:4d} {c["vmem"]:4d} {c["salu"]:4d} {c["waitcnt"]:4d} {c["nop"]:4d} | {t}')


if __name__ == '__main__':
    main()
