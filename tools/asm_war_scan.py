#!/usr/bin/env python3
"""Static scan of device assembly: vector instructions that write a register which a v_mfma issued at most WINDOW instructions earlier reads as SrcA /
SrcB.  Written while hunting the run-to-run differences of f8_cchain.hip's float-converter instance (round 6); tools/ubench/ubench_mfma_hazard.hip then showed
that gfx950 INTERLOCKS this write-after-read (no wrong result in 4 M trials at 0 .. 16 wait states), so an occurrence is NOT an error — the tool stays as a
way to look at how close the compiler packs vector code behind MFMAs.   python tools/asm_war_scan.py file.s [window=2]"""
import re
import sys


def regs(tok):
    tok = tok.strip().rstrip(',')
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    return {int(m.group(1))} if m else set()


def main():
    path = sys.argv[1]
    window = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    kernel, hits, recent = None, {}, []
    for ln, line in enumerate(open(path, errors='replace'), 1):
        s = line.strip()
        m = re.match(r'^(_Z\w+):', s)
        if m:
            kernel, recent = m.group(1), []
            continue
        if not s or s.startswith(('.', ';', '//')) or s.endswith(':'):
            continue
        op, _, rest = s.partition(' ')
        if op.startswith('s_nop'):
            recent = []                     # explicit wait states: taken as a guard
            continue
        if op.startswith(('s_', 'ds_', 'buffer_', 'global_', 'scratch_', 'flat_')):
            recent = [(o, r, l, a + 1) for o, r, l, a in recent if a + 1 <= window]
            continue
        ops = [t for t in rest.split(';')[0].split(',')]
        if op.startswith('v_mfma'):
            srcs = regs(ops[1]) | regs(ops[2]) if len(ops) > 2 else set()
            recent = [(o, r, l, a + 1) for o, r, l, a in recent if a + 1 <= window]
            recent.append((s, srcs, ln, 0))
            continue
        if op.startswith('v_'):
            dst = regs(ops[0]) if ops else set()
            if op.startswith('v_permlane32_swap') or op.startswith('v_swap'):
                dst |= regs(ops[1]) if len(ops) > 1 else set()
            for o, r, l, a in recent:
                if dst & r:
                    hits.setdefault(kernel, []).append((ln, s, l, o))
            recent = [(o, r, l, a + 1) for o, r, l, a in recent if a + 1 <= window]
    n = sum(len(v) for v in hits.values())
    for k, v in hits.items():
        print(f'{k}: {len(v)} write(s) to an operand of an MFMA issued <= {window} instructions earlier')
        for ln, s, l, o in v[:3]:
            print(f'    line {ln}: {s[:70]}   <-  line {l}: {o[:80]}')
    print(f'{path}: {n} hazard candidate(s)')
    return 1 if n else 0


if __name__ == '__main__':
    sys.exit(main())
