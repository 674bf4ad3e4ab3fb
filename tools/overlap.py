#!/usr/bin/env python3
"""Occupancy statistics of a rocprofv3 --kernel-trace database over the steady part of a run: share of the wall time with
0 / 1 / >= 2 of our kernels resident, and the gaps between consecutive kernels of each stream.

    python tools/overlap.py gpurun_out/prof_x/trace/trace_results.db [lo hi]      # window as fractions of the trace, default 0.4 0.9
"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    rows = [r for r in c.execute('select name, stream_id, start, end from kernels order by start') if 'f8::' in r[0]]
    lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.4, 0.9)
    t_a, t_b = rows[0][2], rows[-1][3]
    w0, w1 = t_a + lo * (t_b - t_a), t_a + hi * (t_b - t_a)
    ks = [r for r in rows if r[2] >= w0 and r[3] <= w1]
    ev = sorted([(r[2], 1) for r in ks] + [(r[3], -1) for r in ks])
    occ = {0: 0, 1: 0, 2: 0}
    cur, last = 0, ev[0][0]
    for t, d in ev:
        occ[min(cur, 2)] += t - last
        cur += d
        last = t
    wall = ev[-1][0] - ev[0][0]
    busy = sum(r[3] - r[2] for r in ks)
    outs = sum('output_kernel' in r[0] for r in ks)
    print(f'{len(ks)} kernels, {outs} passes in {wall / 1e3:.1f} us  ({wall / 1e3 / max(outs, 1):.1f} us per pass, sum(kernel) {busy / 1e3 / max(outs, 1):.1f} us per pass)')
    print(f'idle {100 * occ[0] / wall:.1f} %   1 resident {100 * occ[1] / wall:.1f} %   >= 2 resident {100 * occ[2] / wall:.1f} %')
    by = {}
    for r in ks:
        by.setdefault(r[1], []).append(r)
    for sid, lst in sorted(by.items()):
        lst.sort(key=lambda r: r[2])
        g = sorted(b[2] - a[3] for a, b in zip(lst, lst[1:]))
        if g:
            print(f'  stream {sid}: {len(lst)} kernels, gaps median {g[len(g) // 2] / 1e3:.2f} us, p90 {g[int(len(g) * 0.9)] / 1e3:.2f} us, total {sum(g) / 1e3:.1f} us')


if __name__ == '__main__':
    main()
