#!/usr/bin/env python3
"""Timeline statistics of one rocprofv3 --kernel-trace database: per forward pass (input_kernel .. output_kernel)
wall time, sum of kernel durations, time with 0 / 1 / >=2 kernels resident, and per-stream gaps between kernels.

    python tools/timeline.py gpurun_out/prof_x/trace/trace_results.db
"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    rows = list(c.execute('select name, stream_id, start, end from kernels order by start'))
    rows = [r for r in rows if 'f8::' in r[0]]
    # passes: from the first input_kernel to the last output_kernel of a group (parts run on separate streams)
    outs = [i for i, r in enumerate(rows) if 'output_kernel' in r[0]]
    ins = [i for i, r in enumerate(rows) if 'input_kernel' in r[0]]
    parts = 2 if len(ins) >= 2 and rows[ins[1]][2] < rows[outs[0]][2] else 1
    n_pass = len(outs) // parts
    print(f'{len(rows)} kernels, {n_pass} passes, {parts} sub-batches per pass')
    res = []
    for p in range(n_pass):
        t0 = rows[ins[p * parts]][2]
        t1 = max(rows[o][3] for o in outs[p * parts:(p + 1) * parts])
        ks = [r for r in rows if r[2] >= t0 and r[3] <= t1]
        ev = sorted([(r[2], 1) for r in ks] + [(r[3], -1) for r in ks])
        occ = {0: 0, 1: 0, 2: 0}
        cur, last = 0, t0
        for t, d in ev:
            occ[min(cur, 2)] += t - last
            cur += d
            last = t
        gaps = {}
        by_stream = {}
        for r in ks:
            by_stream.setdefault(r[1], []).append(r)
        for sid, lst in by_stream.items():
            lst.sort(key=lambda r: r[2])
            g = [b[2] - a[3] for a, b in zip(lst, lst[1:])]
            gaps[sid] = (len(g), sum(g) / 1e3, (sorted(g)[len(g) // 2] / 1e3) if g else 0)
        res.append((t1 - t0, sum(r[3] - r[2] for r in ks), occ, gaps, len(ks)))
    sel = res if len(sys.argv) < 3 else res[int(sys.argv[2]):int(sys.argv[3])]
    for i, (wall, busy, occ, gaps, n) in enumerate(sel):
        print(f'pass {i}: wall {wall/1e3:8.1f} us  sum(kernel) {busy/1e3:8.1f} us  kernels {n}  '
              f'idle {occ[0]/1e3:6.1f} us  1 resident {occ[1]/1e3:7.1f} us  >=2 resident {occ[2]/1e3:7.1f} us')
        for sid, (k, tot, med) in gaps.items():
            print(f'     stream {sid}: {k} gaps, total {tot:7.1f} us, median {med:5.2f} us')


if __name__ == '__main__':
    main()
