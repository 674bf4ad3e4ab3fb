#!/usr/bin/env python3
"""Turn the rocprofv3 databases written by tools/profile.sh into the committed summaries:

    python tools/summarize_prof.py gpurun_out/prof_r01 r01
      -> profiles/rocprof_r01_kernel_stats.md   (rocprofv3 --kernel-trace --stats summary)
      -> profiles/rocprof_r01_pmc.md            (FETCH_SIZE / WRITE_SIZE per kernel, separate passes)
      -> profiles/pmc_traffic.json              (HBM bytes per launch per kernel, read by bench.py)

HBM-byte correction (per /opt/skills/guides/MI355X_MICROARCH.md §HBM): on gfx950 FETCH_SIZE tallies
128-byte fabric requests at 64 B, i.e. reports half the bytes of wide coalesced reads -> doubled here.
Calibration on a known byte count in this code base: f8::input_kernel reads 128*3*224*224*4 =
77.07 MB and FETCH_SIZE reports 38.6 MB.  WRITE_SIZE is taken as reported (KB).
"""
import json
import os
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'^void ', '', name)
    return re.sub(r'\(.*\)$', '', name)


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, 'profiles')
    os.makedirs(out, exist_ok=True)
    c = sqlite3.connect(os.path.join(src, 'trace', 'trace_results.db'))
    rows = list(c.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    with open(os.path.join(out, f'rocprof_{tag}_kernel_stats.md'), 'w') as f:
        f.write(f'# rocprofv3 --kernel-trace --stats — `python bench.py --steps 5 --warmup 2 --no-cpu-baseline` ({tag})\n\n')
        f.write('Durations in microseconds (7 steps + 7 profiled passes = 14 forward passes of ResNet-50, bs 128; every launch covers the whole batch — pipelining mode 2, F8_SPLIT_STREAMS=0 so that no two runs overlap).\n\n')
        f.write('| kernel | calls | total us | avg us | % |\n|---|---:|---:|---:|---:|\n')
        for name, calls, tot, avg, pct in rows:
            f.write(f'| `{short(name)}` | {calls} | {tot:.1f} | {avg:.2f} | {pct:.2f} |\n')
    traffic = {}
    for cname, sub in (('FETCH_SIZE', 'pmc_fetch'), ('WRITE_SIZE', 'pmc_write')):
        c = sqlite3.connect(os.path.join(src, sub, 'pmc_results.db'))
        q = ("select kernel_name, count(*), avg(value) from counters_collection "
             "where counter_name = ? group by kernel_name")
        for name, n, avg_kb in c.execute(q, (cname,)):
            e = traffic.setdefault(short(name), {})
            e[cname + '_KB_avg'] = avg_kb
            e['launches'] = n
    for k, e in traffic.items():
        f_kb, w_kb = e.get('FETCH_SIZE_KB_avg', 0.0), e.get('WRITE_SIZE_KB_avg', 0.0)
        e['hbm_bytes_per_launch'] = round(2 * f_kb * 1024 + w_kb * 1024)
    with open(os.path.join(out, f'rocprof_{tag}_pmc.md'), 'w') as f:
        f.write(f'# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), same command ({tag})\n\n')
        f.write('Per-launch averages. `hbm bytes` = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE correction, see '
                'tools/summarize_prof.py).\n\n| kernel | launches | FETCH_SIZE KB | WRITE_SIZE KB | hbm MB / launch |\n|---|---:|---:|---:|---:|\n')
        for k, e in sorted(traffic.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch']):
            if not k.startswith('f8::'):
                continue
            f.write(f"| `{k}` | {e['launches']} | {e.get('FETCH_SIZE_KB_avg', 0):.0f} | {e.get('WRITE_SIZE_KB_avg', 0):.0f} | "
                    f"{e['hbm_bytes_per_launch'] / 1e6:.1f} |\n")
    json.dump({k: v for k, v in traffic.items() if k.startswith('f8::')},
              open(os.path.join(out, 'pmc_traffic.json'), 'w'), indent=1, sort_keys=True)
    print('wrote', sorted(os.listdir(out)))


if __name__ == '__main__':
    main()
