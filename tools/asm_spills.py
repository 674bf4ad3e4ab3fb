"""Tuning helper: per kernel of a .s file (hipcc -S --cuda-device-only), code size, MFMA / scratch instruction counts and where the
scratch instructions sit (line ranges between barriers).  python tools/asm_spills.py file.s [substring of the mangled name]"""
import re
import sys

src = open(sys.argv[1]).read().split('\n')
want = sys.argv[2] if len(sys.argv) > 2 else ''
name, start = None, 0
for i, l in enumerate(src + ['_end:']):
    m = re.match(r'^(_Z\w+):', l) or (l == '_end:' and re.match(r'(_end):', l))
    if m:
        if name and want in name and 'kernel' in name:
            body = src[start:i]
            sl = [j for j, x in enumerate(body) if 'scratch_load' in x]
            ss = [j for j, x in enumerate(body) if 'scratch_store' in x]
            bar = [j for j, x in enumerate(body) if 's_barrier' in x]
            mf = sum('v_mfma' in x for x in body)
            size = next((x for x in body if 'codeLenInByte' in x), '')
            print(f'{name[:110]}\n  lines {len(body)} mfma {mf} scratch loads {len(sl)} stores {len(ss)} {size.strip()}')
            print('  barriers at', bar)
            ranges, cur = [], None
            for j in sorted(sl + ss):
                if cur and j - cur[1] <= 60:
                    cur[1] = j; cur[2] += 1
                else:
                    cur = [j, j, 1]; ranges.append(cur)
            print('  scratch clusters', ' '.join(f'{a}-{b}({c})' for a, b, c in ranges))
        name, start = m.group(1), i
