#!/bin/bash
# sample sclk / power while a bench runs
for v in 1 0; do
  ( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > /tmp/clk_$v.txt &
  SP=$!
  F8_STEM_ROWS=$v timeout 300 python bench.py --steps 3000 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rows=$v', d['value'], d['value_unpipelined'])"
  wait $SP
  echo "--- rows=$v"; sort /tmp/clk_$v.txt | uniq -c | sort -rn | head -8
done
