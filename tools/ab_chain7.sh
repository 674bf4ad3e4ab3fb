for r in 1 2 3; do
for v in 1 0; do
  F8_FUSE_CHAIN7=$v F8_BENCH_LEAN=1 timeout 300 python bench.py --steps 150 --warmup 20 --per-layer --no-cpu-baseline > /tmp/b_$v.json 2> /tmp/p_$v.txt
  echo "== chain7=$v rep $r: $(python -c "import json; d=json.load(open('/tmp/b_$v.json')); print(d['value'], 'unpipelined', d.get('value_unpipelined'), 'sum_kernel_ms', d['whole_net']['sum_kernel_ms'])")"
  grep -E "^ +[0-9]+ .*(stage_3|avgpool|stage_chain_x2)" /tmp/p_$v.txt | sed -E 's/^ +[0-9]+ ([^ ]+) +([0-9.]+) us.*/     \2 us  \1/' | cut -c1-100
done
done
