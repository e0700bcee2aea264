#!/bin/bash
# the secondary configurations: cfg3 / cfg4 bench lines (+ kernel stats of cfg3)
tag=${1:-c}
out=gpurun_out/r05$tag
mkdir -p $out
for c in cfg3 cfg4; do
  timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_$c.json 2> $out/bench_$c.err
  python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.load(open("$out/bench_$c.json"))
    print("$c", d["ms_per_step"], d["engine"], "host", d["host_enqueue_ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["kernel"][:90])
except Exception as e:
    print("$c failed", e); print(open("$out/bench_$c.err").read()[-1500:])
PY
done
