#!/bin/bash
# the secondary configurations: cfg3 / cfg4 bench lines, the speculative-graph test, kernel stats of cfg3 / cfg4
tag=${1:-c}
out=gpurun_out/r05$tag
mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "timesnet_spec" > $out/test_spec.log 2>&1; echo "spec test rc=$?" | tee -a $out/summary.txt
tail -5 $out/test_spec.log | tee -a $out/summary.txt
for c in cfg3 cfg4; do
  timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_$c.json 2> $out/bench_$c.err
  python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.load(open("$out/bench_$c.json"))
    print("$c", d["ms_per_step"], d["engine"], "host", d["host_enqueue_ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["kernel"][:90], d.get("spec_graph"))
except Exception as e:
    print("$c failed", e); print(open("$out/bench_$c.err").read()[-2500:])
PY
done
PROF_EXTRA="--config cfg3" bash tools/prof_windows.sh 64 cfg3
PROF_EXTRA="--config cfg4" bash tools/prof_windows.sh 64 cfg4
head -32 gpurun_out/prof_cfg3_stats.csv | cut -c1-150 | tee -a $out/summary.txt
head -24 gpurun_out/prof_cfg4_stats.csv | cut -c1-150 | tee -a $out/summary.txt
