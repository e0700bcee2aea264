#!/bin/bash
tag=${1:-z}
out=gpurun_out/r05$tag
mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests -x -q -m gpu > $out/test.log 2>&1; echo "tests rc=$?" | tee -a $out/summary.txt
tail -4 $out/test.log | cut -c1-220 | tee -a $out/summary.txt
for c in cfg3 cfg4; do
timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_$c.json 2> $out/bench_$c.err
python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.load(open("$out/bench_$c.json"))
    print("$c", d["ms_per_step"], d["engine"], "host", d["host_enqueue_ms_per_step"])
except Exception as e:
    print("$c failed", e); print(open("$out/bench_$c.err").read()[-3000:])
PY
done
PROF_EXTRA="--config cfg3" bash tools/prof_windows.sh 64 cfg3
grep "gr_train\|instance_norm" gpurun_out/prof_cfg3_stats.csv | cut -c1-200 | tee -a $out/summary.txt
