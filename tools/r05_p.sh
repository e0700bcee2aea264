#!/bin/bash
tag=${1:-p}
out=gpurun_out/r05$tag
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_backbone.py tests/test_gpu_train.py -x -q -m gpu -k "TimesNet or timesnet or cfg4" > $out/test.log 2>&1; echo "tests rc=$?" | tee -a $out/summary.txt
tail -6 $out/test.log | tee -a $out/summary.txt
timeout 900 python -m pytest tests/test_gpu_fusion.py -x -q -m gpu -k "cfg4" >> $out/test.log 2>&1; echo "cfg4 fusion tests rc=$?" | tee -a $out/summary.txt
timeout 600 python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_cfg4.json 2> $out/bench_cfg4.err
python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.load(open("$out/bench_cfg4.json"))
    print("cfg4", d["ms_per_step"], d["engine"], "host", d["host_enqueue_ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["kernel"][:90])
except Exception as e:
    print("cfg4 failed", e); print(open("$out/bench_cfg4.err").read()[-2500:])
PY
PROF_EXTRA="--config cfg4" bash tools/prof_windows.sh 64 cfg4
head -24 gpurun_out/prof_cfg4_stats.csv | cut -c1-150 | tee -a $out/summary.txt
