#!/bin/bash
tag=${1:-t}
out=gpurun_out/r05$tag
mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu > $out/test_train.log 2>&1; echo "train tests rc=$?" | tee -a $out/summary.txt
tail -3 $out/test_train.log | tee -a $out/summary.txt
timeout 900 python -m pytest tests/test_gpu_backbone.py -x -q -m gpu > $out/test_bb.log 2>&1; echo "backbone tests rc=$?" | tee -a $out/summary.txt
tail -3 $out/test_bb.log | tee -a $out/summary.txt
timeout 600 python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_cfg4.json 2> $out/bench_cfg4.err
python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.load(open("$out/bench_cfg4.json"))
    print("cfg4", d["ms_per_step"], d["engine"], "host", d["host_enqueue_ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["kernel"][:90])
except Exception as e:
    print("cfg4 failed", e); print(open("$out/bench_cfg4.err").read()[-2500:])
PY
PROF_EXTRA="--config cfg4" bash tools/prof_windows.sh 64 cfg4
head -22 gpurun_out/prof_cfg4_stats.csv | cut -c1-150 | tee -a $out/summary.txt
for ho in 1 0; do
  IMMTSF_Z_HANDOVER=$ho PROF_EXTRA="--fusion-only" bash tools/prof_windows.sh 4096 fo$ho
  echo "fusion-only 4096 handover $ho" | tee -a $out/summary.txt
  head -16 gpurun_out/prof_fo${ho}_stats.csv | cut -c1-150 | tee -a $out/summary.txt
done
