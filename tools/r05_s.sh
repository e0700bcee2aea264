#!/bin/bash
# the Z hand-over pass: fusion + train tests, 64 / 4096 windows with and without the hand-over, kernel stats at 4096, cfg4
tag=${1:-s}
out=gpurun_out/r05$tag
mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_fusion.py -x -q -m gpu > $out/test_fusion.log 2>&1; echo "fusion tests rc=$?" | tee -a $out/summary.txt
tail -4 $out/test_fusion.log | tee -a $out/summary.txt
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu > $out/test_train.log 2>&1; echo "train tests rc=$?" | tee -a $out/summary.txt
tail -3 $out/test_train.log | tee -a $out/summary.txt
for w in 64 4096; do
  for ho in 1 0; do
    IMMTSF_Z_HANDOVER=$ho timeout 600 python bench.py --windows-per-gpu $w --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-roofline > $out/b_${w}_$ho.json 2> $out/b_${w}_$ho.err
    python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.load(open("$out/b_${w}_$ho.json")); print("windows $w handover $ho:", d["ms_per_step"], d["engine"], d.get("step_stats"))
except Exception as e:
    print("windows $w handover $ho failed", e); print(open("$out/b_${w}_$ho.err").read()[-1500:])
PY
  done
done
bash tools/prof_windows.sh 4096 w4096
head -40 gpurun_out/prof_w4096_stats.csv | cut -c1-170 | tee -a $out/summary.txt
timeout 600 python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_cfg4.json 2> $out/bench_cfg4.err
python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.load(open("$out/bench_cfg4.json"))
    print("cfg4", d["ms_per_step"], d["engine"], "host", d["host_enqueue_ms_per_step"])
except Exception as e:
    print("cfg4 failed", e); print(open("$out/bench_cfg4.err").read()[-1500:])
PY
