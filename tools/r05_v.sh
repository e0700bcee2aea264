#!/bin/bash
tag=${1:-v}
out=gpurun_out/r05$tag
mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_fusion.py -x -q -m gpu -k "t2v or fold or handover or fused_tail or golden or oracle" > $out/test_fusion.log 2>&1; echo "fusion tests rc=$?" | tee -a $out/summary.txt
tail -3 $out/test_fusion.log | tee -a $out/summary.txt
IMMTSF_Z_HANDOVER=1 PROF_EXTRA="--fusion-only" bash tools/prof_windows.sh 4096 fo1
echo "fusion-only 4096 handover 1" | tee -a $out/summary.txt
head -8 gpurun_out/prof_fo1_stats.csv | cut -c1-150 | tee -a $out/summary.txt
grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_fo1.log | tee -a $out/summary.txt
for w in 64 4096; do
  for ho in 1 0 1 0; do
    IMMTSF_Z_HANDOVER=$ho timeout 600 python bench.py --windows-per-gpu $w --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-roofline > $out/b_${w}_$ho.json 2> $out/b_${w}_$ho.err
    python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.load(open("$out/b_${w}_$ho.json")); print("windows $w handover $ho:", d["ms_per_step"], d["engine"])
except Exception as e:
    print("windows $w handover $ho failed", e); print(open("$out/b_${w}_$ho.err").read()[-1500:])
PY
  done
done
