#!/bin/bash
tag=${1:-w}
out=gpurun_out/r05$tag
mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_fusion.py -x -q -m gpu -k "fold or handover" > $out/test_fusion.log 2>&1; echo "fusion tests rc=$?" | tee -a $out/summary.txt
tail -3 $out/test_fusion.log | tee -a $out/summary.txt
for w in 4096 1024; do
  timeout 300 python tools/flag_timeline.py $w 4 > $out/flag_timeline_$w.txt 2>&1
  tail -28 $out/flag_timeline_$w.txt | tee -a $out/summary.txt
done
timeout 600 python bench.py --windows-per-gpu 4096 --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-roofline > $out/b_4096.json 2> $out/b_4096.err
python - <<PY | tee -a $out/summary.txt
import json
d=json.load(open("$out/b_4096.json")); print("windows 4096:", d["ms_per_step"], d["engine"])
PY
bash tools/prof_windows.sh 4096 w4096
head -30 gpurun_out/prof_w4096_stats.csv | cut -c1-150 | tee -a $out/summary.txt
