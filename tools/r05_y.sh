#!/bin/bash
tag=${1:-y}
out=gpurun_out/r05$tag
mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_fusion.py -x -q -m gpu -k "gr_add or gr_" > $out/test_gr.log 2>&1; echo "gr tests rc=$?" | tee -a $out/summary.txt
tail -12 $out/test_gr.log | cut -c1-220 | tee -a $out/summary.txt
timeout 600 python bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_cfg3.json 2> $out/bench_cfg3.err
python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.load(open("$out/bench_cfg3.json"))
    print("cfg3", d["ms_per_step"], d["engine"], "rejected", d.get("flag_step_rejected"), "host", d["host_enqueue_ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["kernel"][:90])
except Exception as e:
    print("cfg3 failed", e); print(open("$out/bench_cfg3.err").read()[-3000:])
PY
tail -5 $out/bench_cfg3.err | cut -c1-300 | tee -a $out/summary.txt
PROF_EXTRA="--config cfg3" bash tools/prof_windows.sh 64 cfg3
head -14 gpurun_out/prof_cfg3_stats.csv | cut -c1-150 | tee -a $out/summary.txt
