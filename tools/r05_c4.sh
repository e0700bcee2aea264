#!/bin/bash
# dev pass for cfg4: TimesNet tests, bench line, kernel stats of the period convolution
out=gpurun_out/r05c4
mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_backbone.py tests/test_gpu_train.py -x -q -m gpu -k "timesnet or TimesNet or cfg4 or period or inception or cfg3_flag" > $out/test.log 2>&1; echo "tests rc=$?" | tee -a $out/summary.txt
tail -3 $out/test.log | cut -c1-220 | tee -a $out/summary.txt
timeout 600 python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_cfg4.json 2> $out/bench_cfg4.err
python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.load(open("$out/bench_cfg4.json"))
    print("cfg4", d["ms_per_step"], d["engine"], "host", d["host_enqueue_ms_per_step"])
except Exception as e:
    print("cfg4 failed", e); print(open("$out/bench_cfg4.err").read()[-3000:])
PY
PROF_EXTRA="--config cfg4" bash tools/prof_windows.sh 64 cfg4
grep "conv_period\|im2col\|period" gpurun_out/prof_cfg4_stats.csv | cut -c1-200 | tee -a $out/summary.txt
