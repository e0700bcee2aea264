#!/bin/bash
tag=${1:-x}
out=gpurun_out/r05$tag
mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -x -q -m gpu > $out/test_all.log 2>&1; echo "all gpu tests rc=$?" | tee -a $out/summary.txt
tail -4 $out/test_all.log | tee -a $out/summary.txt
for c in cfg3 cfg4; do
  timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_$c.json 2> $out/bench_$c.err
  python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.load(open("$out/bench_$c.json"))
    print("$c", d["ms_per_step"], d["engine"], "host", d["host_enqueue_ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["kernel"][:90])
except Exception as e:
    print("$c failed", e); print(open("$out/bench_$c.err").read()[-2500:])
PY
done
for kw in '{}' '{"sched_gate": false}'; do
  IMMTSF_BENCH_FLAG_KW="$kw" timeout 600 python bench.py --windows-per-gpu 4096 --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-roofline > $out/b_4096.json 2> $out/b_4096.err
  python - <<PY | tee -a $out/summary.txt
import json
d=json.load(open("$out/b_4096.json")); print("windows 4096 kw $kw:", d["ms_per_step"], d["engine"])
PY
done
bash tools/prof_windows.sh 64 w64
cp gpurun_out/prof_w64_seq.txt $out/
PROF_EXTRA="--config cfg3" bash tools/prof_windows.sh 64 cfg3
head -12 gpurun_out/prof_cfg3_stats.csv | cut -c1-150 | tee -a $out/summary.txt
