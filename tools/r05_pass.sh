#!/bin/bash
# one GPU pass: tests (train file first), headline bench, the 1-rank data-parallel step, flag timelines
# usage: tools/r05_pass.sh <tag> [quick]
tag=${1:-a}
out=gpurun_out/r05$tag
mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu > $out/test_train.log 2>&1; echo "train tests rc=$?" | tee -a $out/summary.txt
tail -3 $out/test_train.log | tee -a $out/summary.txt
if [ "$2" != "quick" ]; then
  timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_train.py > $out/test_all.log 2>&1; echo "other tests rc=$?" | tee -a $out/summary.txt
  tail -3 $out/test_all.log | tee -a $out/summary.txt
fi
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" | tee -a $out/summary.txt
python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.load(open("$out/bench.json"))
    print("ms_per_step", d["ms_per_step"], "engine", d["engine"], "stats", d.get("step_stats"), "host", d["host_enqueue_ms_per_step"])
    print("sweep", [(x["windows_per_gpu"], x["ms_per_step"]) for x in d.get("sweep", [])])
    print("dropin", d.get("dropin", {}).get("ms_per_step"), "fp32", d.get("ms_per_step_fp32"), "padded", d.get("padded", {}).get("ms_per_step"))
    print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["kernel"][:80])
except Exception as e:
    print("bench parse failed", e)
PY
timeout 300 python bench.py --steps 20 --warmup 5 --force-dist --no-extras --no-cpu-baseline --no-roofline > $out/bench_force_dist.json 2> $out/bench_force_dist.err; echo "force-dist rc=$?" | tee -a $out/summary.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-roofline > $out/bench_single.json 2> $out/bench_single.err
python - <<PY | tee -a $out/summary.txt
import json
for f in ("bench_force_dist","bench_single"):
    try:
        d=json.load(open("$out/%s.json"%f))
        print(f, d["ms_per_step"], d["engine"], d["flag_step_rejected"], d["host_enqueue_ms_per_step"], d["config"]["grad_allreduce"][:300])
    except Exception as e:
        print(f, "failed", e)
PY
timeout 300 python tools/flag_timeline.py 64 8 > $out/flag_timeline_64.txt 2>&1
DIST=1 timeout 300 python tools/flag_timeline.py 64 8 > $out/flag_timeline_64_dist.txt 2>&1
tail -30 $out/flag_timeline_64.txt | tee -a $out/summary.txt
tail -40 $out/flag_timeline_64_dist.txt | tee -a $out/summary.txt
