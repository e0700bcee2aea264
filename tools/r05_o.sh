#!/bin/bash
tag=${1:-o}
out=gpurun_out/r05$tag
mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "two_rank_flag or timesnet_spec" > $out/test_train.log 2>&1; echo "tests rc=$?" | tee -a $out/summary.txt
tail -4 $out/test_train.log | tee -a $out/summary.txt
timeout 300 python bench.py --steps 40 --warmup 10 --force-dist --no-extras --no-cpu-baseline --no-roofline > $out/fd.json 2> $out/fd.err
timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline --no-roofline > $out/single.json 2> $out/single.err
python - <<PY | tee -a $out/summary.txt
import json
for f in ("fd","single"):
    try:
        d=json.load(open("$out/%s.json"%f)); print(f, d["ms_per_step"], d["flag_step_rejected"], d["host_enqueue_ms_per_step"], d["config"]["grad_allreduce"][:400])
    except Exception as e:
        print(f, "failed", e)
PY
DIST=1 timeout 300 python tools/flag_timeline.py 64 6 > $out/flag_timeline_64_dist.txt 2>&1
tail -40 $out/flag_timeline_64_dist.txt | tee -a $out/summary.txt
timeout 600 python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_cfg4.json 2> $out/bench_cfg4.err
python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.load(open("$out/bench_cfg4.json"))
    print("cfg4", d["ms_per_step"], d["engine"], "host", d["host_enqueue_ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["kernel"][:90], d.get("spec_graph"))
except Exception as e:
    print("cfg4 failed", e); print(open("$out/bench_cfg4.err").read()[-2500:])
PY
