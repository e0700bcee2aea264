#!/bin/bash
# data-parallel step on a 1-rank RCCL group: A/B of the FlagStep options, then the flag timeline of the default
tag=${1:-d}
out=gpurun_out/r05$tag
mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {  # name, kw json
  IMMTSF_BENCH_FLAG_KW="$2" timeout 300 python bench.py --steps 40 --warmup 10 --force-dist --no-extras --no-cpu-baseline --no-roofline > $out/fd_$1.json 2> $out/fd_$1.err
  python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.load(open("$out/fd_$1.json"))
    print("$1", d["ms_per_step"], d["engine"], d["flag_step_rejected"], "host", d["host_enqueue_ms_per_step"], d["config"]["grad_allreduce"][:260])
except Exception as e:
    print("$1 failed", e)
PY
}
timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline --no-roofline > $out/single.json 2> $out/single.err
python -c "import json; d=json.load(open('$out/single.json')); print('single', d['ms_per_step'], d['host_enqueue_ms_per_step'])" | tee -a $out/summary.txt
run default '{}'
run no_wgrad_tail '{"ttf_wgrad_tail": false}'
run no_seed '{"seed_reduce": false}'
run neither '{"ttf_wgrad_tail": false, "seed_reduce": false}'
run no_merge '{"merge_adjacent": false}'
DIST=1 timeout 300 python tools/flag_timeline.py 64 6 > $out/flag_timeline_64_dist.txt 2>&1
timeout 300 python tools/flag_timeline.py 64 6 > $out/flag_timeline_64.txt 2>&1
tail -45 $out/flag_timeline_64_dist.txt | tee -a $out/summary.txt
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "two_rank_flag or phases or cfg2_step" > $out/test_train.log 2>&1; echo "train tests rc=$?" | tee -a $out/summary.txt
tail -3 $out/test_train.log | tee -a $out/summary.txt
