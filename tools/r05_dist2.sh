#!/bin/bash
tag=${1:-e}
out=gpurun_out/r05$tag
mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {  # name, env assignments..., uses default FlagStep kw
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --force-dist --no-extras --no-cpu-baseline --no-roofline > $out/fd_$name.json 2> $out/fd_$name.err
  python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.load(open("$out/fd_$name.json"))
    print("$name", d["ms_per_step"], d["flag_step_rejected"], "host", d["host_enqueue_ms_per_step"], d["config"]["grad_allreduce"][90:200])
except Exception as e:
    print("$name failed", e)
PY
}
run default X=1
run prio0 IMMTSF_COMM_PRIO=0
run sidestream IMMTSF_BENCH_STREAM=1
run hwq8 GPU_MAX_HW_QUEUES=8
run hwq8_prio0 GPU_MAX_HW_QUEUES=8 IMMTSF_COMM_PRIO=0
run hwq2 GPU_MAX_HW_QUEUES=2
run fp32wire X=1 IMMTSF_BENCH_ARGS=1
