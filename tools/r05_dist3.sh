#!/bin/bash
tag=${1:-f}
out=gpurun_out/r05$tag
mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {  # name, kw json, extra env
  name=$1; kw=$2; shift; shift
  env IMMTSF_BENCH_FLAG_KW="$kw" "$@" timeout 300 python bench.py --steps 40 --warmup 10 --force-dist --no-extras --no-cpu-baseline --no-roofline > $out/fd_$name.json 2> $out/fd_$name.err
  python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.load(open("$out/fd_$name.json"))
    print("$name", d["ms_per_step"], d["flag_step_rejected"], "host", d["host_enqueue_ms_per_step"], d["config"]["grad_allreduce"][90:230])
except Exception as e:
    print("$name failed", e)
    print(open("$out/fd_$name.err").read()[-1500:])
PY
}
timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline --no-roofline > $out/single.json 2> $out/single.err
python -c "import json; d=json.load(open('$out/single.json')); print('single', d['ms_per_step'], d['host_enqueue_ms_per_step'])" | tee -a $out/summary.txt
run default '{}' X=1
run seeds '{"seed_reduce": true}' X=1
run nowgt '{"ttf_wgrad_tail": false}' X=1
run hwq5 '{}' GPU_MAX_HW_QUEUES=5
run hwq6 '{}' GPU_MAX_HW_QUEUES=6
run nomerge '{"merge_adjacent": false}' X=1
DIST=1 timeout 300 python tools/flag_timeline.py 64 6 > $out/flag_timeline_64_dist.txt 2>&1
tail -50 $out/flag_timeline_64_dist.txt | tee -a $out/summary.txt
timeout 300 python tools/flag_timeline.py 64 6 > $out/flag_timeline_64.txt 2>&1
tail -25 $out/flag_timeline_64.txt | tee -a $out/summary.txt
for v in; do
timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline --no-roofline --gemm2-variant $v > $out/single_v$v.json 2> $out/single_v$v.err
python -c "import json; d=json.load(open('$out/single_v$v.json')); print('single nn-dyn variant $v', d['ms_per_step'], d['host_enqueue_ms_per_step'])" | tee -a $out/summary.txt
done
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu > $out/test_train.log 2>&1; echo "train tests rc=$?" | tee -a $out/summary.txt
tail -3 $out/test_train.log | tee -a $out/summary.txt
timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline --no-roofline > $out/single_attn2.json 2> $out/single_attn2.err
python -c "import json; d=json.load(open('$out/single_attn2.json')); print('single again', d['ms_per_step'])" | tee -a $out/summary.txt
IMMTSF_BENCH_FLAG_KW='{"ttf_wgrad_tail": false}' timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline --no-roofline > $out/single_nowgt.json 2> $out/single_nowgt.err
python -c "import json; d=json.load(open('$out/single_nowgt.json')); print('single, weight gradients on the text branch', d['ms_per_step'])" | tee -a $out/summary.txt
timeout 600 python -m pytest tests/test_gpu_fusion.py -x -q -m gpu -k "t2v or index or ragged or golden" > $out/test_fusion.log 2>&1; echo "fusion tests rc=$?" | tee -a $out/summary.txt
tail -3 $out/test_fusion.log | tee -a $out/summary.txt
for q in 5 6; do
GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline --no-roofline > $out/single_hwq$q.json 2> $out/single_hwq$q.err
python -c "import json; d=json.load(open('$out/single_hwq$q.json')); print('single hwq$q', d['ms_per_step'], d['engine'])" | tee -a $out/summary.txt
done
