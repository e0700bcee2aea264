#!/bin/bash
cd /root/repo
for w in 512 1024 4096; do
for kw in '{"sched_gate": false}' '{}'; do
echo -n "windows $w flags KW=$kw: "
IMMTSF_BENCH_FLAG_KW="$kw" python3 bench.py --no-cpu-baseline --no-roofline --no-extras --steps 60 --warmup 10 --windows-per-gpu $w --flags 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['engine'], d['flag_step_rejected'])"
done
done
echo -n "windows 4096 graphed: "; python3 bench.py --no-cpu-baseline --no-roofline --no-extras --steps 60 --warmup 10 --windows-per-gpu 4096 --no-flags 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['engine'])"
python3 -m pytest tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -2
