#!/bin/bash
# scratch A/B: FlagStep branch layouts at 64 windows
cd /root/repo
for rep in 1 2 3; do
for kw in '{"backbone_side": false}' '{}' '{"param_tail": 1}' '{"backbone_side": false, "param_tail": 1}'; do
  echo -n "KW=$kw  "
  IMMTSF_BENCH_FLAG_KW="$kw" python3 bench.py --no-cpu-baseline --no-roofline --no-extras --steps 400 --warmup 40 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['engine'], d.get('flag_step_rejected'))"
done
done
