#!/bin/bash
cd /root/repo
python3 -m pytest tests -x -q -m gpu 2>&1 | tail -2
for w in 64 1024 4096; do
echo -n "windows $w: "
python3 bench.py --no-cpu-baseline --no-roofline --no-extras --steps 60 --warmup 10 --windows-per-gpu $w 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['engine'])"
done
