# usage: ab_step.sh "<bench args A>" "<bench args B>" ...   (each run prints ms/step, windows/s, host enqueue ms/step)
for a in "$@"; do
  echo "args: $a"; python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline $a 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], 'host enqueue', d.get('host_enqueue_ms_per_step'))"
done
