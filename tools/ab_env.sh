# usage: ab_env.sh "VAR=val VAR2=val" ...   one bench run per argument, prints ms/step
for e in "$@"; do
  echo "env: $e"; env $e python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
