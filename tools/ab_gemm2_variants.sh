for v in 0 9 8 14 26 1; do
  echo "variant $v"; python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --gemm2-variant $v 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
echo "old kernel only"; python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --gemm-config 0x10000 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
