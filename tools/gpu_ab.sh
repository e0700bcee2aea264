cd $GRAFT_REPO_ROOT
for rep in 1 2; do for g in 0 448 512; do
  timeout 600 python bench.py --no-extras --no-cpu-baseline --no-roofline --gemm-config $((g << 17)) --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ttcn grid $g', d['ms_per_step'])"
done; done
