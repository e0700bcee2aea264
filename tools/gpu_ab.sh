cd $GRAFT_REPO_ROOT
for w in 64 256 1024 4096; do
  for ft in on off; do
    timeout 600 python bench.py --no-extras --no-cpu-baseline --no-roofline --fuse-tail $ft --windows-per-gpu $w --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fuse_tail $ft', $w, d['ms_per_step'], d['engine'])"
  done
done
