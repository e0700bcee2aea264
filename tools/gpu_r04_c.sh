set -x
cd $GRAFT_REPO_ROOT
for form in auto chain; do
  for w in 64 1024 4096; do
    timeout 600 python bench.py --no-extras --no-cpu-baseline --no-roofline --t2v-form $form --windows-per-gpu $w --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$form', $w, d['ms_per_step'], d['engine'])" >> gpurun_out/r04c_ab.txt
  done
done
cat gpurun_out/r04c_ab.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04c_gpu_tests.log
cat gpurun_out/r04c_gpu_tests.log
