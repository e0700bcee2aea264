set -x
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_fusion.py -x -q 2>&1 | tail -5 > gpurun_out/r04d_fusion_tests.log
cat gpurun_out/r04d_fusion_tests.log
for w in 64 128 256 512 1024; do
  for form in fold chain; do
    timeout 600 python bench.py --no-extras --no-cpu-baseline --no-roofline --t2v-form $form --windows-per-gpu $w --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$form', $w, d['ms_per_step'], d['engine'])" >> gpurun_out/r04d_ab.txt
  done
done
cat gpurun_out/r04d_ab.txt
