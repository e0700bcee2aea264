set -x
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_fusion.py -x -q -k "folded_form" 2>&1 | tail -25 > gpurun_out/r04b_fold_tests.log
cat gpurun_out/r04b_fold_tests.log
