set -x
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -15 > gpurun_out/r04a_train_tests.log
timeout 600 python -m pytest tests/test_gpu_fusion.py -x -q -k "head_loss_backward" 2>&1 | tail -5 > gpurun_out/r04a_fusion_tests.log
timeout 600 python bench.py --no-extras --no-cpu-baseline --no-roofline > gpurun_out/r04a_bench_n1.json 2> gpurun_out/r04a_bench_n1.err
timeout 600 python bench.py --no-extras --no-cpu-baseline --no-roofline --force-dist > gpurun_out/r04a_bench_forcedist.json 2> gpurun_out/r04a_bench_forcedist.err
IMMTSF_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --no-extras --no-cpu-baseline --no-roofline > gpurun_out/r04a_bench_share2.json 2> gpurun_out/r04a_bench_share2.err
tail -3 gpurun_out/r04a_*.log; cat gpurun_out/r04a_bench_*.json; tail -5 gpurun_out/r04a_bench_share2.err
