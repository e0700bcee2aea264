cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python3 bench.py --config cfg5 --fusion-only --steps 20 --warmup 5 --no-cpu-baseline > $O/r04_bench_line_cfg5_fusion_only.json 2> $O/r04_bench_cfg5fo.err
python3 -c "import json; d=json.load(open('$O/r04_bench_line_cfg5_fusion_only.json')); r=d['roofline']; print('cfg5 fusion-only', d['ms_per_step'], d['engine'], r['kernel'][:90], r['avg_launch_us'], r['frac'])"
tail -3 $O/r04_bench_cfg5fo.err
