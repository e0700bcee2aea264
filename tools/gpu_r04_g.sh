cd $GRAFT_REPO_ROOT
IMMTSF_PMC_COMMIT=$1 PMC_TAG=r04a bash tools/pmc_pass_r04.sh 2>&1 | tail -12
mkdir -p profiles; cp gpurun_out/r04a_pmc_traffic.json gpurun_out/r04a_pmc_traffic_w4096.json profiles/
IMMTSF_BENCH_GEMM_TABLE=1 timeout 900 python bench.py --no-extras --no-cpu-baseline 2> gpurun_out/r04g_bench.err | tee gpurun_out/r04g_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step']); print(json.dumps(d['roofline'], indent=1))"
grep "^# gemm" gpurun_out/r04g_bench.err | head -30
