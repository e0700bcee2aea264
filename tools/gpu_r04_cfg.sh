# cfg3 / cfg4 / cfg5: bench lines (with cpu_baseline and roofline) + the two PMC passes per configuration, stamped for this build
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
for C in cfg3 cfg4 cfg5; do
  T=${PMC_TAG:-r04}_pmc_$C
  ST=4; [ $C = cfg5 ] && ST=2
  rm -rf $O/${T}_fetch $O/${T}_write
  timeout 1200 rocprofv3 --pmc FETCH_SIZE -d $O/${T}_fetch -o r --output-format csv -- python3 bench.py --config $C --steps $ST --warmup 1 --no-cpu-baseline --no-roofline --no-extras --no-graph > $O/${T}_fetch.log 2>&1
  timeout 1200 rocprofv3 --pmc WRITE_SIZE -d $O/${T}_write -o r --output-format csv -- python3 bench.py --config $C --steps $ST --warmup 1 --no-cpu-baseline --no-roofline --no-extras --no-graph > $O/${T}_write.log 2>&1
  IMMTSF_PMC_WINDOWS=64 IMMTSF_PMC_CONFIG=$C IMMTSF_PMC_COMMIT=$1 python3 tools/pmc_summary.py $O/${T}_fetch $O/${T}_write $O/${PMC_TAG:-r04}_pmc_traffic_$C.json | head -3
  rm -rf $O/${T}_fetch $O/${T}_write
  mkdir -p profiles; cp $O/${PMC_TAG:-r04}_pmc_traffic_$C.json profiles/
  timeout 1500 python3 bench.py --config $C --steps 20 --warmup 5 > $O/${PMC_TAG:-r04}_bench_line_$C.json 2> $O/${PMC_TAG:-r04}_bench_$C.err
  python3 -c "import json; d=json.load(open('$O/${PMC_TAG:-r04}_bench_line_$C.json')); r=d['roofline']; print('$C', d['ms_per_step'], d['engine'], r['kernel'][:90], r['avg_launch_us'], r['frac'], r['traffic'], (d['cpu_baseline'] or {}).get('value'))"
done
timeout 900 python3 bench.py --config cfg5 --fusion-only --steps 20 --warmup 5 --no-cpu-baseline > $O/${PMC_TAG:-r04}_bench_line_cfg5_fusion_only.json 2> $O/${PMC_TAG:-r04}_bench_cfg5fo.err
python3 -c "import json; d=json.load(open('$O/${PMC_TAG:-r04}_bench_line_cfg5_fusion_only.json')); r=d['roofline']; print('cfg5 fusion-only', d['ms_per_step'], d['engine'], r['kernel'][:90], r['avg_launch_us'], r['frac'])"
tail -3 $O/${PMC_TAG:-r04}_bench_cfg5fo.err
