# round 5 evidence pass (on the GPU box): kernel traces, PMC traffic + SQ counter passes, flag timelines, bench lines.
#   usage: bash tools/r05_final.sh <commit> [part]      part: a = traces + counters + timelines, b = bench lines (default: both)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out
C=${1:-unknown}
PART=${2:-ab}
mkdir -p $O/r05final
S=$O/r05final/summary.txt
if [[ $PART == *a* ]]; then
  timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/r05final/test_gpu.log 2>&1; echo "gpu tests rc=$?" | tee -a $S; tail -2 $O/r05final/test_gpu.log | tee -a $S
  timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $S
  for W in 64 1024 4096; do
    bash tools/prof_windows.sh $W r05_w$W > /dev/null 2>&1
    cp $O/prof_r05_w${W}_stats.csv $O/r05final/r05_w${W}_kernel_stats.csv; cp $O/prof_r05_w${W}_seq.txt $O/r05final/r05_w${W}_kernel_sequence.txt
  done
  for c in cfg3 cfg4; do
    PROF_EXTRA="--config $c" bash tools/prof_windows.sh 64 r05_$c > /dev/null 2>&1
    cp $O/prof_r05_${c}_stats.csv $O/r05final/r05_${c}_kernel_stats.csv
  done
  IMMTSF_PMC_COMMIT=$C PMC_TAG=r05 bash tools/pmc_pass_r04.sh 2>&1 | tail -4 | tee -a $S
  cp $O/r05_pmc_traffic.json $O/r05_pmc_traffic_w4096.json $O/r05final/
  IMMTSF_PMC_COMMIT=$C PMC_TAG=r05 bash tools/sq_pass_r05.sh 2>&1 | tail -30 | tee -a $S
  cp $O/r05_sq_w64.json $O/r05_sq_w4096.json $O/r05final/
  # bench quotes traffic / mfma_busy only from profiles/ of this build: put them there before the bench lines are taken
  mkdir -p profiles; cp $O/r05_pmc_traffic.json $O/r05_pmc_traffic_w4096.json $O/r05_sq_w64.json $O/r05_sq_w4096.json profiles/
  timeout 300 python3 tools/flag_timeline.py 64 40 > $O/r05final/r05_flag_timeline.txt 2>&1; tail -3 $O/r05final/r05_flag_timeline.txt | tee -a $S
  DIST=1 timeout 300 python3 tools/flag_timeline.py 64 12 > $O/r05final/r05_flag_timeline_dist.txt 2>&1; tail -3 $O/r05final/r05_flag_timeline_dist.txt | tee -a $S
  timeout 300 python3 tools/flag_timeline.py 1024 8 > $O/r05final/r05_flag_timeline_w1024.txt 2>&1; tail -2 $O/r05final/r05_flag_timeline_w1024.txt | tee -a $S
  timeout 300 python3 tools/flag_timeline.py 4096 8 > $O/r05final/r05_flag_timeline_w4096.txt 2>&1; tail -2 $O/r05final/r05_flag_timeline_w4096.txt | tee -a $S
fi
if [[ $PART == *b* ]]; then
  mkdir -p profiles; cp $O/r05final/r05_pmc_traffic*.json $O/r05final/r05_sq_w*.json profiles/ 2>/dev/null
  timeout 1800 python3 bench.py > $O/r05final/r05_bench_line.json 2> $O/r05final/r05_bench.err; tail -2 $O/r05final/r05_bench.err
  python3 -c "
import json; d=json.load(open('$O/r05final/r05_bench_line.json')); r=d['roofline']
print('ms', d['ms_per_step'], 'value', d['value'], d['engine']); print('roofline', r['kernel'][:100], r['avg_launch_us'], r['frac'], r['traffic'], r['algorithmic_bytes'], 'mfma_busy', r.get('mfma_busy'))
print('hbm', [(k.get('kernel','')[:30], k.get('us'), k.get('frac')) for k in d['roofline_hbm']['kernels']])
print('sweep', [(s['windows_per_gpu'], s['ms_per_step']) for s in d['sweep']]); print('other notes form', (d.get('padded') or d.get('packed'))['ms_per_step'], 'fp32', d['ms_per_step_fp32'], 'dropin', d['dropin']['ms_per_step'], d['dropin']['ms_per_step_nan_guards_sync'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores']); print('step_stats', d['step_stats'])" | tee -a $S
  timeout 600 python3 bench.py --no-extras --no-cpu-baseline --no-roofline --force-dist > $O/r05final/r05_bench_line_force_dist.json 2>/dev/null
  timeout 600 python3 bench.py --no-extras --no-cpu-baseline --no-roofline > $O/r05final/r05_bench_line_single.json 2>/dev/null
  python3 -c "
import json
for f in ('force_dist','single'):
    d=json.loads(open('$O/r05final/r05_bench_line_%s.json'%f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['config']['grad_allreduce'][:400])" | tee -a $S
  for c in cfg3 cfg4 cfg5; do
    timeout 900 python3 bench.py --config $c --steps 20 --warmup 5 > $O/r05final/r05_bench_line_$c.json 2> $O/r05final/r05_bench_$c.err
    python3 -c "
import json; d=json.load(open('$O/r05final/r05_bench_line_$c.json')); r=d['roofline']; print('$c', d['ms_per_step'], d['engine'], 'host', d['host_enqueue_ms_per_step'], 'roofline', r['frac'], r['avg_launch_us'], r['kernel'][:90], 'cpu', (d.get('cpu_baseline') or {}).get('value'))" | tee -a $S
  done
  timeout 900 python3 bench.py --config cfg5 --fusion-only --steps 10 --warmup 3 --no-cpu-baseline > $O/r05final/r05_bench_line_cfg5_fusion_only.json 2> $O/r05final/r05_bench_cfg5_fo.err
  python3 -c "
import json; d=json.load(open('$O/r05final/r05_bench_line_cfg5_fusion_only.json')); print('cfg5 fusion-only', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'][:90])" | tee -a $S
fi
