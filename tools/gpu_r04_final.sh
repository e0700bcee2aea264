# round 4 evidence pass (on the GPU box): kernel traces, PMC passes, flag timeline, bench lines.  usage: bash tools/gpu_r04_final.sh <commit>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
C=${1:-unknown}
for W in 64 1024 4096; do
  bash tools/prof_windows.sh $W r04_w$W > /dev/null 2>&1
  cp $O/prof_r04_w${W}_stats.csv $O/r04_w${W}_kernel_stats.csv; cp $O/prof_r04_w${W}_seq.txt $O/r04_w${W}_kernel_sequence.txt
done
IMMTSF_PMC_COMMIT=$C PMC_TAG=r04 bash tools/pmc_pass_r04.sh 2>&1 | tail -4
mkdir -p profiles; cp $O/r04_pmc_traffic.json $O/r04_pmc_traffic_w4096.json profiles/
timeout 300 python3 tools/flag_timeline.py 64 40 > $O/r04_flag_timeline.txt 2>&1; tail -5 $O/r04_flag_timeline.txt
timeout 1500 python3 bench.py > $O/r04_bench_line.json 2> $O/r04_bench.err; tail -2 $O/r04_bench.err
python3 -c "
import json; d=json.load(open('$O/r04_bench_line.json')); r=d['roofline']
print('ms', d['ms_per_step'], 'value', d['value'], d['engine']); print('roofline', r['kernel'][:100], r['avg_launch_us'], r['frac'], r['traffic'], r['algorithmic_bytes'])
print('sweep', [(s['windows_per_gpu'], s['ms_per_step']) for s in d['sweep']]); print('other notes form', (d.get('padded') or d.get('packed'))['ms_per_step'], 'fp32', d['ms_per_step_fp32'], 'dropin', d['dropin']['ms_per_step'], d['dropin']['ms_per_step_nan_guards_sync'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores']); print('step_stats', d['step_stats'])"
timeout 600 python3 bench.py --no-extras --no-cpu-baseline --no-roofline --force-dist > $O/r04_bench_line_force_dist.json 2>/dev/null
python3 -c "import json; d=json.loads(open('$O/r04_bench_line_force_dist.json').read().strip().splitlines()[-1]); print('force-dist', d['ms_per_step'], d['config']['grad_allreduce'])"
