# kernel-trace stats of bench.py --config $1 (eager or graph as the config runs): gpurun_out/prof_$1_stats.csv
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
c=$1; st=${2:-10}; wu=${3:-3}
rm -rf gpurun_out/prof_$c
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$c -o r -- python3 bench.py --config $c --steps $st --warmup $wu --no-cpu-baseline --no-roofline --no-extras > gpurun_out/prof_$c.log 2>&1
f=$(ls gpurun_out/prof_$c/*results.db 2>/dev/null | head -1)
if [ -n "$f" ]; then python3 tools/rocpd_stats.py $f gpurun_out/prof_${c}_stats.csv; rm -f $f; fi
tail -1 gpurun_out/prof_$c.log | cut -c1-200
