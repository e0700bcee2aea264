cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_b
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b -o r -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras > gpurun_out/prof_b.log 2>&1
ls gpurun_out/prof_b | head
f=$(ls gpurun_out/prof_b/*results.db 2>/dev/null | head -1)
if [ -n "$f" ]; then python3 tools/rocpd_stats.py $f gpurun_out/prof_b_stats.csv; fi
tail -2 gpurun_out/prof_b.log | cut -c1-300
