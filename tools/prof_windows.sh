cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_w
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_w -o r -- python3 bench.py --windows-per-gpu 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > gpurun_out/prof_w.log 2>&1
f=$(ls gpurun_out/prof_w/*results.db 2>/dev/null | head -1)
python3 tools/rocpd_stats.py $f gpurun_out/prof_w_stats.csv; rm -f $f
tail -1 gpurun_out/prof_w.log | cut -c1-200
