# usage: bash tools/prof_windows.sh [windows=1024] [tag=w]   -> gpurun_out/prof_<tag>_stats.csv (per-kernel totals) and
# gpurun_out/prof_<tag>_seq.txt (ordered kernel sequence of the last step: start offset, duration, queue, name, grid)
W=${1:-1024}; TAG=${2:-w}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o r -- python3 bench.py --windows-per-gpu $W --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras $PROF_EXTRA > gpurun_out/prof_$TAG.log 2>&1
f=$(ls gpurun_out/prof_$TAG/*results.db 2>/dev/null | head -1)
python3 tools/rocpd_stats.py $f gpurun_out/prof_${TAG}_stats.csv
python3 tools/rocpd_seq.py $f > gpurun_out/prof_${TAG}_seq.txt
rm -f $f
tail -1 gpurun_out/prof_$TAG.log | cut -c1-200
