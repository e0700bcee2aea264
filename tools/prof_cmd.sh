# usage: bash tools/prof_cmd.sh <tag> <python script and args...>  -> gpurun_out/prof_<tag>_stats.csv
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o r -- python3 "$@" > gpurun_out/prof_$TAG.log 2>&1
f=$(ls gpurun_out/prof_$TAG/*results.db 2>/dev/null | head -1)
python3 tools/rocpd_stats.py $f gpurun_out/prof_${TAG}_stats.csv
rm -f $f
grep -v "rocprofv3\|simple_timer\|tool.cpp\|generateRocpd" gpurun_out/prof_$TAG.log | tail -3
head -12 gpurun_out/prof_${TAG}_stats.csv | cut -d, -f1-4 | cut -c1-150
