# round 5 evidence, second half on a fresh box: the device-clock flag timelines, then the bench lines (tools/r05_final.sh part b)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out
mkdir -p $O/r05final
S=$O/r05final/summary_c.txt
timeout 300 python3 tools/flag_timeline.py 64 40 > $O/r05final/r05_flag_timeline.txt 2>&1; tail -3 $O/r05final/r05_flag_timeline.txt | tee -a $S
DIST=1 timeout 300 python3 tools/flag_timeline.py 64 12 > $O/r05final/r05_flag_timeline_dist.txt 2>&1; tail -3 $O/r05final/r05_flag_timeline_dist.txt | tee -a $S
timeout 300 python3 tools/flag_timeline.py 1024 8 > $O/r05final/r05_flag_timeline_w1024.txt 2>&1; tail -2 $O/r05final/r05_flag_timeline_w1024.txt | tee -a $S
timeout 300 python3 tools/flag_timeline.py 4096 8 > $O/r05final/r05_flag_timeline_w4096.txt 2>&1; tail -2 $O/r05final/r05_flag_timeline_w4096.txt | tee -a $S
CFG=cfg4 timeout 300 python3 tools/flag_timeline.py 64 8 > $O/r05final/r05_flag_timeline_cfg4.txt 2>&1; tail -2 $O/r05final/r05_flag_timeline_cfg4.txt | tee -a $S
bash tools/r05_final.sh ${1:-unknown} b
