cd $GRAFT_REPO_ROOT
bash tools/prof_windows.sh 4096 r04a_w4096
bash tools/prof_windows.sh 64 r04a_w64
head -40 gpurun_out/prof_r04a_w4096_stats.csv
