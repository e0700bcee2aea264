cd $GRAFT_REPO_ROOT
for C in cfg3 cfg4; do
PROF_EXTRA="--config $C" bash tools/prof_windows.sh 64 r04_$C > /dev/null 2>&1
cp gpurun_out/prof_r04_${C}_stats.csv gpurun_out/r04_${C}_kernel_stats.csv
head -4 gpurun_out/r04_${C}_kernel_stats.csv | cut -c1-150
done
PROF_EXTRA="--config cfg5 --fusion-only" bash tools/prof_windows.sh 64 r04_cfg5fo > /dev/null 2>&1
cp gpurun_out/prof_r04_cfg5fo_stats.csv gpurun_out/r04_cfg5_fusion_only_kernel_stats.csv
head -3 gpurun_out/r04_cfg5_fusion_only_kernel_stats.csv | cut -c1-150
