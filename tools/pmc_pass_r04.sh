# round 4: the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, no trace domains -- the program directly after `--`)
# over an eagerly launched bench at the headline batch (64 windows) and at 4096 windows; the summaries are stamped with the content
# hash of the kernel sources (bench.csrc_sha), which is what lets bench.py quote them as `roofline.traffic` for this build only
#   usage (on the GPU box): IMMTSF_PMC_COMMIT=<sha> PMC_TAG=r04 bash tools/pmc_pass_r04.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
for W in 64 4096; do
  T=${PMC_TAG:-r04}_pmc_w$W
  rm -rf $O/${T}_fetch $O/${T}_write
  timeout 900 rocprofv3 --pmc FETCH_SIZE -d $O/${T}_fetch -o r --output-format csv -- python3 bench.py --windows-per-gpu $W --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extras --no-graph > $O/${T}_fetch.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE -d $O/${T}_write -o r --output-format csv -- python3 bench.py --windows-per-gpu $W --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extras --no-graph > $O/${T}_write.log 2>&1
  IMMTSF_PMC_WINDOWS=$W python3 tools/pmc_summary.py $O/${T}_fetch $O/${T}_write $O/${T}.json | head -4
  rm -rf $O/${T}_fetch $O/${T}_write
done
cp $O/${PMC_TAG:-r04}_pmc_w64.json $O/${PMC_TAG:-r04}_pmc_traffic.json
cp $O/${PMC_TAG:-r04}_pmc_w4096.json $O/${PMC_TAG:-r04}_pmc_traffic_w4096.json
