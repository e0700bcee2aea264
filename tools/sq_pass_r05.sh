# round 5: SQ counter passes (MFMA busy cycles / MFMA op counts / wave-cycle breakdown) over an eagerly launched bench at 64 and 4096
# windows -- one rocprofv3 --pmc run per counter group, the program directly after `--`, no trace domains.  The summaries carry the
# content hash of the kernel sources (bench.csrc_sha) like the traffic profiles.
#   usage (on the GPU box): IMMTSF_PMC_COMMIT=<sha> PMC_TAG=r05 bash tools/sq_pass_r05.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"
B="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
for W in 64 4096; do
  T=${PMC_TAG:-r05}_sq_w$W
  rm -rf $O/${T}_a $O/${T}_b
  timeout 900 rocprofv3 --pmc $A -d $O/${T}_a -o r --output-format csv -- python3 bench.py --windows-per-gpu $W --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extras --no-graph > $O/${T}_a.log 2>&1
  timeout 900 rocprofv3 --pmc $B -d $O/${T}_b -o r --output-format csv -- python3 bench.py --windows-per-gpu $W --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extras --no-graph > $O/${T}_b.log 2>&1
  IMMTSF_PMC_WINDOWS=$W python3 tools/sq_summary.py $O/${T}_a $O/${T}_b $O/${T}.json | head -12
  rm -rf $O/${T}_a $O/${T}_b
done
