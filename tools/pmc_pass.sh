# the two rocprofv3 PMC passes over an eagerly launched bench (FETCH_SIZE, WRITE_SIZE: separate runs) -> gpurun_out/r02_pmc_traffic.json
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -rf $O/r02_pmc_fetch $O/r02_pmc_write
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/r02_pmc_fetch -o r --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-extras --no-graph > $O/r02_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/r02_pmc_write -o r --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-extras --no-graph > $O/r02_pmc_write.log 2>&1
python3 tools/pmc_summary.py $O/r02_pmc_fetch $O/r02_pmc_write $O/r02_pmc_traffic.json | head -16
