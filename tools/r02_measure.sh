# round-2 measurement pass on the GPU box: tests, bench lines, kernel-trace stats, PMC traffic (separate passes)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/r02_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r02_gpu_tests.log
timeout 900 python3 bench.py > $O/r02_bench.json 2> $O/r02_bench.err; echo "bench rc=$?"; cut -c1-600 $O/r02_bench.json
rm -rf $O/r02_prof $O/r02_pmc_fetch $O/r02_pmc_write
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r02_prof -o r -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras > $O/r02_prof.log 2>&1
f=$(ls $O/r02_prof/*results.db 2>/dev/null | head -1)
if [ -n "$f" ]; then python3 tools/rocpd_stats.py $f $O/r02_prof_stats.csv; python3 tools/rocpd_seq.py $f > $O/seq_b.txt 2>&1; rm -f $O/r02_prof/*.db; fi
tail -1 $O/r02_prof.log | cut -c1-300
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/r02_pmc_fetch -o r --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-extras --no-graph > $O/r02_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/r02_pmc_write -o r --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-extras --no-graph > $O/r02_pmc_write.log 2>&1
ls $O/r02_pmc_fetch $O/r02_pmc_write
python3 tools/pmc_summary.py $O/r02_pmc_fetch $O/r02_pmc_write $O/r02_pmc_traffic.json | head -14
for c in cfg3 cfg5; do timeout 900 python3 bench.py --config $c --no-cpu-baseline > $O/r02_bench_$c.json 2> $O/r02_bench_$c.err; echo "$c rc=$?"; cut -c1-400 $O/r02_bench_$c.json; done
