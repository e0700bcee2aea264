# copy the round-2 measurement artefacts (tools/r02_measure.sh, tools/prof_cfg.sh, tools/gemm2_bench.py) from gpurun_out/ into profiles/
cd "$(dirname "$0")/.."
O=gpurun_out; P=profiles
cp $O/r02_prof_stats.csv $P/r02_bench_kernel_stats.csv
cp $O/r02_pmc_traffic.json $P/r02_pmc_traffic.json
cp $O/r02_pmc_fetch/r_counter_collection.csv $P/r02_pmc_fetch_counter_collection.csv
cp $O/r02_pmc_write/r_counter_collection.csv $P/r02_pmc_write_counter_collection.csv
cp $O/r02_bench.json $P/r02_bench_line.json
for c in cfg3 cfg4 cfg5; do [ -f $O/r02_bench_$c.json ] && cp $O/r02_bench_$c.json $P/r02_bench_line_$c.json; [ -f $O/prof_${c}_stats.csv ] && cp $O/prof_${c}_stats.csv $P/r02_${c}_kernel_stats.csv; done
[ -f $O/seq_b.txt ] && cp $O/seq_b.txt $P/r02_step_kernel_sequence.txt
[ -f $O/gemm2_c.log ] && cp $O/gemm2_c.log $P/r02_gemm2_sweep.txt
ls -la $P | grep r02
