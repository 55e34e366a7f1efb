#!/bin/bash
# round-3 final GPU session: whole GPU suite, smoke, default bench, c3 / c4 / c5, rocprofv3 kernel stats + PMC passes of the bench command
R=$PWD; O=$R/gpurun_out/r3z; mkdir -p $O
python -m pytest tests -m gpu -q --maxfail=30 2>&1 | tail -40 > $O/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
for c in c3 c4 c5; do python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -- python3 $R/bench.py --config c5 --no-cpu-baseline --no-extras --steps 4 --warmup 2 > $O/bench_c5_under_rocprof.json 2> $O/bench_c5_under_rocprof.err
cd $R
mkdir -p $O/pmc
bash tools/pmc_bench.sh
python tools/pmc_to_json.py 16 r3 > $O/pmc_to_json.txt 2>&1
cp profiles/r3_tail_conv_pmc.json $O/ 2>/dev/null
bash tools/pmc_bench_sq.sh
python tools/pmc_sq_to_json.py r3 > $O/pmc_sq_to_json.txt 2>&1
cp profiles/r3_tail_conv_sq.json $O/ 2>/dev/null
for d in sq1 sq2; do find gpurun_out/pmc_bench_$d -name "*counter_collection.csv" -exec cp {} $O/pmc/r3_bench_${d}_counter_collection.csv \; 2>/dev/null; done
mkdir -p $O/pmc; cp profiles/pmc/r3_bench_* $O/pmc/ 2>/dev/null
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats_raw.csv \;
find $O/prof -name "*kernel_trace.csv" -exec python tools/trace_summary.py {} 3 \; > $O/bench_kernel_stats.csv
find $O/prof_c5 -name "*kernel_stats.csv" -exec cp {} $O/bench_c5_kernel_stats_raw.csv \;
find $O/prof_c5 -name "*kernel_trace.csv" -exec python tools/c5_trace_summary.py {} 96 \; > $O/c5_kernels.txt
rm -rf $O/prof $O/prof_c5 gpurun_out/pmc_bench_fetch gpurun_out/pmc_bench_write gpurun_out/pmc_bench_sq1 gpurun_out/pmc_bench_sq2
tail -6 $O/pytest_all.txt; tail -2 $O/smoke.txt; cut -c1-600 $O/bench.json; for c in c3 c4 c5; do cut -c1-300 $O/bench_$c.json; tail -2 $O/bench_$c.err; done; head -8 $O/bench_kernel_stats.csv; head -12 $O/pmc_to_json.txt
