#!/bin/bash
# round-4 final GPU session (second: after the epilogue work): whole GPU suite, smoke, default bench (with the precise leg and the c3 / c4 / c5 child legs), the stand-alone c3 / c4 / c5
# lines, the 200-set-up stress, rocprofv3 kernel stats + PMC passes of the bench command, the c5 kernel list
R=$PWD; O=$R/gpurun_out/r4y; mkdir -p $O
python -m pytest tests -m gpu -q --maxfail=30 -s 2>&1 | grep -E "^seeds|^pooled|passed|failed|error|FAILED|ERROR" | tail -60 > $O/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
for c in c3 c4 c5; do python bench.py --config $c --steps 8 --warmup 3 > $O/bench_$c.json 2> $O/bench_$c.err; done
HAVC_TUNE_CACHE=0 timeout 900 python tools/setup_stress.py --reps 50 --threads 4 > $O/setup_stress.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-precise --steps 10 --warmup 3 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -- python3 $R/bench.py --config c5 --no-cpu-baseline --no-extras --steps 6 --warmup 3 > $O/bench_c5_under_rocprof.json 2> $O/bench_c5_under_rocprof.err
cd $R
mkdir -p $O/pmc
bash tools/pmc_bench.sh
python tools/pmc_to_json.py 16 r4 > $O/pmc_to_json.txt 2>&1
cp profiles/r4_tail_conv_pmc.json $O/ 2>/dev/null
bash tools/pmc_bench_sq.sh
python tools/pmc_sq_to_json.py r4 > $O/pmc_sq_to_json.txt 2>&1
cp profiles/r4_tail_conv_sq.json $O/ 2>/dev/null
for d in sq1 sq2; do find gpurun_out/pmc_bench_$d -name "*counter_collection.csv" -exec cp {} $O/pmc/r4_bench_${d}_counter_collection.csv \; 2>/dev/null; done
cp profiles/pmc/r4_bench_* $O/pmc/ 2>/dev/null
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats_raw.csv \;
find $O/prof -name "*kernel_trace.csv" -exec python tools/trace_summary.py {} 3 \; > $O/bench_kernel_stats.csv
find $O/prof_c5 -name "*kernel_stats.csv" -exec cp {} $O/bench_c5_kernel_stats_raw.csv \;
find $O/prof_c5 -name "*kernel_trace.csv" -exec python tools/c5_trace_summary.py {} 96 \; > $O/c5_kernels.txt
rm -rf $O/prof $O/prof_c5 gpurun_out/pmc_bench_fetch gpurun_out/pmc_bench_write gpurun_out/pmc_bench_sq1 gpurun_out/pmc_bench_sq2
tail -8 $O/pytest_all.txt; tail -2 $O/smoke.txt; cut -c1-400 $O/bench.json; for c in c3 c4 c5; do cut -c1-200 $O/bench_$c.json; tail -2 $O/bench_$c.err; done; head -8 $O/bench_kernel_stats.csv; head -12 $O/pmc_to_json.txt; tail -2 $O/setup_stress.txt; grep -c "at::" $O/c5_kernels.txt
