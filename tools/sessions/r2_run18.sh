#!/bin/bash
# round-2 final GPU session: whole GPU suite, smoke, default bench, rocprofv3 kernel stats + PMC passes of the bench command, c3 / c4
R=$PWD; O=$R/gpurun_out/r2u; mkdir -p $O
python -m pytest tests -m gpu -q --maxfail=30 2>&1 | tail -40 > $O/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --config c3 --no-cpu-baseline --no-extras > $O/bench_c3.json 2>/dev/null
python bench.py --config c4 --no-cpu-baseline --no-extras > $O/bench_c4.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cd $R
bash tools/pmc_bench.sh
python tools/pmc_to_json.py 16 r2 > $O/pmc_to_json.txt 2>&1
cp profiles/r2_tail_conv_pmc.json $O/ 2>/dev/null
mkdir -p $O/pmc; cp profiles/pmc/r2_* $O/pmc/ 2>/dev/null
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats_raw.csv \;
find $O/prof -name "*kernel_trace.csv" -exec python tools/trace_summary.py {} 3 \; > $O/bench_kernel_stats.csv
tail -6 $O/pytest_all.txt; tail -2 $O/smoke.txt; cat $O/bench.json; cat $O/bench_c3.json; cat $O/bench_c4.json; cat $O/bench_under_rocprof.json | cut -c1-400; head -8 $O/bench_kernel_stats.csv; cat $O/pmc_to_json.txt | head -30
