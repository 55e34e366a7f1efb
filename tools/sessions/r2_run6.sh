#!/bin/bash
# round-2 GPU session 6: ColorMNet kernels, the whole GPU suite, rocprofv3 kernel stats + PMC passes of the bench command
R=$PWD; O=$R/gpurun_out/r2f; mkdir -p $O
python -m pytest tests/test_colormnet.py -m gpu -q 2>&1 | tail -25 > $O/pytest_colormnet.txt
python -m pytest tests -m gpu -q --maxfail=30 2>&1 | tail -40 > $O/pytest_all.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cd $R
bash tools/pmc_bench.sh
python tools/pmc_to_json.py 16 r2 > $O/pmc_to_json.txt 2>&1
cp profiles/r2_tail_conv_pmc.json $O/ 2>/dev/null
mkdir -p $O/pmc; cp profiles/pmc/r2_* $O/pmc/ 2>/dev/null
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
tail -6 $O/pytest_colormnet.txt; tail -12 $O/pytest_all.txt; cat $O/bench_under_rocprof.json; head -12 $O/bench_kernel_stats.csv; cat $O/pmc_to_json.txt | head -30
