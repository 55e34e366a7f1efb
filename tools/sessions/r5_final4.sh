#!/bin/bash
# round-5 closing session, second edition (after the shorter ColorMNet chain): the whole GPU suite, smoke, c3 / c4 / c5 lines, c5 under rocprofv3, the default bench line;
# the stand-alone c3 / c4 / c5 lines, c5 under rocprofv3 (per-kernel + per-stream + per-frame tables), the default bench line
R=$PWD; O=$R/gpurun_out/r5w; mkdir -p $O
timeout 1400 python -m pytest tests -m gpu -q --maxfail=30 -s 2>&1 | grep -E "^seeds|^pooled|passed|failed|error|FAILED|ERROR|low-latency|precise|@1080p|worst frame|deepex|zhang|HAVC_SPLITK|read-ahead" | tail -100 > $O/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
for c in c3 c4 c5; do timeout 400 python bench.py --config $c --steps 8 --warmup 3 > $O/bench_$c.json 2> $O/bench_$c.err; done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -- python3 $R/bench.py --config c5 --no-cpu-baseline --no-extras --steps 6 --warmup 3 > $O/bench_c5_under_rocprof.json 2> $O/bench_c5_under_rocprof.err
cd $R
find $O/prof_c5 -name "*kernel_stats.csv" -exec cp {} $O/bench_c5_kernel_stats_raw.csv \;
find $O/prof_c5 -name "*kernel_trace.csv" -exec python tools/c5_trace_summary.py {} 96 \; > $O/c5_kernels.txt
rm -rf $O/prof_c5
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
tail -6 $O/pytest_all.txt; tail -1 $O/smoke.txt; for c in c3 c4 c5; do cut -c1-200 $O/bench_$c.json; done; head -3 $O/c5_kernels.txt | cut -c1-200; grep "bench:" $O/bench.err | tail -3; cut -c1-300 $O/bench.json
for la in 8 12; do HAVC_CMN_LOOKAHEAD=$la timeout 300 python bench.py --config c5 --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-140 | sed "s/^/lookahead $la: /" >> $O/c5_lookahead_sweep.txt; done; cat $O/c5_lookahead_sweep.txt
