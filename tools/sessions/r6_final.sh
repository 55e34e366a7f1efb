#!/bin/bash
# round-6 closing session: whole GPU suite, smoke (both modes), the default bench line (driver shape: --steps 20), stand-alone c3 / c4 / c5 lines, rocprofv3 kernel
# stats of the bench and of c5, PMC passes (HBM traffic + SQ counters of the tail conv; HBM traffic of the c3 / c4 dominant kernels), per-op tables (fast 64 frames,
# precise 16 frames), the race stress once more.  Every command under its own timeout.
R=$PWD; O=$R/gpurun_out/r6z; mkdir -p $O $O/pmc
timeout 1300 python -m pytest tests -m gpu -q --maxfail=30 -s 2>&1 | grep -E "^seeds|^pooled|passed|failed|error|FAILED|ERROR|low-latency|precise|@1080p|worst frame|deepex|zhang|read-ahead|long clip" | tail -90 > $O/pytest_all.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout 1500 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
for c in c3 c4 c5; do timeout 500 python bench.py --config $c --steps 8 --warmup 3 --min-seconds 2 > $O/bench_$c.json 2> $O/bench_$c.err; done
timeout 400 python tools/cmn_race_stress.py 500 300 60 > $O/race_stress_500.txt 2>&1
TOP=200 timeout 300 python tools/gpu_profile.py wide 560 64 > $O/perop_b64.txt 2>&1
PRECISION=precise TOP=200 timeout 400 python tools/gpu_profile.py wide 560 16 > $O/perop_precise_b16.txt 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-precise --no-other-configs --steps 10 --warmup 3 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -- python3 $R/bench.py --config c5 --no-cpu-baseline --no-extras --steps 6 --warmup 3 > $O/bench_c5_under_rocprof.json 2> $O/bench_c5_under_rocprof.err
cd $R
timeout 900 bash tools/pmc_bench.sh
python tools/pmc_to_json.py 16 r6 > $O/pmc_to_json.txt 2>&1
timeout 900 bash tools/pmc_bench_sq.sh
python tools/pmc_sq_to_json.py r6 > $O/pmc_sq_to_json.txt 2>&1
for c in c3 c4; do timeout 600 bash tools/pmc_cfg.sh $c; python tools/pmc_cfg_to_json.py $c r6 > $O/pmc_${c}_to_json.txt 2>&1; done
cp profiles/r6_tail_conv_pmc.json profiles/r6_tail_conv_sq.json profiles/r6_c3_pmc.json profiles/r6_c4_pmc.json $O/ 2>/dev/null
cp profiles/pmc/r6_* $O/pmc/ 2>/dev/null
for d in sq1 sq2; do find gpurun_out/pmc_bench_$d -name "*counter_collection.csv" -exec cp {} $O/pmc/r6_bench_${d}_counter_collection.csv \; 2>/dev/null; done
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats_raw.csv \;
find $O/prof -name "*kernel_trace.csv" -exec python tools/trace_summary.py {} 3 \; > $O/bench_kernel_stats.csv
find $O/prof_c5 -name "*kernel_stats.csv" -exec cp {} $O/bench_c5_kernel_stats_raw.csv \;
find $O/prof_c5 -name "*kernel_trace.csv" -exec python tools/c5_trace_summary.py {} 96 \; > $O/c5_kernels.txt
rm -rf $O/prof $O/prof_c5 gpurun_out/pmc_bench_fetch gpurun_out/pmc_bench_write gpurun_out/pmc_bench_sq1 gpurun_out/pmc_bench_sq2 gpurun_out/pmc_c3_fetch gpurun_out/pmc_c3_write gpurun_out/pmc_c4_fetch gpurun_out/pmc_c4_write
tail -6 $O/pytest_all.txt; tail -2 $O/smoke.txt; cut -c1-300 $O/bench.json; for c in c3 c4 c5; do cut -c1-160 $O/bench_$c.json; done; head -6 $O/bench_kernel_stats.csv; head -8 $O/pmc_to_json.txt; tail -3 $O/pmc_c3_to_json.txt; tail -1 $O/race_stress_500.txt; head -3 $O/c5_kernels.txt
