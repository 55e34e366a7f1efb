#!/bin/bash
# round-5 session 9: per-kernel evidence for the other configs (rocprofv3 kernel stats of c3 / c4, DDColor per-group table at 128 frames, precise c3 per-kernel stats)
R=$PWD; O=$R/gpurun_out/r5i; mkdir -p $O
timeout 600 python tools/ddcolor_bench.py 512 128 > $O/ddcolor_bench_512_b128.txt 2>&1
cd /tmp; export TMPDIR=/tmp
for c in c3 c4; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -- python3 $R/bench.py --config $c --no-cpu-baseline --no-extras --steps 6 --warmup 2 > $O/bench_${c}_under_rocprof.json 2> $O/bench_${c}_under_rocprof.err
  find $O/prof_$c -name "*kernel_stats.csv" -exec cp {} $O/${c}_kernel_stats_raw_incl_autotune.csv \;
  rm -rf $O/prof_$c
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3p -- python3 $R/bench.py --config c3 --precision precise --batch 16 --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $O/bench_c3_precise_under_rocprof.json 2> $O/bench_c3_precise_under_rocprof.err
find $O/prof_c3p -name "*kernel_stats.csv" -exec cp {} $O/c3_precise_kernel_stats_raw_incl_autotune.csv \;
rm -rf $O/prof_c3p
cd $R
tail -25 $O/ddcolor_bench_512_b128.txt; head -12 $O/c3_kernel_stats_raw_incl_autotune.csv | cut -c1-160; head -10 $O/c3_precise_kernel_stats_raw_incl_autotune.csv | cut -c1-160
