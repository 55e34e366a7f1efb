#!/bin/bash
# round-5 session 13: the cheaper GELU (logistic of an odd polynomial) in the conv epilogue: DDColor / ColorMNet / kernel tests, per-group table, c3 / c4 lines;
# c5 with the read-ahead under rocprofv3: per-stream timeline of the steady frames
R=$PWD; O=$R/gpurun_out/r5m; mkdir -p $O
timeout 900 python -m pytest tests/test_ddcolor.py tests/test_gpu_epilogue_special.py tests/test_gpu_kernels.py tests/test_colormnet_net.py tests/test_gpu_configs.py -m gpu -q 2>&1 | tail -5 > $O/pytest.txt
timeout 600 python tools/ddcolor_bench.py 512 128 2>&1 | grep -v amdgpu | head -20 > $O/ddcolor_bench_512_b128.txt
for c in c3 c4; do timeout 400 python bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_c5 -- python3 $R/bench.py --config c5 --no-cpu-baseline --no-extras --steps 6 --warmup 3 > $O/bench_c5_under_rocprof.json 2> $O/bench_c5_under_rocprof.err
cd $R
find $O/prof_c5 -name "*kernel_trace.csv" -exec python tools/c5_trace_summary.py {} 96 12:6 \; > $O/c5_kernels.txt
rm -rf $O/prof_c5
cat $O/pytest.txt $O/ddcolor_bench_512_b128.txt; for c in c3 c4; do cut -c1-200 $O/bench_$c.json; done; head -12 $O/c5_kernels.txt
