#!/bin/bash
# round-6 session 1: (a) the new kernels' tests, (b) tail conv A/B -- round-5 library, this library with / without the computed first K entries, the
# interleaved schedule (cfg 261) -- interleaved on one box, (c) the ColorMNet race stress under stream jitter, (d) the whole GPU suite with durations
R=$PWD; O=$R/gpurun_out/r6a; mkdir -p $O
export HAVC_TUNE_CACHE=0
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -5 > $O/pytest_kernels.txt
cat $O/pytest_kernels.txt
for rep in 1 2 3; do
  HAVC_MI355_LIB=$R/tools/bin/lib_r5base.so timeout 300 python tools/conv_bench.py 16 7 tail259 61 2>&1 | grep tail259 | sed 's/$/   r5 library/'
  timeout 300 python tools/conv_bench.py 16 7 tail259 61 2>&1 | grep tail259 | sed 's/$/   r6 (k01)/'
  HAVC_K01=0 timeout 300 python tools/conv_bench.py 16 7 tail259 61 2>&1 | grep tail259 | sed 's/$/   r6, HAVC_K01=0/'
  timeout 300 python tools/conv_bench.py 16 7 tail259 261 2>&1 | grep tail259 | sed 's/$/   r6 interleaved schedule/'
done > $O/conv_ab.txt 2>&1
cat $O/conv_ab.txt
timeout 900 python tools/cmn_race_stress.py 100 300 60 > $O/race_stress_100.txt 2>&1
tail -12 $O/race_stress_100.txt
timeout 1500 python -m pytest tests -m gpu -q -x --durations=60 2>&1 | tail -90 > $O/pytest_gpu.txt
tail -75 $O/pytest_gpu.txt | cut -c1-200
