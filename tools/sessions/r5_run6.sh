#!/bin/bash
# round-5 session 6: precise attention on MFMA + fused RGB8 precise epilogue (tests, per-op table), batched 1080p tail of a ColorMNet window (tests, c5),
# the default bench line with progress stamps (every command under its own timeout)
R=$PWD; O=$R/gpurun_out/r5f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_precise_models.py tests/test_gpu_precise.py -m gpu -q -x -s 2>&1 | grep -E "seeds|pooled|passed|failed|error|FAILED|ERROR|precise|Error|assert" | tail -60 > $O/pytest_precise.txt
PRECISION=precise TOP=25 timeout 600 python tools/gpu_profile.py wide 560 16 > $O/perop_precise_b16.txt 2>&1
timeout 900 python -m pytest tests/test_colormnet_net.py tests/test_gpu_configs.py tests/test_gpu_deoldify.py -m gpu -q -s 2>&1 | grep -E "passed|failed|error|FAILED|ERROR|worst|@1080p|deepex|low-latency|assert" | tail -50 > $O/pytest_cmn.txt
timeout 600 python bench.py --config c5 --steps 8 --warmup 4 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err
cat $O/pytest_precise.txt; head -24 $O/perop_precise_b16.txt | tail -20; tail -3 $O/perop_precise_b16.txt; cat $O/pytest_cmn.txt; cut -c1-300 $O/bench_c5.json; tail -2 $O/bench_c5.err; grep "bench:" $O/bench.err; cut -c1-600 $O/bench.json
