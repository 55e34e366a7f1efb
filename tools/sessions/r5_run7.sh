#!/bin/bash
# round-5 session 7: S = f.g of the precise attention on MFMA (tests, per-op table), the ColorMNet window squash, the default bench line of the final code
R=$PWD; O=$R/gpurun_out/r5g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_precise_models.py tests/test_gpu_precise.py -m gpu -q -x -s 2>&1 | grep -E "pooled|passed|failed|error|FAILED|ERROR|Error|assert" | tail -30 > $O/pytest_precise.txt
PRECISION=precise TOP=12 timeout 600 python tools/gpu_profile.py wide 560 16 > $O/perop_precise_b16.txt 2>&1
HAVC_PRECISE_ATTN_S_MFMA=0 PRECISION=precise TOP=12 timeout 600 python tools/gpu_profile.py wide 560 16 > $O/perop_precise_b16_valu_s.txt 2>&1
timeout 900 python -m pytest tests/test_colormnet_net.py tests/test_gpu_configs.py -m gpu -q 2>&1 | tail -5 > $O/pytest_cmn.txt
timeout 600 python bench.py --config c5 --steps 8 --warmup 4 > $O/bench_c5.json 2> $O/bench_c5.err
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
cat $O/pytest_precise.txt; head -12 $O/perop_precise_b16.txt | tail -9; tail -1 $O/perop_precise_b16.txt; grep "layers.5.conv.3 \|whole" $O/perop_precise_b16_valu_s.txt; cat $O/pytest_cmn.txt; cut -c1-250 $O/bench_c5.json; grep "bench:" $O/bench.err; cut -c1-300 $O/bench.json
