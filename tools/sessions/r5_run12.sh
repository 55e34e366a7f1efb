#!/bin/bash
# round-5 session 12: ColorMNet read-ahead (the read of frame t+1 under the decoder of frame t): tests, c5 A/B; DDColor ab map as 16-byte chunks: tests
R=$PWD; O=$R/gpurun_out/r5l; mkdir -p $O
timeout 900 python -m pytest tests/test_colormnet_net.py tests/test_ddcolor.py tests/test_gpu_configs.py -m gpu -q -s 2>&1 | grep -E "passed|failed|error|FAILED|ERROR|read-ahead|deepex|Error" | tail -30 > $O/pytest.txt
for ra in 0 1 0 1; do HAVC_CMN_READ_AHEAD=$ra timeout 400 python bench.py --config c5 --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_c5_ra$ra.json 2> $O/bench_c5_ra$ra.err; cut -c1-160 $O/bench_c5_ra$ra.json; done
cat $O/pytest.txt
