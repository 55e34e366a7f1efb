#!/bin/bash
# round-5 session 11: DDColor's prep writes whole 16-byte chunks into a compact image tensor: tests + per-group table + c3 / c4 lines
R=$PWD; O=$R/gpurun_out/r5k; mkdir -p $O
timeout 900 python -m pytest tests/test_ddcolor.py tests/test_gpu_precise_models.py tests/test_gpu_configs.py tests/test_havc_harness.py -m gpu -q 2>&1 | tail -4 > $O/pytest.txt
timeout 600 python tools/ddcolor_bench.py 512 128 2>&1 | grep -v amdgpu | head -20 > $O/ddcolor_bench_512_b128.txt
for c in c3 c4; do timeout 400 python bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; done
cat $O/pytest.txt $O/ddcolor_bench_512_b128.txt; for c in c3 c4; do cut -c1-200 $O/bench_$c.json; done
