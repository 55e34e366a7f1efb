#!/bin/bash
# round-5 session 2: the 64-byte-segment prep kernel (parity tests + per-op time), the fused shuffle+blur conv for the 70 -> 140 stage (HAVC_FUSE_BLUR_MIN_H=64), short bench
R=$PWD; O=$R/gpurun_out/r5b; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_deoldify.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -5 > $O/pytest.txt
TOP=30 python tools/gpu_profile.py wide 560 64 > $O/perop_b64.txt 2>&1
HAVC_FUSE_BLUR_MIN_H=64 TOP=30 python tools/gpu_profile.py wide 560 64 > $O/perop_b64_fuse70.txt 2>&1
python bench.py --no-cpu-baseline --no-extras --no-other-configs --no-precise --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
HAVC_FUSE_BLUR_MIN_H=64 python bench.py --no-cpu-baseline --no-extras --no-other-configs --no-precise --steps 10 --warmup 3 > $O/bench_fuse70.json 2> $O/bench_fuse70.err
cat $O/pytest.txt; grep -E "prep|layers.6|whole|total" $O/perop_b64.txt; grep -E "prep|layers.6|whole|total" $O/perop_b64_fuse70.txt; cut -c1-300 $O/bench.json; cut -c1-300 $O/bench_fuse70.json
