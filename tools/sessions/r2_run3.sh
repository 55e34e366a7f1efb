#!/bin/bash
O=gpurun_out/r2c; mkdir -p $O
python -m pytest tests/test_gpu_deoldify.py -m gpu -q -x -k fused_final 2>&1 | tail -40 > $O/pytest_fused.txt
python -m pytest tests/test_havc_harness.py tests/test_gpu_kernels.py tests/test_gpu_deoldify.py tests/test_gpu_boundary.py -m gpu -q --maxfail=20 2>&1 | tail -60 > $O/pytest.txt
python tools/conv_bench.py 16 5 l8blur,l7blur 60,62,64,65,73 > $O/convbench_blur.txt 2>&1
python tools/gpu_profile.py wide 560 16 > $O/perop_wide560_b16.txt 2>&1
HAVC_ATTENTION_V1=1 python tools/gpu_profile.py wide 560 16 2>&1 | grep "conv.3 " > $O/attn_v1.txt
python bench.py --steps 20 --warmup 5 --sustain-seconds 5 > $O/bench.json 2> $O/bench.err
cat $O/pytest_fused.txt | tail -25; tail -12 $O/pytest.txt; cat $O/convbench_blur.txt; grep "conv.3 " $O/perop_wide560_b16.txt $O/attn_v1.txt; cat $O/bench.json
