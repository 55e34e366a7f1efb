#!/bin/bash
O=gpurun_out/r2e; mkdir -p $O
python -m pytest tests/test_ddcolor.py tests/test_gpu_deoldify.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q --maxfail=20 2>&1 | tail -30 > $O/pytest.txt
python tools/conv_bench.py 16 5 l8blur,l7blur 60,65,66,67 > $O/convbench_blur.txt 2>&1
python tools/ddcolor_bench.py 512 8 > $O/ddcolor_bench_512_b8.txt 2>&1
HAVC_MHA_V1=1 HAVC_AUTOTUNE=0 python tools/ddcolor_bench.py 512 8 > $O/ddcolor_bench_512_b8_r1kernels.txt 2>&1
python tools/gpu_profile.py wide 560 16 > $O/perop_wide560_b16.txt 2>&1
python bench.py --steps 20 --warmup 5 --sustain-seconds 5 > $O/bench.json 2> $O/bench.err
python bench.py --config c4 --steps 10 --warmup 2 > $O/bench_c4.json 2> $O/bench_c4.err
tail -8 $O/pytest.txt; cat $O/convbench_blur.txt; head -18 $O/ddcolor_bench_512_b8.txt; head -18 $O/ddcolor_bench_512_b8_r1kernels.txt; tail -12 $O/perop_wide560_b16.txt; cat $O/bench.json $O/bench_c4.json
