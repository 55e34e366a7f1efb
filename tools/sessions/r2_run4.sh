#!/bin/bash
O=gpurun_out/r2d; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_deoldify.py -m gpu -q --maxfail=20 2>&1 | tail -30 > $O/pytest.txt
python tools/conv_bench.py 16 5 l8blur 60,65,66,67 > $O/convbench_blur.txt 2>&1
python tools/conv_bench.py 16 5 enc3x3_35,enc1x1_35,e3_c3_1x1,e2_c2,e2_c1,e2_c3,e4_c 0,70,71,72,90,91,93,95,96,97,98 > $O/convbench_tiles.txt 2>&1
python tools/gpu_profile.py wide 560 16 > $O/perop_wide560_b16.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err
HAVC_NT_STORE_MB=512 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_nt.json 2> $O/bench_nt.err
tail -8 $O/pytest.txt; cat $O/convbench_blur.txt; cat $O/convbench_tiles.txt; tail -16 $O/perop_wide560_b16.txt; cat $O/bench.json $O/bench_nt.json
