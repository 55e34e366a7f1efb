#!/bin/bash
# round-5 session 1: same-day baseline of the round-4 library: headline bench (short), per-op tables at batch 16 and 64,
# ablations of the fused shuffle+blur conv (cfg 65 no epilogue, 66 epilogue without stores, 67 plain stores) and the encoder shapes
R=$PWD; O=$R/gpurun_out/r5a; mkdir -p $O
python bench.py --no-cpu-baseline --no-extras --no-other-configs --no-precise --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
TOP=200 python tools/gpu_profile.py wide 560 16 > $O/perop_b16.txt 2>&1
TOP=200 python tools/gpu_profile.py wide 560 64 > $O/perop_b64.txt 2>&1
python tools/conv_bench.py 16 7 l8blur,l7blur,l8nops,l8ps 0,60,65,66,67 > $O/convbench_blur.txt 2>&1
python tools/conv_bench.py 64 7 enc3x3_35,enc1x1_35,e3_c3_1x1 0 > $O/convbench_enc_b64.txt 2>&1
python tools/conv_bench.py 16 7 enc3x3_35,enc1x1_35,e3_c3_1x1 0 > $O/convbench_enc_b16.txt 2>&1
cut -c1-600 $O/bench.json; tail -3 $O/bench.err; tail -20 $O/perop_b64.txt; cat $O/convbench_blur.txt $O/convbench_enc_b64.txt $O/convbench_enc_b16.txt
