#!/bin/bash
O=gpurun_out/r3c; mkdir -p $O
PEROP=1 timeout 900 python -u tools/colormnet_clip_bench.py 40 216 384 > $O/clip_bench.txt 2>&1
cat $O/clip_bench.txt
