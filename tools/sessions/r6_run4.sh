#!/bin/bash
# round-6 session 4: where a PRECISE DDColor pass (c3's contract figure, 298 frames/s against 1 334 fast) spends its time
R=$PWD; O=$R/gpurun_out/r6d; mkdir -p $O
export HAVC_TUNE_CACHE=0
PRECISION=precise KINDS=1 TOP=25 timeout 900 python tools/ddcolor_bench.py 512 16 2>&1 | grep -v amdgpu.ids > $O/ddcolor_precise_b16.txt
cat $O/ddcolor_precise_b16.txt | cut -c1-170
