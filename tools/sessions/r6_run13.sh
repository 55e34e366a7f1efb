#!/bin/bash
# round-6 session 13: traced race stress, third long run (seeds 100000 ..)
R=$PWD; O=$R/gpurun_out/r6m; mkdir -p $O
export HAVC_TUNE_CACHE=0
STRESS_TRACE=1 STRESS_SEED0=100000 timeout 2350 python tools/cmn_race_stress.py 24000 300 60 2>&1 | grep -v "amdgpu.ids" > $O/stress_traced3.txt
grep -v "jittered runs, 0 mismatches" $O/stress_traced3.txt | head -60; tail -2 $O/stress_traced3.txt
