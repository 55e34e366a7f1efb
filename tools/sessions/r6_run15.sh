#!/bin/bash
# round-6 session 15: traced race stress with LONGER delays (up to 800 us per call: the streams drift by several frames' worth of GPU time), seeds 200000 ..
R=$PWD; O=$R/gpurun_out/r6o; mkdir -p $O
export HAVC_TUNE_CACHE=0
STRESS_TRACE=1 STRESS_SEED0=200000 timeout 1250 python tools/cmn_race_stress.py 9000 800 60 2>&1 | grep -v "amdgpu.ids" > $O/stress_traced_800us.txt
grep -v "jittered runs, 0 mismatches" $O/stress_traced_800us.txt | head -40; tail -2 $O/stress_traced_800us.txt
