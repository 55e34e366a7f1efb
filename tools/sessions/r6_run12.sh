#!/bin/bash
# round-6 session 12: the traced race stress once more (another 20 000 clips, other seeds), hunting the ~3e-5-per-clip event with the per-frame trace on
R=$PWD; O=$R/gpurun_out/r6l; mkdir -p $O
export HAVC_TUNE_CACHE=0
STRESS_TRACE=1 STRESS_SEED0=50000 timeout 2300 python tools/cmn_race_stress.py 20000 300 60 2>&1 | grep -v "amdgpu.ids" > $O/stress_traced2.txt
grep -v "jittered runs, 0 mismatches" $O/stress_traced2.txt | head -60; tail -2 $O/stress_traced2.txt
