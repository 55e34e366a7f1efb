#!/bin/bash
# round-6 session 10: the traced race stress (STRESS_TRACE=1: per-frame sums of the decoder output, usage / life counters and banks, enqueued on the step's stream):
# as many jittered clips as 35 minutes hold, to catch the ~7e-5-per-clip event WITH the quantity that diverges first
R=$PWD; O=$R/gpurun_out/r6j; mkdir -p $O
export HAVC_TUNE_CACHE=0
STRESS_TRACE=1 timeout 2200 python tools/cmn_race_stress.py 18000 300 60 2>&1 | grep -v "amdgpu.ids" > $O/stress_traced.txt
grep -v "jittered runs, 0 mismatches" $O/stress_traced.txt | head -60; tail -2 $O/stress_traced.txt
