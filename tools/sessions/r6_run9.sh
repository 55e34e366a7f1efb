#!/bin/bash
# round-6 session 9: the race stress found ONE mismatch in 2 500 jittered clips (first differing frame 48 = a look-ahead window boundary): bisect by variant
R=$PWD; O=$R/gpurun_out/r6i; mkdir -p $O
export HAVC_TUNE_CACHE=0
run() { name=$1; shift; env "$@" timeout 600 python tools/cmn_race_stress.py 3000 300 60 2>&1 | grep -v "jittered runs, 0 mismatches\|amdgpu.ids" > $O/stress_$name.txt; echo "== $name: $*"; grep -E "variant|MISMATCH|first differing|same seed|cmn_race_stress" $O/stress_$name.txt | head -40; }
run default_L8
run L2 STRESS_LOOKAHEAD=2
run L2_no_read_ahead STRESS_LOOKAHEAD=2 STRESS_READ_AHEAD=0
run L2_sync_lookahead STRESS_LOOKAHEAD=2 STRESS_ASYNC_LOOKAHEAD=0
