#!/bin/bash
# round-5 session 25: test_colormnet_net.py (all GPU tests, suite order inside the file) twelve times in fresh processes: more samples of the one-off failure
R=$PWD; O=$R/gpurun_out/r5z3; mkdir -p $O
for i in $(seq 1 12); do timeout 300 python -m pytest tests/test_colormnet.py tests/test_colormnet_core.py tests/test_colormnet_memory.py tests/test_colormnet_net.py -m gpu -q -s > $O/run_$i.txt 2>&1; tail -1 $O/run_$i.txt | sed "s/^/run $i: /"; grep -E "ATTEMPT 1 FAILED|^E |FAILED" $O/run_$i.txt | head -12 | cut -c1-300; done
