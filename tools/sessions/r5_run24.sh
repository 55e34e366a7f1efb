#!/bin/bash
# round-5 session 24: hunting the one-off failure of the long-clip test: the ColorMNet GPU tests of test_colormnet_net.py 14 times in fresh processes, first failure text kept
R=$PWD; O=$R/gpurun_out/r5z2; mkdir -p $O
for i in $(seq 1 14); do timeout 300 python -m pytest tests/test_colormnet_net.py -m gpu -q -x -k "long_clip or read_ahead or lookahead or fast_step" > $O/run_$i.txt 2>&1; tail -1 $O/run_$i.txt | sed "s/^/run $i: /"; if grep -q failed $O/run_$i.txt; then grep -E "^E |assert|Error" $O/run_$i.txt | head -20 | cut -c1-300; fi; done
