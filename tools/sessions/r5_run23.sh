#!/bin/bash
# round-5 session 23: the long-clip ColorMNet test eight times in fresh processes: which assertion is fragile, and by how much
R=$PWD; O=$R/gpurun_out/r5y2; mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do timeout 300 python -m pytest tests/test_colormnet_net.py -m gpu -q -s -k "long_clip" 2>&1 | grep -E "long clip|passed|failed|assert|Error|^E " | head -12 | sed "s/^/run $i: /" >> $O/long_clip.txt; done
cut -c1-330 $O/long_clip.txt
