#!/bin/bash
O=gpurun_out/r2j; mkdir -p $O
for S in 0 12800 25600 0x400031f0 0; do
  echo "== stagger $S" >> $O/stagger2.txt
  HAVC_CONV_STAGGER=$S python tools/conv_bench.py 16 7 tail >> $O/stagger2.txt 2>&1
done
cat $O/stagger2.txt
