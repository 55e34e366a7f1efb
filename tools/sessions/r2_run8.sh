#!/bin/bash
# stagger experiment: start-up spread of the first round of blocks of conv_pipe_kernel (HAVC_CONV_STAGGER, 10-ns ticks)
O=gpurun_out/r2h; mkdir -p $O
for S in 0 400 800 1600 3200 6400 12800 0x40000000 0x40000640 0; do
  echo "== stagger $S" >> $O/stagger.txt
  HAVC_CONV_STAGGER=$S python tools/conv_bench.py 16 7 tail259,tail256,l7conv,l6conv >> $O/stagger.txt 2>&1
done
for S in 0 1600 3200 0; do
  echo "== stagger $S" >> $O/stagger_bench.txt
  HAVC_CONV_STAGGER=$S python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 >> $O/stagger_bench.txt 2>&1
done
cat $O/stagger.txt; grep -o '"value": [0-9.]*\|"avg_launch_ms": [0-9.]*\|== stagger.*' $O/stagger_bench.txt
