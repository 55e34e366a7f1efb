#!/bin/bash
O=gpurun_out/r3e; mkdir -p $O
for c in c5 c3 c4; do
  timeout 900 python bench.py --config $c --steps 6 --warmup 2 > $O/bench_$c.json 2> $O/bench_$c.err
  tail -3 $O/bench_$c.err; head -c 3000 $O/bench_$c.json; echo
done
