#!/bin/bash
O=gpurun_out/r3j; mkdir -p $O
for b in 32 48 64; do
  timeout 900 python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 --batch $b --clip-frames $b > $O/bench_b$b.json 2> $O/bench_b$b.err
  python -c "
import json;d=json.load(open('$O/bench_b$b.json'));print('batch $b', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['whole_path_tflops'])" || tail -3 $O/bench_b$b.err
done
timeout 900 python bench.py --config c5 --steps 6 --warmup 2 > $O/bench_c5.json 2> $O/bench_c5.err; head -c 1800 $O/bench_c5.json; echo
timeout 600 python -m pytest tests/test_colormnet_memory.py tests/test_colormnet_core.py -m gpu -q 2>&1 | tail -2
