#!/bin/bash
O=gpurun_out/r3h; mkdir -p $O
timeout 1500 python -m pytest tests/test_havc_harness.py tests/test_tweaks.py tests/test_zhang.py tests/test_ddcolor.py tests/test_colormnet.py tests/test_gpu_boundary.py tests/test_gpu_edges.py -m gpu -q --tb=short 2>&1 | tail -8 > $O/pytest_rest.txt
cat $O/pytest_rest.txt
# schedule experiments on the dominant kernel: cfg 61 = shipped, 101 = pieces in the first NS/4 steps, 102 = three early pieces, 103 = both
python tools/conv_bench.py 16 7 tail259 61,101,102,103,61 > $O/sched.txt 2>&1; cat $O/sched.txt
# non-temporal stores for conv outputs >= 1 GB (r1 of the tail: 2.65 GB per 16 frames)
for mb in 0 1024; do
  HAVC_NT_STORE_MB=$mb python tools/conv_bench.py 16 7 tail259 61 >> $O/nt.txt 2>&1
  HAVC_NT_STORE_MB=$mb python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $O/bench_nt$mb.json 2> /dev/null
  python -c "
import json;d=json.load(open('$O/bench_nt$mb.json'));print('NT_STORE_MB=$mb', d['value'], d['roofline']['avg_launch_ms'], d['whole_path_tflops'])" >> $O/nt.txt
done
cat $O/nt.txt
