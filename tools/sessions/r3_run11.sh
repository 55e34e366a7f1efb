#!/bin/bash
O=gpurun_out/r3k; mkdir -p $O
for rep in 1 2; do for pr in 0 1; do
  HAVC_SETPRIO=$pr python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $O/bench_prio${pr}_$rep.json 2>/dev/null
  python -c "
import json;d=json.load(open('$O/bench_prio${pr}_$rep.json'));print('SETPRIO=$pr rep $rep', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['whole_path_tflops'])"
done; done | tee $O/prio_bench.txt
for pr in 0 1; do HAVC_SETPRIO=$pr python tools/gpu_profile.py wide 560 16 > $O/perop_prio$pr.txt 2>&1; grep "whole pass\|^tail\|encoder (layers.0.6)\|^layers" $O/perop_prio$pr.txt | head -12; done
for pr in 0 1; do HAVC_SETPRIO=$pr python tools/ddcolor_bench.py 512 16 2>&1 | grep "GPU ops total"; done | tee $O/prio_dd.txt
python -m pytest tests/test_gpu_kernels.py -m gpu -q 2>&1 | tail -2
