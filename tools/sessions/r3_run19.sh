#!/bin/bash
O=gpurun_out/r3s; mkdir -p $O
timeout 1500 python -m pytest tests/test_colormnet_net.py tests/test_colormnet.py tests/test_colormnet_memory.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest_colormnet.txt
for la in 16; do timeout 900 python bench.py --config c5 --no-cpu-baseline --no-extras --steps 6 --warmup 2 > $O/bench_c5_la$la.json 2> $O/bench_c5_la$la.err
  python -c "
import json;d=json.load(open('$O/bench_c5_la$la.json'));print('LOOKAHEAD=$la', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frames_per_launch'], d['roofline']['frac'])" || tail -5 $O/bench_c5_la$la.err; done | tee $O/bench_c5_lookahead.txt
bash tools/sessions/r3_run18.sh > $O/r3_c5_kernels.txt 2>&1; head -24 $O/r3_c5_kernels.txt
