#!/bin/bash
O=gpurun_out/r2s; mkdir -p $O
python -m pytest tests/test_gpu_boundary.py -m gpu -q -x 2>&1 | tail -8 > $O/pytest_boundary.txt; tail -4 $O/pytest_boundary.txt
python bench.py --no-cpu-baseline --steps 5 --warmup 2 --sustain-seconds 3 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2s/bench.json").read().strip().splitlines()[-1])
print(d["value"], d.get("batch1"), d.get("per_frame_calls_16_threads"))
PY
