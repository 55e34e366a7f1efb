#!/bin/bash
# round-5 session 6: precise attention on MFMA + fused RGB8 precise epilogue (tests, per-op table, precise leg), batched 1080p tail of a ColorMNet window (tests, c5)
R=$PWD; O=$R/gpurun_out/r5f; mkdir -p $O
python -m pytest tests/test_gpu_precise_models.py tests/test_gpu_precise.py -m gpu -q -x -s 2>&1 | grep -E "seeds|pooled|passed|failed|error|FAILED|ERROR|precise|Error" | tail -60 > $O/pytest_precise.txt
PRECISION=precise TOP=25 python tools/gpu_profile.py wide 560 16 > $O/perop_precise_b16.txt 2>&1
python -m pytest tests/test_colormnet_net.py tests/test_gpu_configs.py -m gpu -q -s 2>&1 | grep -E "passed|failed|error|FAILED|ERROR|worst|c3|c4|c5|deepex" | tail -40 > $O/pytest_cmn.txt
python bench.py --config c5 --steps 8 --warmup 4 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
python bench.py --no-cpu-baseline --no-extras --no-other-configs --steps 6 --warmup 2 > $O/bench_precise_leg.json 2> $O/bench_precise_leg.err
cat $O/pytest_precise.txt; head -24 $O/perop_precise_b16.txt | tail -20; tail -3 $O/perop_precise_b16.txt; cat $O/pytest_cmn.txt; cut -c1-300 $O/bench_c5.json; tail -2 $O/bench_c5.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r5f/bench_precise_leg.json"))
print("headline", d["value"], "precise", {k: d["precise"].get(k) for k in ("value","ms_per_step","error")}, d["precise"].get("roofline",{}).get("avg_launch_ms"))
PY
