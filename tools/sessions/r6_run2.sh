#!/bin/bash
# round-6 session 2: whole GPU suite on the new defaults (shared packed weights, precise default behind HAVC_PRECISION=fast for the suite), smoke in both
# modes, the 2 000-clip race stress, tail conv r5 library vs this one, the default bench line
R=$PWD; O=$R/gpurun_out/r6b; mkdir -p $O
export HAVC_TUNE_CACHE=0
timeout 1500 python -m pytest tests -m gpu -q -x --durations=25 2>&1 | tail -60 > $O/pytest_gpu.txt
tail -45 $O/pytest_gpu.txt | cut -c1-180
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/smoke.txt
timeout 900 python tools/cmn_race_stress.py 2000 300 60 > $O/race_stress_2000.txt 2>&1
tail -4 $O/race_stress_2000.txt
for rep in 1 2 3; do
  HAVC_MI355_LIB=$R/tools/bin/lib_r5base.so timeout 300 python tools/conv_bench.py 16 7 tail259 61 2>&1 | grep tail259 | sed 's/$/   r5 library/'
  timeout 300 python tools/conv_bench.py 16 7 tail259 61 2>&1 | grep tail259 | sed 's/$/   r6 library/'
done > $O/conv_ab.txt 2>&1
cat $O/conv_ab.txt
timeout 1500 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err; python - <<'PY'
import json
o=json.loads([l for l in open("gpurun_out/r6b/bench.json") if l.startswith("{")][-1])
print({k:o[k] for k in ("value","ms_per_step","dtype")}, o["roofline"]["frac"], o["roofline"]["avg_launch_ms"])
print("contract", o.get("contract"))
print("precise", {k:o["precise"].get(k) for k in ("value","steps","seconds_timed")}, o["precise"]["roofline"])
for c,v in o.get("other_configs",{}).items(): print(c, {k:v.get(k) for k in ("value","steps","seconds_timed","contract")})
print({k:o[k]["value"] for k in ("sustained","pcie_inclusive","batch1","batch1_low_latency","per_frame_calls_16_threads") if k in o})
PY
