#!/bin/bash
# round-6 re-verification on the last commit (after the closing session: precise attention rows as ds_read_b128, smoke on video / stable, stream rule, traced stress tool,
# the default-mode per-frame leg of bench.py): whole GPU suite, smoke, the default bench line
R=$PWD; O=$R/gpurun_out/r6y; mkdir -p $O
timeout 1300 python -m pytest tests -m gpu -q --maxfail=30 2>&1 | tail -5 > $O/pytest_all.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout 1500 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
tail -3 $O/pytest_all.txt; tail -2 $O/smoke.txt; grep "bench: \[" $O/bench.err | sed -n '1p;$p'; python - <<'PY'
import json
o=json.loads([l for l in open("gpurun_out/r6y/bench.json") if l.startswith("{")][-1])
print({k:o[k] for k in ("value","ms_per_step","dtype")}, "frac", o["roofline"]["frac"], "traffic", o["roofline"]["traffic"], o["roofline"].get("traffic_source","")[:40])
print("contract", {k:o["contract"][k] for k in ("value","meets_contract","seconds_timed")})
for c,v in o.get("other_configs",{}).items(): print(c, v.get("value"), v.get("contract"))
print({k:o[k].get("value", o[k].get("error")) for k in ("sustained","pcie_inclusive","batch1","batch1_low_latency","batch1_default_precise","per_frame_calls_16_threads") if k in o})
PY
