#!/bin/bash
# round-5 session 19: the look-ahead pass on a CU-masked stream (HAVC_CMN_LOOKAHEAD_CUS): c5 sweep
R=$PWD; O=$R/gpurun_out/r5s; mkdir -p $O
for cus in 0 240 224 208 192 160 0 224; do HAVC_CMN_LOOKAHEAD_CUS=$cus timeout 300 python bench.py --config c5 --steps 8 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_c5_$cus.json 2> $O/bench_c5_$cus.err; python - $O/bench_c5_$cus.json $cus <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("cus", sys.argv[2], d["value"], d["parity"] if "parity" in d else "")
except Exception as e:
    print("cus", sys.argv[2], "failed", e); print(open(sys.argv[1][:-4]+"err").read()[-600:])
PY
done
