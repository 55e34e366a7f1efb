#!/bin/bash
# round-5 session 20: c5 against the look-ahead window (frames per batched key-encoder pass)
R=$PWD; O=$R/gpurun_out/r5t; mkdir -p $O
for la in 16 32 64 16 32; do b=32; [ $la = 64 ] && b=64; HAVC_CMN_LOOKAHEAD=$la timeout 300 python bench.py --config c5 --batch $b --steps 8 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_c5_la$la.json 2> $O/bench_c5_la$la.err; python - $O/bench_c5_la$la.json $la <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("lookahead", sys.argv[2], d["value"], d.get("parity", {}))
except Exception as e:
    print("lookahead", sys.argv[2], "failed", e); print(open(sys.argv[1][:-4]+"err").read()[-600:])
PY
done
