#!/bin/bash
O=gpurun_out/r3g; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -x --tb=short 2>&1 | tail -15 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
timeout 600 python tools/range_headroom.py > $O/range_headroom.txt 2>&1; tail -12 $O/range_headroom.txt
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; head -c 2500 $O/bench.json; echo
for c in c3 c4; do timeout 600 python bench.py --config $c --steps 6 --warmup 2 > $O/bench_$c.json 2> $O/bench_$c.err; tail -2 $O/bench_$c.err; python -c "
import json;d=json.load(open('$O/bench_$c.json'));print('$c', d['value'], d['roofline']['frac'], d['parity'])"; done
