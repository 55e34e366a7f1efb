#!/bin/bash
# round 3, session 1: is the tree green on today's box; MFMA-busy counters for the dominant kernel as it stands
O=gpurun_out/r3a; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt
python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
bash tools/pmc_bench_sq.sh
python tools/pmc_sq_to_json.py r3a > $O/sq.txt 2>&1
cp profiles/r3a_tail_conv_sq.json $O/ 2>/dev/null
tail -3 $O/pytest.txt; cat $O/bench.json | head -c 1500; tail -30 $O/sq.txt
