#!/bin/bash
# round-5 closing session, last edition (the banked read on scratch slots of its own, read-ahead owner bookkeeping): the whole GPU suite with the text of any failure, c5
R=$PWD; O=$R/gpurun_out/r5f5; mkdir -p $O
timeout 1400 python -m pytest tests -m gpu -q --maxfail=30 -s -rf > $O/pytest_full.txt 2>&1
grep -E "^seeds|^pooled|passed|failed|error|FAILED|ERROR|low-latency|precise|@1080p|worst frame|deepex|zhang|HAVC_SPLITK|read-ahead|long clip" $O/pytest_full.txt | tail -100 > $O/pytest_all.txt
grep -E "^E |^FAILED|Error" $O/pytest_full.txt | head -40 > $O/pytest_failures.txt
timeout 300 python bench.py --config c5 --steps 8 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_c5.json 2> $O/bench_c5.err
tail -4 $O/pytest_all.txt; cat $O/pytest_failures.txt | cut -c1-300; cut -c1-160 $O/bench_c5.json
