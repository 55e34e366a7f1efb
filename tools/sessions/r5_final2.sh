#!/bin/bash
# round-5 closing session: the whole GPU suite on the FINAL code, the default bench line, c3 at its new 128 frames per step (stand-alone line + PMC traffic), c4 / c5 lines
R=$PWD; O=$R/gpurun_out/r5y; mkdir -p $O
timeout 1300 python -m pytest tests -m gpu -q --maxfail=30 -s 2>&1 | grep -E "^seeds|^pooled|passed|failed|error|FAILED|ERROR|low-latency|precise|@1080p|worst frame|deepex|zhang|HAVC_SPLITK" | tail -95 > $O/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout 600 bash tools/pmc_cfg.sh c3; python tools/pmc_cfg_to_json.py c3 r5 gpurun_out 128 > $O/pmc_c3_to_json.txt 2>&1
cp profiles/r5_c3_pmc.json $O/; mkdir -p $O/pmc; cp profiles/pmc/r5_c3_* $O/pmc/ 2>/dev/null
rm -rf gpurun_out/pmc_c3_fetch gpurun_out/pmc_c3_write
for c in c3 c4 c5; do timeout 400 python bench.py --config $c --steps 8 --warmup 3 > $O/bench_$c.json 2> $O/bench_$c.err; done
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
tail -6 $O/pytest_all.txt; tail -1 $O/smoke.txt; tail -6 $O/pmc_c3_to_json.txt; for c in c3 c4 c5; do cut -c1-200 $O/bench_$c.json; done; grep "bench:" $O/bench.err | tail -3; cut -c1-300 $O/bench.json
