#!/bin/bash
# round-2 GPU session 16: whole GPU suite, smoke, bench (default), bench c4, ddcolor bench
R=$PWD; O=$R/gpurun_out/r2r; mkdir -p $O
python -m pytest tests -m gpu -q --maxfail=30 2>&1 | tail -40 > $O/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --config c4 --no-cpu-baseline --no-extras > $O/bench_c4.json 2> $O/bench_c4.err
python tools/ddcolor_bench.py 512 8 > $O/ddcolor_b8.txt 2>&1
tail -8 $O/pytest_all.txt; tail -2 $O/smoke.txt; cat $O/bench.json; cat $O/bench_c4.json; head -6 $O/ddcolor_b8.txt
