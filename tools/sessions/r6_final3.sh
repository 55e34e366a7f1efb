#!/bin/bash
# round-6 last confirmation on the final commit: whole GPU suite (incl. the grouped-raster and stream-rule tests), smoke
R=$PWD; O=$R/gpurun_out/r6x; mkdir -p $O
timeout 1300 python -m pytest tests -m gpu -q --maxfail=30 2>&1 | tail -5 > $O/pytest_all.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -3 $O/pytest_all.txt; tail -2 $O/smoke.txt
