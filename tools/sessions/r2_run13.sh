#!/bin/bash
O=gpurun_out/r2m; mkdir -p $O
python -m pytest tests/test_ddcolor.py -m gpu -q -x 2>&1 | tail -8 > $O/pytest_dwln.txt
tail -2 $O/pytest_dwln.txt
for V in 0 1 2; do echo "== variant $V"; HAVC_DWLN_VARIANT=$V python tools/dwln_bench.py 16 | grep "fused  "; HAVC_DWLN_VARIANT=$V python tools/dwln_bench.py 8 | grep "fused  "; done 2>&1 | tee $O/dwln_variants.txt
