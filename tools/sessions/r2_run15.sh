#!/bin/bash
O=gpurun_out/r2p; mkdir -p $O
python -m pytest tests/test_ddcolor.py -m gpu -q -x 2>&1 | tail -15 > $O/pytest.txt
tail -15 $O/pytest.txt
KINDS=1 TOP=5 python tools/ddcolor_bench.py 512 16 > $O/ddcolor_b16.txt 2>&1; head -32 $O/ddcolor_b16.txt
