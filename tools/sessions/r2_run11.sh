#!/bin/bash
O=gpurun_out/r2k; mkdir -p $O
python -m pytest tests/test_ddcolor.py -m gpu -q -x 2>&1 | tail -15 > $O/pytest_ddcolor.txt
cat $O/pytest_ddcolor.txt
HAVC_DD_FUSE_DWLN=0 python tools/ddcolor_bench.py 512 16 > $O/ddcolor_b16_unfused.txt 2>&1
python tools/ddcolor_bench.py 512 16 > $O/ddcolor_b16.txt 2>&1
python tools/ddcolor_bench.py 512 8 > $O/ddcolor_b8.txt 2>&1
grep -E "colorize|GPU ops|stage|dwconv" $O/ddcolor_b16_unfused.txt $O/ddcolor_b16.txt $O/ddcolor_b8.txt
