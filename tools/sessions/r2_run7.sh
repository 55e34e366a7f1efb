#!/bin/bash
O=gpurun_out/r2g; mkdir -p $O
python -m pytest tests/test_gpu_boundary.py -m gpu -q 2>&1 | tail -15 > $O/pytest_boundary.txt
python tools/ddcolor_bench.py 512 8 > $O/ddcolor_b8.txt 2>&1
python tools/ddcolor_bench.py 512 16 > $O/ddcolor_b16.txt 2>&1
TOP=140 python tools/gpu_profile.py wide 560 16 > $O/perop_full.txt 2>&1
tail -5 $O/pytest_boundary.txt; grep -E "colorize|GPU ops|stage2|encoder stage 2|colour" $O/ddcolor_b8.txt $O/ddcolor_b16.txt; grep "layers.0.6.1\.\|layers.0.6.5\." $O/perop_full.txt
