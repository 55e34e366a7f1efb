#!/bin/bash
O=gpurun_out/r2o; mkdir -p $O
python -m pytest tests/test_ddcolor.py tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -6 > $O/pytest.txt
tail -3 $O/pytest.txt
python tools/conv_bench.py 16 5 pw1 0 2>&1 | tee $O/pw1.txt
KINDS=1 TOP=5 python tools/ddcolor_bench.py 512 16 > $O/ddcolor_b16.txt 2>&1; head -40 $O/ddcolor_b16.txt
