#!/bin/bash
# round-5 session 3: precise mode for DDColor / Zhang / the HAVC graphs (new kernels of csrc/precise2.hip): op tests, model tests, config-size contract tests;
# regression of the fast-path tests whose kernels got a precise switch
R=$PWD; O=$R/gpurun_out/r5c; mkdir -p $O
python -m pytest tests/test_gpu_precise_models.py -m gpu -q -x -s 2>&1 | tail -40 > $O/pytest_precise_models.txt
python -m pytest tests/test_gpu_precise.py tests/test_zhang.py tests/test_ddcolor.py -m gpu -q 2>&1 | tail -15 > $O/pytest_regress.txt
cat $O/pytest_precise_models.txt $O/pytest_regress.txt
