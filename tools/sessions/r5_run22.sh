#!/bin/bash
# round-5 session 22: the long-clip ColorMNet test failed once inside the whole suite (passes alone): the four ColorMNet files in suite order, full failure text
R=$PWD; O=$R/gpurun_out/r5x; mkdir -p $O
timeout 900 python -m pytest tests/test_colormnet.py tests/test_colormnet_core.py tests/test_colormnet_memory.py tests/test_colormnet_net.py -m gpu -q 2>&1 | tail -60 > $O/pytest_a.txt
timeout 900 python -m pytest tests/test_colormnet_net.py -m gpu -q -k "long_clip or read_ahead" 2>&1 | tail -40 > $O/pytest_b.txt
tail -45 $O/pytest_a.txt | cut -c1-400; tail -8 $O/pytest_b.txt | cut -c1-400
