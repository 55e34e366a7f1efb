#!/bin/bash
O=gpurun_out/r3f; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_configs.py tests/test_gpu_range.py tests/test_gpu_rccl.py tests/test_colormnet_net.py tests/test_colormnet.py tests/test_colormnet_memory.py tests/test_colormnet_core.py -m gpu -q -x -s --tb=short 2>&1 | tail -40 > $O/pytest_new.txt
cat $O/pytest_new.txt
timeout 600 python tools/range_headroom.py > $O/range_headroom.txt 2>&1; tail -45 $O/range_headroom.txt
