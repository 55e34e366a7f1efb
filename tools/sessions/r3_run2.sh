#!/bin/bash
# round 3, session 2: first GPU run of the ColorMNet network ops + modules + render
O=gpurun_out/r3b; mkdir -p $O
timeout 1500 python -m pytest tests/test_colormnet_net.py -m gpu -q -x --tb=short 2>&1 | tail -60 > $O/pytest_cmn.txt
cat $O/pytest_cmn.txt
