#!/bin/bash
# round-2 GPU session 1: new parity tests, per-op profile, encoder conv micro-benchmarks, bench line
O=gpurun_out/r2a; mkdir -p $O
python -m pytest tests -m gpu -q --maxfail=12 -s 2>&1 | tail -60 > $O/pytest.txt
python tools/gpu_profile.py wide 560 16 > $O/perop_wide560_b16.txt 2>&1
python tools/conv_bench.py 16 5 e1_,e2_,e3_,e4_,enc 0,70,72 > $O/convbench_enc_b16.txt 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -5 $O/pytest.txt; cat $O/bench.json
