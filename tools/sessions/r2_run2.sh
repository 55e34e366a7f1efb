#!/bin/bash
# round-2 GPU session 2: full GPU suite (new boundary / merge / autotune tests), small-tile ablation, per-op profile, bench c2 + c4
O=gpurun_out/r2b; mkdir -p $O
python -m pytest tests -m gpu -q --maxfail=30 2>&1 | tail -150 > $O/pytest.txt
python tools/conv_bench.py 16 5 enc3x3_35,e3_c3_1x1,enc1x1_35 70,74,75,76,77 > $O/convbench_abl70.txt 2>&1
python tools/gpu_profile.py wide 560 16 > $O/perop_wide560_b16.txt 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python bench.py --config c4 --steps 10 --warmup 2 > $O/bench_c4.json 2> $O/bench_c4.err
tail -30 $O/pytest.txt; cat $O/bench.json; cat $O/bench_c4.json; tail -3 $O/bench.err $O/bench_c4.err
