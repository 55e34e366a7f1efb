#!/bin/bash
# round-5 session 4: where the precise mode spends its time: per-op table of the precise wide generator at batch 16; c3 / c4 in precise mode (bench children)
R=$PWD; O=$R/gpurun_out/r5d; mkdir -p $O
PRECISION=precise TOP=60 python tools/gpu_profile.py wide 560 16 > $O/perop_precise_b16.txt 2>&1
python bench.py --config c3 --precision precise --batch 16 --steps 2 --warmup 1 --no-extras > $O/bench_c3_precise.json 2> $O/bench_c3_precise.err
python bench.py --config c4 --precision precise --batch 16 --steps 2 --warmup 1 --no-extras > $O/bench_c4_precise.json 2> $O/bench_c4_precise.err
head -50 $O/perop_precise_b16.txt; tail -22 $O/perop_precise_b16.txt; cut -c1-400 $O/bench_c3_precise.json; tail -3 $O/bench_c3_precise.err; cut -c1-400 $O/bench_c4_precise.json; tail -3 $O/bench_c4_precise.err
