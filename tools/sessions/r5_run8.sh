#!/bin/bash
# round-5 session 8: cheap operating-point probes (frames per step, one stream vs two) + the split-K A/B test
R=$PWD; O=$R/gpurun_out/r5h; mkdir -p $O
B="--no-cpu-baseline --no-extras --no-precise --no-other-configs --steps 6 --warmup 2"
for b in 64 96; do timeout 400 python bench.py $B --batch $b > $O/bench_b$b.json 2> $O/bench_b$b.err; done
HAVC_TWO_STREAMS=0 timeout 400 python bench.py $B --batch 64 > $O/bench_b64_one_stream.json 2> $O/bench_b64_one_stream.err
for b in 64 128; do timeout 400 python bench.py --config c3 --no-cpu-baseline --no-extras --steps 6 --warmup 2 --batch $b > $O/bench_c3_b$b.json 2> $O/bench_c3_b$b.err; done
for b in 32 64; do timeout 400 python bench.py --config c4 --no-cpu-baseline --no-extras --steps 6 --warmup 2 --batch $b > $O/bench_c4_b$b.json 2> $O/bench_c4_b$b.err; done
timeout 600 python -m pytest tests/test_gpu_splitk_fused.py -m gpu -q -s 2>&1 | tail -5 > $O/pytest_splitk.txt
for f in $O/bench_*.json; do echo $f; cut -c1-220 $f | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/value \1 ms_per_step \2/'; done; cat $O/pytest_splitk.txt; tail -2 $O/bench_b96.err
