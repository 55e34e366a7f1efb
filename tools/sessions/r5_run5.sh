#!/bin/bash
# round-5 session 5: the whole GPU suite on the library with in-kernel split-K reduction / low-latency default / precise models; A/B of the two split-K
# forms (bytes + time per call); c5 and the default bench line
R=$PWD; O=$R/gpurun_out/r5e; mkdir -p $O
python -m pytest tests -m gpu -q --maxfail=30 -s 2>&1 | grep -E "^seeds|^pooled|passed|failed|error|FAILED|ERROR|low-latency|precise" | tail -80 > $O/pytest_all.txt
HAVC_SPLITK_FUSED=1 python tools/splitk_ab.py 35 > $O/splitk_ab.txt 2>&1
HAVC_SPLITK_FUSED=0 python tools/splitk_ab.py 35 >> $O/splitk_ab.txt 2>&1
HAVC_SPLITK_FUSED=1 python tools/splitk_ab.py 6 >> $O/splitk_ab.txt 2>&1
HAVC_SPLITK_FUSED=0 python tools/splitk_ab.py 6 >> $O/splitk_ab.txt 2>&1
python bench.py --config c5 --steps 8 --warmup 4 > $O/bench_c5.json 2> $O/bench_c5.err
HAVC_SPLITK_FUSED=0 python bench.py --config c5 --steps 8 --warmup 4 --no-cpu-baseline > $O/bench_c5_legacy_splitk.json 2> $O/bench_c5_legacy_splitk.err
python bench.py > $O/bench.json 2> $O/bench.err
tail -30 $O/pytest_all.txt; grep -v amdgpu $O/splitk_ab.txt; cut -c1-300 $O/bench_c5.json; cut -c1-200 $O/bench_c5_legacy_splitk.json; cut -c1-500 $O/bench.json; tail -3 $O/bench.err
