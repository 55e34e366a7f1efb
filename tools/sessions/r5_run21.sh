#!/bin/bash
# round-5 session 21: ColorMNet chain: decoder input in one launch (4 before), CBAM MLP with independent loads, usage update without its memset: tests, c5 x3
R=$PWD; O=$R/gpurun_out/r5v; mkdir -p $O
timeout 900 python -m pytest tests/test_colormnet_net.py tests/test_colormnet.py tests/test_gpu_configs.py -m gpu -q 2>&1 | tail -5 > $O/pytest.txt
for i in 1 2 3; do timeout 400 python bench.py --config c5 --steps 8 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_c5_$i.json 2> $O/bench_c5_$i.err; python - $O/bench_c5_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["whole_path_frac"])
PY
done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_c5 -- python3 $R/bench.py --config c5 --no-cpu-baseline --no-extras --steps 6 --warmup 3 > $O/bench_c5_under_rocprof.json 2> $O/bench_c5_under_rocprof.err
cd $R
find $O/prof_c5 -name "*kernel_trace.csv" -exec python tools/c5_trace_summary.py {} 96 \; > $O/c5_kernels.txt
rm -rf $O/prof_c5
cat $O/pytest.txt; grep -E "cbam|decoder_in|usage|fillBuffer" $O/c5_kernels.txt | cut -c1-150; grep -A8 "^# per stream" $O/c5_kernels.txt | cut -c1-170
