#!/bin/bash
# round-5 session 14: ColorMNet: decoder enqueued before the read-ahead (mark), look-ahead context at the lowest stream priority: tests, A/B of the priority,
# per-frame timeline under rocprofv3
R=$PWD; O=$R/gpurun_out/r5n; mkdir -p $O
timeout 900 python -m pytest tests/test_colormnet_net.py tests/test_gpu_configs.py -m gpu -q -s 2>&1 | grep -E "passed|failed|error|FAILED|ERROR|read-ahead|Error" | tail -30 > $O/pytest.txt
for pr in 0 low 0 low; do HAVC_CMN_LOOKAHEAD_PRIORITY=$pr timeout 400 python bench.py --config c5 --steps 8 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_c5_pr$pr.json 2> $O/bench_c5_pr$pr.err; python - $O/bench_c5_pr$pr.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1][-12:], d["value"], d["config"].get("host_enqueue_share_of_wall"), d["config"].get("read_ahead"))
PY
done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_c5 -- python3 $R/bench.py --config c5 --no-cpu-baseline --no-extras --steps 6 --warmup 3 > $O/bench_c5_under_rocprof.json 2> $O/bench_c5_under_rocprof.err
cd $R
find $O/prof_c5 -name "*kernel_trace.csv" -exec python tools/c5_trace_summary.py {} 96 \; > $O/c5_kernels.txt
rm -rf $O/prof_c5
cat $O/pytest.txt; grep -A60 "^# per stream" $O/c5_kernels.txt | cut -c1-160
