#!/bin/bash
# round-6 session 14: rocprofv3 kernel stats of the bench command INCLUDING its precise (contract) leg -- the average duration of the precise tail conv launches next to
# the live HIP-event figure of the line
R=$PWD; O=$R/gpurun_out/r6n; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-other-configs --steps 10 --warmup 3 > $O/bench_with_precise_under_rocprof.json 2> $O/bench_with_precise_under_rocprof.err
cd $R
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_with_precise_kernel_stats_raw.csv \;
rm -rf $O/prof
head -1 $O/bench_with_precise_kernel_stats_raw.csv; grep -E "conv_pipe_kernel<2, 4, 8, 1" $O/bench_with_precise_kernel_stats_raw.csv | cut -c1-200
python - <<'PY'
import json
o=json.loads([l for l in open("gpurun_out/r6n/bench_with_precise_under_rocprof.json") if l.startswith("{")][-1])
print("fast", o["value"], o["roofline"]["avg_launch_ms"], "precise", o["precise"]["value"], o["precise"]["steps"], o["precise"]["roofline"]["avg_launch_ms"], o["precise"]["roofline"]["launches_timed"])
PY
