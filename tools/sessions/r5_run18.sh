#!/bin/bash
# round-5 session 18: the read's scratch reserved at the banks' capacity (no regrowth + device sync in the middle of a clip): ColorMNet tests, c5 (x3), host probe
R=$PWD; O=$R/gpurun_out/r5r; mkdir -p $O
timeout 900 python -m pytest tests/test_colormnet_net.py tests/test_gpu_configs.py -m gpu -q 2>&1 | tail -3 > $O/pytest.txt
for i in 1 2 3; do timeout 400 python bench.py --config c5 --steps 8 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_c5_$i.json 2> $O/bench_c5_$i.err; python - $O/bench_c5_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["config"].get("host_enqueue_share_of_wall"), d["config"].get("read_ahead"), d["whole_path_frac"])
PY
done
timeout 400 python tools/c5_host_probe.py > $O/probe.json 2> $O/probe.err; grep -v amdgpu $O/probe.err | tail -3 | cut -c1-400
HAVC_BENCH_CPROFILE=1 timeout 400 python bench.py --config c5 --steps 6 --warmup 3 --no-cpu-baseline --no-extras > $O/prof.json 2> $O/prof.txt; grep -v amdgpu $O/prof.txt | head -30 | cut -c1-150
cat $O/pytest.txt
