#!/bin/bash
# round-6 session 11: grouped tile order for GEMMs whose weights exceed an XCD's L2 (kernels.h conv_raster; VERDICT r5 item 4, second half): tests, the ConvNeXt MLP GEMMs
# and c3 with HAVC_RASTER_GROUP = 0 / 1, FETCH_SIZE / WRITE_SIZE of c3's dominant kernel with it on
R=$PWD; O=$R/gpurun_out/r6k; mkdir -p $O
export HAVC_TUNE_CACHE=0
timeout 900 python -m pytest tests/test_ddcolor.py tests/test_gpu_kernels.py tests/test_gpu_epilogue_special.py -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest.txt
for g in 0 1 0 1; do HAVC_RASTER_GROUP=$g timeout 300 python tools/conv_bench.py 128 7 pw1_s2,pw1_s3,pw1n_s2 2>&1 | grep "pw1" | sed "s/^/RASTER_GROUP=$g  /"; done | tee $O/convbench.txt
for g in 0 1 0 1; do HAVC_RASTER_GROUP=$g timeout 600 python bench.py --config c3 --steps 10 --warmup 2 --min-seconds 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c3 fast RASTER_GROUP=$g', o['value'], 'dominant ms', o['roofline']['avg_launch_ms'], 'frac', o['roofline']['frac'])"; done | tee $O/c3_ab.txt
timeout 600 bash tools/pmc_cfg.sh c3; python tools/pmc_cfg_to_json.py c3 r6g > $O/pmc_c3_to_json.txt 2>&1; tail -8 $O/pmc_c3_to_json.txt; cp profiles/r6g_c3_pmc.json $O/ 2>/dev/null
rm -rf gpurun_out/pmc_c3_fetch gpurun_out/pmc_c3_write
