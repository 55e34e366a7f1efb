#!/bin/bash
# round-6 session 6: precise HAVC_F_FUSE_PROJ (DDColor tail folded into the last_shuf conv's fp32 epilogue): tests, per-op table, c3 / c4 precise A/B
R=$PWD; O=$R/gpurun_out/r6f; mkdir -p $O
export HAVC_TUNE_CACHE=0
timeout 900 python -m pytest tests/test_gpu_precise_models.py tests/test_ddcolor.py -m gpu -q -x -s 2>&1 | grep -E "passed|failed|Error|ddcolor precise|c3 precise|c4 precise" | tee $O/pytest.txt
PRECISION=precise TOP=6 timeout 900 python tools/ddcolor_bench.py 512 16 2>&1 | grep -v amdgpu.ids > $O/ddcolor_precise_b16_fused_proj.txt
grep -E "GPU ops total|decoder|refine|colour" $O/ddcolor_precise_b16_fused_proj.txt | cut -c1-150
for cfg in c3 c4; do for f in 0 1; do
  HAVC_DD_PRECISE_FUSE_PROJ=$f timeout 600 python bench.py --config $cfg --precision precise --batch 16 --steps 10 --warmup 2 --min-seconds 2 --no-extras 2>/dev/null | python -c "
import json,sys
o=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=o.get('parity') or {}
print('$cfg precise FUSE_PROJ=$f', o['value'], 'steps', o['steps'], 'parity mean', p.get('ciede2000_mean'), 'p99', p.get('ciede2000_p99'), 'below1', p.get('pixels_with_dE_below_1'), 'max', p.get('ciede2000_max'), 'meets', p.get('meets_contract'))"
done; done | tee $O/c3c4_precise_ab.txt
