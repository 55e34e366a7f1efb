#!/bin/bash
# round-6 session 5: the precise fused depthwise 7x7 + LayerNorm -- kernel test, DDColor precise tests, per-op table, c3 / c4 precise A/B (HAVC_DD_FUSE_DWLN = 0 / 1)
R=$PWD; O=$R/gpurun_out/r6e; mkdir -p $O
export HAVC_TUNE_CACHE=0
timeout 900 python -m pytest tests/test_gpu_precise_models.py tests/test_ddcolor.py -m gpu -q -x 2>&1 | tail -12 | tee $O/pytest.txt
PRECISION=precise TOP=6 timeout 900 python tools/ddcolor_bench.py 512 16 2>&1 | grep -v amdgpu.ids > $O/ddcolor_precise_b16_fused_dwln.txt
grep -E "GPU ops total|encoder stage|colour|decoder|refine|stage2" $O/ddcolor_precise_b16_fused_dwln.txt | cut -c1-150
for cfg in c3 c4; do for f in 0 1; do
  HAVC_DD_FUSE_DWLN=$f timeout 600 python bench.py --config $cfg --precision precise --batch 16 --steps 10 --warmup 2 --min-seconds 2 --no-extras 2>/dev/null | python -c "
import json,sys
o=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=o.get('parity') or {}
print('$cfg precise FUSE_DWLN=$f', o['value'], 'steps', o['steps'], 'parity p99', p.get('ciede2000_p99'), 'below1', p.get('pixels_with_dE_below_1'), 'meets', p.get('meets_contract'))"
done; done | tee $O/c3c4_precise_ab.txt
