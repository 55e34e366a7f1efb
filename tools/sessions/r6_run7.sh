#!/bin/bash
# round-6 session 7: two queries per thread in the precise multi-head attention (mha32_p_kernel<2>): tests, per-op, c3 precise A/B (HAVC_MHA_P_QB = 1 / 2)
R=$PWD; O=$R/gpurun_out/r6g; mkdir -p $O
export HAVC_TUNE_CACHE=0
timeout 900 python -m pytest tests/test_gpu_precise_models.py -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest.txt
for qb in 1 2; do
  HAVC_MHA_P_QB=$qb PRECISION=precise TOP=5 timeout 900 python tools/ddcolor_bench.py 512 16 2>&1 | grep -E "GPU ops total|colour transformer|cross_attention_layers.2.attn" | sed "s/^/QB=$qb  /" | cut -c1-150
done | tee $O/perop.txt
for qb in 1 2 1 2; do
  HAVC_MHA_P_QB=$qb timeout 600 python bench.py --config c3 --precision precise --batch 16 --steps 10 --warmup 2 --min-seconds 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('c3 precise QB=$qb', o['value'], 'steps', o['steps'])"
done | tee $O/c3_precise_ab.txt
