#!/bin/bash
# round-6 session 8: (a) precise MHA with ds_read_b128 rows: tests + per-op + c3 precise; (b) fast dwconv7_ln: 512-thread one-row variant (HAVC_DWLN_VARIANT=3) vs the shipped ones
R=$PWD; O=$R/gpurun_out/r6h; mkdir -p $O
export HAVC_TUNE_CACHE=0
timeout 900 python -m pytest tests/test_gpu_precise_models.py -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest.txt
PRECISION=precise TOP=5 timeout 900 python tools/ddcolor_bench.py 512 16 2>&1 | grep -E "GPU ops total|colour transformer|cross_attention_layers.2.attn" | cut -c1-150 | tee $O/perop_precise.txt
for i in 1 2; do timeout 600 python bench.py --config c3 --precision precise --batch 16 --steps 10 --warmup 2 --min-seconds 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c3 precise (b128 rows)', o['value'], 'steps', o['steps'])"; done | tee $O/c3_precise.txt
for v in 0 3 1 0 3; do HAVC_DWLN_VARIANT=$v timeout 300 python tools/dwln_bench.py 64 7 2>&1 | grep "fused  " | sed "s/^/variant $v: /"; done | tee $O/dwln_variants.txt
for v in 0 3 0 3; do HAVC_DWLN_VARIANT=$v timeout 600 python bench.py --config c3 --steps 10 --warmup 2 --min-seconds 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c3 fast DWLN_VARIANT=$v', o['value'], 'steps', o['steps'])"; done | tee $O/c3_fast_dwln.txt
