#!/bin/bash
# round-6 session 3: the precise form of the fused shuffle + blur conv -- bit identity vs the two-op chain, per-op table of a precise pass with / without it,
# the precise bench leg A/B (HAVC_PRECISE_FUSE_BLUR = 0 / 1, 32 and 48 frames per step)
R=$PWD; O=$R/gpurun_out/r6c; mkdir -p $O
export HAVC_TUNE_CACHE=0
timeout 900 python -m pytest tests/test_gpu_precise.py -m gpu -q -x -k "fused_shuffle_blur or raw_color or precision_argument" 2>&1 | tail -8 | tee $O/pytest_precise.txt
for fb in 1 0; do
  PRECISION=precise FUSE_BLUR=$fb TOP=14 timeout 600 python tools/gpu_profile.py wide 560 16 2>&1 | grep -v amdgpu.ids > $O/perop_precise_b16_fuseblur$fb.txt
  head -18 $O/perop_precise_b16_fuseblur$fb.txt | cut -c1-150; grep -E "^layers.[678] |^tail|whole pass" $O/perop_precise_b16_fuseblur$fb.txt
done
for rep in 1 2; do for fb in 0 1; do
  HAVC_PRECISE_FUSE_BLUR=$fb timeout 900 python bench.py --steps 10 --warmup 2 --no-extras --no-other-configs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=o['precise']
print('FUSE_BLUR=$fb rep $rep fast', o['value'], 'precise', p['value'], p['frames_per_step'], p['steps'], p['roofline']['avg_launch_ms'])"
done; done | tee $O/precise_ab.txt
HAVC_BENCH_PRECISE_48=1 timeout 900 python bench.py --steps 10 --warmup 2 --no-extras --no-other-configs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=o['precise']
print('48 frames per step: precise', p.get('value'), p.get('frames_per_step'), p.get('steps'), p.get('error'))" | tee -a $O/precise_ab.txt
