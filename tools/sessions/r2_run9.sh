#!/bin/bash
# pitch experiment: power-of-two pixel pitches vs an odd number of 128-byte lines per pixel (HAVC_PITCH_ODD)
O=gpurun_out/r2i; mkdir -p $O
for P in 0 1; do
  echo "== pitch_odd $P" >> $O/pitch.txt
  HAVC_PITCH_ODD=$P python tools/conv_bench.py 16 7 tail256,l6conv,l5conv,l4conv,mid,l8blur,l7blur,l8nops,enc,e3,e4,e2,e1,l5ps,l4ps >> $O/pitch.txt 2>&1
done
for P in 0 1; do
  echo "== pitch_odd $P" >> $O/pitch_bench.txt
  HAVC_PITCH_ODD=$P python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 >> $O/pitch_bench.txt 2>&1
  HAVC_PITCH_ODD=$P python tools/gpu_profile.py wide 560 16 > $O/perop_pitch$P.txt 2>&1
done
paste <(grep -A60 "pitch_odd 0" $O/pitch.txt | grep cfg= | head -40) <(grep -A60 "pitch_odd 1" $O/pitch.txt | grep cfg= | awk '{print $4,$5,$6,$7}' | head -40)
grep -o '"value": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"whole_path_tflops": [0-9.]*\|== pitch.*\|Error.*' $O/pitch_bench.txt
grep "whole pass\|^wide" $O/perop_pitch*.txt
