#!/bin/bash
# round-5 session 15: host profile of the c5 frame loop (cProfile over the timed steps)
R=$PWD; O=$R/gpurun_out/r5o; mkdir -p $O
timeout 600 python tools/c5_host_profile.py 6 > $O/c5_host_profile.json 2> $O/c5_host_profile.txt
grep -v amdgpu $O/c5_host_profile.txt | head -120 | cut -c1-170
