#!/bin/bash
O=gpurun_out/r3i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_havc_harness.py tests/test_colormnet_net.py -m gpu -q -x --tb=short -k "split_k or pre_tweak or render_matches or network_modules" 2>&1 | tail -6 > $O/pytest.txt; cat $O/pytest.txt
python tools/conv_bench.py 16 7 tail256,l7conv,l6conv,l5conv,mid0,mid1,l8nops,l7nops 60,100,60 > $O/sched_other.txt 2>&1; cat $O/sched_other.txt
python tools/conv_bench.py 16 7 tail259 61,101,61,101 >> $O/sched_other.txt 2>&1; tail -4 $O/sched_other.txt
for sk in 1 0; do echo "== HAVC_CMN_SPLITK=$sk" >> $O/splitk.txt; HAVC_CMN_SPLITK=$sk timeout 600 python -u tools/colormnet_clip_bench.py 40 216 384 >> $O/splitk.txt 2>&1; done
grep -E "==|frames 216|slice|sum of" $O/splitk.txt
