#!/bin/bash
# MFMA-busy / wait counters of the dominant kernel on the bench command itself (own --pmc passes, kernel-trace only).
# Output: gpurun_out/pmc_bench_sq{1,2}/ ; $1 = extra bench args
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_bench_sq1 -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-precise --steps 2 --warmup 1 $1 > $R/gpurun_out/pmc_bench_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_bench_sq2 -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-precise --steps 2 --warmup 1 $1 > $R/gpurun_out/pmc_bench_sq2.log 2>&1
cd $R
