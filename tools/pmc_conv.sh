#!/bin/bash
# PMC passes on the conv micro-benchmark (rocprofv3 --pmc in its own run, kernel-trace only).  $1 = shape, $2 = cfg list
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq_$1 -- python3 $R/tools/conv_bench.py 4 2 $1 $2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_tcc_$1 -- python3 $R/tools/conv_bench.py 4 2 $1 $2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$1 -- python3 $R/tools/conv_bench.py 4 2 $1 $2 > /dev/null 2>&1
cd $R
