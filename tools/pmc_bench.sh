#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters, on the bench command itself (separate --pmc passes,
# kernel-trace only, as MI355X_MICROARCH.md §HBM prescribes).  Output: gpurun_out/pmc_bench_{fetch,write}/
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_bench_fetch -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-precise --steps 2 --warmup 1 > $R/gpurun_out/pmc_bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_bench_write -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-precise --steps 2 --warmup 1 > $R/gpurun_out/pmc_bench_write.log 2>&1
cd $R
