#!/bin/bash
# HBM traffic of the dominant kernel of `bench.py --config $1` (c3 | c4) from PMC counters: separate --pmc passes with --kernel-trace only, as
# MI355X_MICROARCH.md's HBM section prescribes.  Output: gpurun_out/pmc_$1_{fetch,write}/ ; summarise with tools/pmc_cfg_to_json.py $1
CFG=${1:-c3}
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_${CFG}_fetch -- python3 $R/bench.py --config $CFG --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $R/gpurun_out/pmc_${CFG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_${CFG}_write -- python3 $R/bench.py --config $CFG --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $R/gpurun_out/pmc_${CFG}_write.log 2>&1
cd $R
