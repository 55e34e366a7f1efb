#!/bin/bash
# SQ counters of the fused depthwise-7x7 + LayerNorm kernel (tools/dwln_bench.py at 64 frames): where a wave's cycles go.
# Output: gpurun_out/pmc_dwln{1,2}/
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmc_dwln1 -- python3 $R/tools/dwln_bench.py 64 3 > $R/gpurun_out/pmc_dwln1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/pmc_dwln2 -- python3 $R/tools/dwln_bench.py 64 3 > $R/gpurun_out/pmc_dwln2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc_dwln1", "pmc_dwln2"):
    for f in glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "dwconv7_ln" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in agg.items():
            print(d, k)
            for n, v in c.items(): print(f"   {n:28s} mean {sum(v)/len(v):.4g}  launches {len(v)}")
PY
