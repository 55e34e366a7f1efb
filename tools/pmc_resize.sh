#!/bin/bash
# SQ counters of the Spline64 resize kernels inside the c3 bench (64 frames per step)
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_resize1 -- python3 $R/bench.py --config c3 --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/pmc_resize1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_resize2 -- python3 $R/bench.py --config c3 --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/pmc_resize2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc_resize1", "pmc_resize2"):
    for f in glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "resize_" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in agg.items():
            print(d, k)
            for n, v in c.items(): print(f"   {n:28s} mean {sum(v)/len(v):.4g}  launches {len(v)}")
PY
