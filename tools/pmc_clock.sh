#!/bin/bash
# effective shader clock per kernel variant = GRBM_GUI_ACTIVE / kernel duration (MI355X_MICROARCH.md "DVFS give-back").  $1 shape $2 cfgs
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_clock -- python3 $R/tools/conv_bench.py 8 3 $1 $2 > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_clock/*/*counter_collection.csv")[0]
t = glob.glob("gpurun_out/pmc_clock/*/*kernel_trace.csv")[0]
dur = {}
for r in csv.DictReader(open(t)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    d, name = dur[r["Dispatch_Id"]]
    if "conv_" not in name: continue
    agg[name[:70]].append((float(r["Counter_Value"]), d))
for k, v in agg.items():
    v = v[len(v)//2:]
    cyc = sum(a for a, _ in v) / len(v); ns = sum(b for _, b in v) / len(v)
    print(f"{k:72s} {ns/1e6:7.3f} ms  GUI_ACTIVE {cyc:.4g}  -> {cyc/ns:.3f} GHz")
PY
