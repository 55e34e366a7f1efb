#!/bin/bash
# Memory-path PMC passes on the conv micro-benchmark (each --pmc set in its own run, kernel-trace only).
# $1 = shape, $2 = cfg list, $3 = batch
R=$PWD; cd /tmp; export TMPDIR=/tmp
B=${3:-8}
run() { name=$1; shift; timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcm_$name -- python3 $R/tools/conv_bench.py $B 2 $SHAPE $CFG > /dev/null 2>&1; }
SHAPE=$1; CFG=$2
[ -z "$SKIP_TCP" ] && run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
run ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum GRBM_GUI_ACTIVE
run tcc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_avr
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmcm_*")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_pipe" not in r["Kernel_Name"]: continue
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        print(d.split("/")[-1], k, {c: "%.4g" % (x / cnt[(k, c)]) for c, x in v.items()})
PY
