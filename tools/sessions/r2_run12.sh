#!/bin/bash
O=gpurun_out/r2l; mkdir -p $O
python -m pytest tests/test_ddcolor.py -m gpu -q -x -k fused 2>&1 | tail -8 > $O/pytest_dwln.txt
cat $O/pytest_dwln.txt
python tools/dwln_bench.py 16 > $O/dwln_bench.txt 2>&1; cat $O/dwln_bench.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES -d $GRAFT_REPO_ROOT/$O/pmc1 -o dwln --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dwln_bench.py 16 1 > $GRAFT_REPO_ROOT/$O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $GRAFT_REPO_ROOT/$O/pmc2 -o dwln --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dwln_bench.py 16 1 > $GRAFT_REPO_ROOT/$O/pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for d in ("pmc1", "pmc2"):
    for f in glob.glob(f"gpurun_out/r2l/{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "dwconv7_ln" not in k: continue
            agg[(k[:40], row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in agg.items():
            print(d, k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
