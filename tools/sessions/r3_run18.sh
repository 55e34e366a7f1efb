#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3s; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
HAVC_CMN_LOOKAHEAD=16 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -- python3 $R/bench.py --config c5 --no-cpu-baseline --no-extras --steps 4 --warmup 2 > $O/bench_c5_under_rocprof.json 2> $O/bench_c5_under_rocprof.err
cd $R
# timed region = the last 4 of 6 steps: take the last 4/6 of the trace by time as an approximation -> per-kernel totals
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r3s/prof_c5/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps are delimited by bursts of resize_h_kernel (32 squashes per step come 16 at a time with look-ahead); use the last 128 frames = last 4 steps
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("cmn_lab_to_rgb")]
first = idx[-128] if len(idx) >= 128 else 0
sel = rows[first:]
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
agg = collections.OrderedDict()
for r in sel:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(r["Kernel_Name"][:90], [0, 0])
    a[0] += 1; a[1] += d
tot = sum(a[1] for a in agg.values())
print(f"# last 128 frames: {len(sel)} launches, kernel time {tot/1e6:.2f} ms = {tot/1e6/128:.3f} ms per frame, wall {((t1-t0)/1e6):.2f} ms = {((t1-t0)/1e6/128):.3f} ms per frame")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{a[1]/1e6/128*1000:8.1f} us/frame  {a[0]/128:7.2f} launches/frame  {a[1]/a[0]/1e3:8.1f} us avg  {k}")
PY
rm -rf $O/prof_c5
