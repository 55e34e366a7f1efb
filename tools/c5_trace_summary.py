"""Per-kernel summary of the last N frames of a rocprofv3 --kernel-trace run of `bench.py --config c5` (every ColorMNet frame ends with one
cmn_lab_to_rgb / cmn_frame_out launch): time per frame, launches per frame, average duration.   python tools/c5_trace_summary.py <kernel_trace.csv> [frames]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(("cmn_lab_to_rgb", "cmn_frame_out"))]      # (round 4: the fast step ends a frame with cmn_frame_out)
if not idx:
    sys.exit("c5_trace_summary: no frame-ending kernel (cmn_lab_to_rgb / cmn_frame_out) in the trace")
first = idx[-N - 1] + 1 if len(idx) > N else 0
sel = rows[first:idx[-1] + 1]
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
agg = collections.OrderedDict()
for r in sel:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(r["Kernel_Name"][:110], [0, 0])
    a[0] += 1
    a[1] += d
tot = sum(a[1] for a in agg.values())
print(f"# last {N} frames: {len(sel)} launches = {len(sel) / N:.1f} per frame, kernel time {tot / 1e6:.2f} ms = {tot / 1e6 / N:.3f} ms per frame, "
      f"wall {(t1 - t0) / 1e6:.2f} ms = {(t1 - t0) / 1e6 / N:.3f} ms per frame; source {sys.argv[1]}")
print("# us/frame  launches/frame  avg us  kernel")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{a[1] / 1e3 / N:9.1f} {a[0] / N:9.2f} {a[1] / a[0] / 1e3:9.1f}  {k}")
