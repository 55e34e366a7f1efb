"""Per-kernel summary of the TIMED steps of a rocprofv3 --kernel-trace run of bench.py (the raw --stats file also counts the
autotuner's trial launches and the warm-up).  A bench step starts with resize_h_kernel / resize_h_rows_kernel (Spline64 squash) and holds two of them
(down, up): kernels from the (2 * warmup)-th resize_h launch on are the timed region.
   python tools/trace_summary.py <kernel_trace.csv> <warmup> > profiles/..."""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "resize_h_kernel" in r["Kernel_Name"] or "resize_h_rows_kernel" in r["Kernel_Name"]]
first = starts[2 * warm] if len(starts) > 2 * warm else 0
sel = rows[first:]
agg = collections.OrderedDict()
for r in sel:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(r["Kernel_Name"], [0, 0, 1 << 62, 0])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
steps = (len(starts) - 2 * warm) / 2
print(f"# timed region: {len(sel)} launches, {steps:g} steps, kernel time {tot / 1e6:.2f} ms ({tot / 1e6 / max(steps, 1):.3f} ms per step); source {sys.argv[1]}")
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'"{k}",{a[0]},{a[1]},{a[1] / a[0]:.1f},{100 * a[1] / tot:.2f},{a[2]},{a[3]}')
