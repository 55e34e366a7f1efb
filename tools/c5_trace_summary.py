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

# ---- per-stream timeline (round 5): which stream carries the frame chain, how busy it is, what one steady frame looks like on it ----
by = collections.defaultdict(list)
for r in sel:
    by[(r.get("Queue_Id", "?"), r.get("Stream_Id", "?"))].append(r)
print("# per stream (queue, stream): launches/frame, busy us/frame, gaps (us between a kernel's end and the next start on the same stream) p50 / p90, share of the wall busy")
for k, rs in sorted(by.items(), key=lambda kv: -len(kv[1])):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    gaps = sorted(max(0, int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rs, rs[1:]))
    g50, g90 = (gaps[len(gaps) // 2], gaps[len(gaps) * 9 // 10]) if gaps else (0, 0)
    print(f"#   {k}: {len(rs) / N:6.1f} launches/frame, {busy / 1e3 / N:8.1f} us/frame busy, gaps p50 {g50:.1f} p90 {g90:.1f} us, {100 * busy / (t1 - t0):.1f} % of the wall")
if len(sys.argv) > 3:                                                     # dump of frames [a, b) counted from the end of the trace: every launch with its stream
    a, b = (int(x) for x in sys.argv[3].split(":"))
    lo, hi = idx[-a - 1] + 1, idx[-b - 1] + 1 if b > 0 else idx[-1] + 1
    base = int(rows[lo]["Start_Timestamp"])
    print(f"# launches of the frames {a} .. {b} before the end: start us (from the first), duration us, stream, kernel")
    for r in rows[lo:hi]:
        print(f"{(int(r['Start_Timestamp']) - base) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f}  s{r.get('Stream_Id', '?')}  {r['Kernel_Name'][:90]}")

# ---- per-frame table: the period between two frame-ending kernels, how much of it the chain's stream worked, its longest idle gap, and how busy the other streams were meanwhile ----
ends = [(int(rows[i]["End_Timestamp"]), rows[i].get("Stream_Id", "?")) for i in idx[-min(len(idx), 50):]]
main = ends[-1][1]
evs = collections.defaultdict(list)
for r in rows[idx[-min(len(idx), 50)]:idx[-1] + 1]:
    evs[r.get("Stream_Id", "?")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
def busy_in(lst, a, b):
    return sum(max(0, min(e, b) - max(s0, a)) for s0, e, _ in lst)
print(f"# per frame (chain stream s{main}): period us, chain busy us, longest chain idle gap us (kernel after it), busy us of the other streams " + " ".join(f"s{k}" for k in sorted(evs) if k != main))
for (a, _), (b, _) in zip(ends, ends[1:]):
    ch = [e for e in evs[main] if a <= e[0] < b]
    gap, after, prev = 0, "", a
    for s0, e, nm in ch:
        if s0 - prev > gap:
            gap, after = s0 - prev, nm[:40]
        prev = max(prev, e)
    others = " ".join(f"{busy_in(evs[k], a, b) / 1e3:7.1f}" for k in sorted(evs) if k != main)
    print(f"{(b - a) / 1e3:9.1f} {busy_in(ch, a, b) / 1e3:8.1f} {gap / 1e3:8.1f}  {others}   {after}")
