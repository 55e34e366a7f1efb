"""Turn the counter CSVs of tools/pmc_bench.sh into the per-launch HBM traffic record that bench.py reports as roofline.traffic
(profiles/<label>_tail_conv_pmc.json), and copy the CSVs next to it.
   python tools/pmc_to_json.py [frames_per_launch] [label] [dir holding pmc_bench_fetch/ pmc_bench_write/ or the two CSVs]
The dominant kernel runs twice per generator pass: conv1 (plain store of r1) and conv2 (residual read, fused RGB8 output).  The
autotuner's trial launches (conv1 only) precede the steps in the trace, so the two kinds are told apart by their counter values
and the reported figure is the mean of (mean conv1, mean conv2) = the average launch of a pass."""
import csv, glob, json, os, shutil, sys

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
label = sys.argv[2] if len(sys.argv) > 2 else "r2"
src = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out"
KERNEL = "conv_pipe_kernel<2, 4, 8, 1, 0"       # <2,4,8,1,0,EF>: conv1 (EF = RELU) and conv2 (EF = RELU | RESIDUAL | FUSE_RGB8) are two instantiations since round 4


def per_launch(kind, counter):
    cands = glob.glob(f"{src}/pmc_bench_{kind}/*/*counter_collection.csv") + glob.glob(f"{src}/*{kind}_size_counter_collection.csv")
    if not cands:
        sys.exit(f"pmc_to_json: no {kind} counter CSV under {src} (did the rocprofv3 pass of tools/pmc_bench.sh fail? see {src}/pmc_bench_{kind}.log)")
    f = cands[0]
    os.makedirs("profiles/pmc", exist_ok=True)
    dst = f"profiles/pmc/{label}_bench_{kind}_size_counter_collection.csv"
    if os.path.abspath(f) != os.path.abspath(dst):
        shutil.copy(f, dst)
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter]


def split(vals):
    """two clusters (conv1 / conv2) around the midpoint of the extremes"""
    if not vals:
        sys.exit(f"pmc_to_json: kernel {KERNEL} has no rows in the counter CSV")
    mid = (min(vals) + max(vals)) / 2
    lo, hi = [v for v in vals if v < mid], [v for v in vals if v >= mid]
    if not lo or not hi:
        sys.exit("pmc_to_json: the launches do not fall into two clusters (conv1 / conv2): trace of another command?")
    return (sum(lo) / len(lo), len(lo)), (sum(hi) / len(hi), len(hi))


fetch, write = per_launch("fetch", "FETCH_SIZE"), per_launch("write", "WRITE_SIZE")
(f1, nf1), (f2, nf2) = split(fetch)              # conv1 fetches less (no residual)
(w2, nw2), (w1, nw1) = split(write)              # conv2 writes less (RGB8 only)
KIB = 1024.0
rec = {
    "kernel": KERNEL + " (layers.10 3x3 259->259 @560x560; the second launch of a pass also runs layers.11/12 in its epilogue)",
    "frames_per_launch": frames,
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 (tools/pmc_bench.sh)",
    "correction": "counter unit KiB (x1024; round 1 used x1000); gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) reads -> x2 (MI355X_MICROARCH.md HBM section); WRITE_SIZE is taken 1:1",
    "conv1": {"launches_seen": [nf1, nw1], "FETCH_SIZE_kib_raw": f1, "WRITE_SIZE_kib_raw": w1, "fetch_bytes": f1 * KIB * 2, "write_bytes": w1 * KIB,
              "algorithmic_bytes": "2.65 GB in + 2.65 GB out"},
    "conv2": {"launches_seen": [nf2, nw2], "FETCH_SIZE_kib_raw": f2, "WRITE_SIZE_kib_raw": w2, "fetch_bytes": f2 * KIB * 2, "write_bytes": w2 * KIB,
              "algorithmic_bytes": "2.65 GB in + 2.65 GB residual + 15 MB RGB8 out"},
    "fetch_bytes_corrected": (f1 + f2) / 2 * KIB * 2, "write_bytes": (w1 + w2) / 2 * KIB,
    "traffic_bytes_per_launch": (f1 + f2) / 2 * KIB * 2 + (w1 + w2) / 2 * KIB,
    "algorithmic_bytes_per_launch_avg": 5.31e9,
}
json.dump(rec, open(f"profiles/{label}_tail_conv_pmc.json", "w"), indent=1)
print(json.dumps(rec, indent=1))
