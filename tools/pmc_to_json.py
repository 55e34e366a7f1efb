"""Turn the counter CSVs of tools/pmc_bench.sh (gpurun_out/pmc_bench_{fetch,write}) into the per-launch HBM traffic record that
bench.py reports as roofline.traffic (profiles/r2_tail_conv_pmc.json), and copy the CSVs next to it.
   python tools/pmc_to_json.py [frames_per_launch] [label]"""
import csv, glob, json, os, shutil, sys

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
label = sys.argv[2] if len(sys.argv) > 2 else "r2"
KERNEL = "conv_pipe_kernel<2, 4, 8, 1, 0>"


def per_launch(kind, counter):
    f = glob.glob(f"gpurun_out/pmc_bench_{kind}/*/*counter_collection.csv")[0]
    os.makedirs("profiles/pmc", exist_ok=True)
    shutil.copy(f, f"profiles/pmc/{label}_bench_{kind}_size_counter_collection.csv")
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return vals

fetch, write = per_launch("fetch", "FETCH_SIZE"), per_launch("write", "WRITE_SIZE")
favg, wavg = sum(fetch) / len(fetch), sum(write) / len(write)
rec = {
    "kernel": KERNEL + " (layers.10 3x3 259->259 @560x560; the second launch of a pass also runs layers.11/12 in its epilogue)",
    "frames_per_launch": frames,
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 (tools/pmc_bench.sh)",
    "launches": {"fetch_pass": len(fetch), "write_pass": len(write)},
    "FETCH_SIZE_kb_raw_avg": favg, "WRITE_SIZE_kb_raw_avg": wavg,
    "correction": "counter unit KiB (x1024; round 1 used x1000); gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) reads -> x2 (MI355X_MICROARCH.md HBM section); WRITE_SIZE is taken 1:1",
    "fetch_bytes_corrected": favg * 1024.0 * 2.0, "write_bytes": wavg * 1024.0,
    "traffic_bytes_per_launch": favg * 1024.0 * 2.0 + wavg * 1024.0,
    "algorithmic_bytes_per_launch_avg": "conv1: 2.65 GB in + 2.65 GB out; conv2 (fused with the final 1x1 conv): 2.65 GB in + 2.65 GB residual + 15 MB RGB8 out -> avg 5.31 GB",
}
json.dump(rec, open(f"profiles/{label}_tail_conv_pmc.json", "w"), indent=1)
print(json.dumps(rec, indent=1))
