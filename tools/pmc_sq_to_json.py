"""Summarise the SQ counter CSVs of tools/pmc_bench_sq.sh for the dominant kernel -> profiles/<label>_tail_conv_sq.json
   python tools/pmc_sq_to_json.py [label] [dir]"""
import collections, csv, glob, json, sys

label = sys.argv[1] if len(sys.argv) > 1 else "r3"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
KERNEL = sys.argv[3] if len(sys.argv) > 3 else "conv_pipe_kernel<2, 4, 8, 1, 0"
rec = {"kernel": KERNEL, "command": "tools/pmc_bench_sq.sh (rocprofv3 --kernel-trace --pmc ..., bench.py --steps 2 --warmup 1)", "counters": {}}
for d in ("pmc_bench_sq1", "pmc_bench_sq2"):
    cands = glob.glob(f"{src}/{d}/*/*counter_collection.csv")
    if not cands:
        sys.exit(f"no counter CSV under {src}/{d}: see {src}/{d}.log")
    trace = glob.glob(f"{src}/{d}/*/*kernel_trace.csv")
    dur = {}
    if trace:
        for r in csv.DictReader(open(trace[0])):
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg = collections.defaultdict(list)
    durs = []
    seen = set()
    for r in csv.DictReader(open(cands[0])):
        if KERNEL not in r["Kernel_Name"]:
            continue
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen and r["Dispatch_Id"] in dur:
            seen.add(r["Dispatch_Id"]); durs.append(dur[r["Dispatch_Id"]])
    if not agg:
        sys.exit(f"kernel {KERNEL} not found in {cands[0]}")
    for k, v in agg.items():
        v = v[len(v) // 2:]                     # the timed steps come last (autotune trials first)
        rec["counters"][k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
    if durs:
        durs = durs[len(durs) // 2:]
        rec["counters"].setdefault("_duration_ns_" + d, sum(durs) / len(durs))
c = rec["counters"]
def m(k): return c[k]["mean_per_launch"] if k in c else None
N_XCD, N_SIMD = 8, 1024
if m("GRBM_GUI_ACTIVE") and "_duration_ns_pmc_bench_sq2" in c:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs: per-XCD busy cycles / duration = the effective shader clock under this kernel's power draw
    rec["effective_clock_ghz"] = m("GRBM_GUI_ACTIVE") / N_XCD / c["_duration_ns_pmc_bench_sq2"]
if m("SQ_VALU_MFMA_BUSY_CYCLES") and m("GRBM_GUI_ACTIVE"):
    # SQ_VALU_MFMA_BUSY_CYCLES counts pipe cycles summed over all SIMDs (16 per v_mfma_f32_16x16x32_f16); the cycles a SIMD had = GUI_ACTIVE / 8
    rec["mfma_util_at_effective_clock"] = m("SQ_VALU_MFMA_BUSY_CYCLES") / (N_SIMD * m("GRBM_GUI_ACTIVE") / N_XCD)
    rec["mfma_instructions_16x16x32"] = m("SQ_VALU_MFMA_BUSY_CYCLES") / 16
    rec["mfma_util_at_2p4ghz"] = m("SQ_VALU_MFMA_BUSY_CYCLES") / (N_SIMD * 2.4 * c["_duration_ns_pmc_bench_sq1"])
if m("SQ_WAVE_CYCLES"):
    for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
        if m(k):
            rec[k.lower() + "_over_wave_cycles"] = m(k) / m("SQ_WAVE_CYCLES")
rec["reading"] = ("WAIT_ANY = wave parked at s_waitcnt / barrier, WAIT_INST_ANY = issue stall (MFMA dependency / pipe), ACTIVE_INST_ANY = issuing; "
                  "the three are disjoint shares of SQ_WAVE_CYCLES (MI355X_MICROARCH.md, counter table)")
json.dump(rec, open(f"profiles/{label}_tail_conv_sq.json", "w"), indent=1)
print(json.dumps(rec, indent=1))
