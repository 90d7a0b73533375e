#!/usr/bin/env python3
"""gpurun_out/<tag>_l1_{2p9mb,30mb}_* (tools/collect_l1_profiles.sh) -> profiles/<tag>_l1_<w>_{kernel_stats.csv,
bench_under_rocprof.log, pmc_counters.json}.  The counters are those of the chain kernel (k_chain2), summed over the calls of the
pass (two: one warm-up, one timed) and divided by their number."""
import csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
commit = open(os.path.join(src, f"{tag}_commit")).read().strip() if os.path.exists(os.path.join(src, f"{tag}_commit")) else None
hpath = os.path.join(src, f"{tag}_l1_source_hash")
shash = open(hpath).read().strip() if os.path.exists(hpath) else None
for w in ("2p9mb", "30mb"):
    shutil.copy(os.path.join(src, f"{tag}_l1_{w}_kernel_stats.csv"), os.path.join(dst, f"{tag}_l1_{w}_kernel_stats.csv"))
    with open(os.path.join(src, f"{tag}_l1_{w}_bench.log")) as f:
        lines = [l for l in f if l.startswith("{")]
    with open(os.path.join(dst, f"{tag}_l1_{w}_bench_under_rocprof.log"), "w") as g:
        g.write(lines[-1])
    bench = json.loads(lines[-1])
    counters, launches = {}, {}
    for name in ("fetch", "write", "sq"):
        hits = sorted(glob.glob(os.path.join(src, f"{tag}_l1_{w}_{name}/**/*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
        if not hits:
            continue
        with open(hits[-1]) as f:
            for row in csv.DictReader(f):
                if "k_chain2" not in row["Kernel_Name"]:
                    continue
                counters[row["Counter_Name"]] = counters.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                launches.setdefault(row["Counter_Name"], set()).add(row.get("Dispatch_Id", row.get("Correlation_Id", "")))
    per_call = {k: v / max(1, len(launches[k])) for k, v in counters.items()}
    rec = {"commit": commit, "source_hash": shash, "kernel": "k_chain2", "workload": bench["workload"], "cells_per_call": bench["cells"],
           "counters_per_launch": per_call, "launches_seen": {k: len(v) for k, v in launches.items()}}
    if "FETCH_SIZE" in per_call and "WRITE_SIZE" in per_call:
        # MI355X_MICROARCH.md: KiB units; gfx950 counts a 128-B read request as 64 B (read side doubled), writes as they are
        rec["hbm_bytes_per_launch"] = per_call["FETCH_SIZE"] * 1024.0 * 2.0 + per_call["WRITE_SIZE"] * 1024.0
        rec["traffic_over_algorithmic"] = rec["hbm_bytes_per_launch"] / (bench["cells"] * 0.2507)
    if "SQ_INSTS_VALU" in per_call:
        rec["valu_insts_per_cell"] = per_call["SQ_INSTS_VALU"] * 64.0 / bench["cells"]
    with open(os.path.join(dst, f"{tag}_l1_{w}_pmc_counters.json"), "w") as g:
        json.dump(rec, g, indent=1)
    print(w, json.dumps(rec)[:600])
