#!/usr/bin/env python3
"""Summary of tools/ab_r05.sh: per variant GCUPS, kernel ms, launch info and the FETCH_SIZE / WRITE_SIZE of its DP kernels."""
import csv, glob, json, os, sys
out, variants = sys.argv[1], sys.argv[2:]
for v in variants:
    rec = None
    try:
        for l in open(os.path.join(out, v + "_bench.log")):
            if l.startswith("{"):
                rec = json.loads(l)
    except OSError:
        pass
    cnt = {}
    for name in ("fetch", "write"):
        for path in glob.glob(os.path.join(out, "%s_%s" % (v, name), "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                if "k_align" in r["Kernel_Name"]:
                    cnt[r["Counter_Name"]] = cnt.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    f, w = cnt.get("FETCH_SIZE"), cnt.get("WRITE_SIZE")
    line = "%-14s" % v
    if rec:
        rf = rec["roofline"]
        line += " %8.0f GCUPS  step %8.2f ms  kernel %8.2f ms  %s" % (rec["value"], rec["ms_per_step"], rf["kernel_ms_per_launch"], rf.get("kernel"))
    else:
        line += " (no bench line)"
    if f is not None and w is not None:
        line += "  | FETCH x2 %.4f TB  WRITE %.4f TB  sum %.4f TB" % (f * 2048 / 1e12, w * 1024 / 1e12, (f * 2048 + w * 1024) / 1e12)
        if rec:
            line += " = %.2f x algorithmic" % ((f * 2048 + w * 1024) / rec["roofline"]["algorithmic_bytes_per_launch"])
    print(line)
    if rec and rec.get("launch_info"):
        for li in rec["launch_info"]:
            print("      ", json.dumps(li))
