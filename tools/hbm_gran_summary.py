#!/usr/bin/env python3
"""counter / payload per kernel of tools/hbm_gran (see tools/hbm_gran.sh)."""
import csv, glob, os, re, sys
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/hbm_gran"
order, payload = [], {}
for line in open(os.path.join(out, "plain.log")):
    m = re.match(r"(\S+)\s+payload ([0-9.]+) GiB\s+([0-9.]+) ms", line)
    if m:
        order.append(m.group(1)); payload[m.group(1)] = (float(m.group(2)) * 2**30, float(m.group(3)))
vals = {}
for path in glob.glob(os.path.join(out, "*", "**", "*_counter_collection.csv"), recursive=True):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    per = {}
    for r in rows:
        per.setdefault(r["Counter_Name"], []).append((int(r.get("Dispatch_Id", 0)), r["Kernel_Name"], float(r["Counter_Value"])))
    for cname, lst in per.items():
        lst = [x for x in lst if "k_pieces" in x[1] or "k_dwords" in x[1] or "k_rows80" in x[1] or "k_line_by_lane" in x[1]]
        for (k, x) in zip(order, lst):
            vals.setdefault(k, {})[cname] = x[2]
names = sorted({c for v in vals.values() for c in v})
print("%-34s %9s %8s " % ("kernel", "payloadMB", "ms") + " ".join("%22s" % n for n in names))
for k in order:
    pb, ms = payload[k]
    cells = []
    for n in names:
        v = vals.get(k, {}).get(n)
        if v is None: cells.append("%22s" % "-")
        elif n in ("FETCH_SIZE", "WRITE_SIZE"): cells.append("%13.1f KB %6.3fx" % (v, v * 1024 / pb))
        else: cells.append("%13.0f %5.1f B/r" % (v, pb / v if v else 0))
    print("%-34s %9.1f %8.3f " % (k, pb / 1e6, ms) + " ".join(cells))
