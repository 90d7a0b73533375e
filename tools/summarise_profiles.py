#!/usr/bin/env python3
"""Turns gpurun_out/<tag>_* (written by tools/collect_profiles.sh on the GPU box) into the files under profiles/:
<tag>_kernel_stats.csv, <tag>_kernel_trace.csv (k_align rows), <tag>_bench_under_rocprof.log,
<tag>_pmc_counters.json and <tag>_traffic.json.  Usage: tools/summarise_profiles.py r02 [pairs len band [kernel-substring]]
The commit the data was collected on is read from gpurun_out/<tag>_commit (written by whoever launched the collection)."""
import csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
pairs, length, band = (int(x) for x in (sys.argv[2:5] if len(sys.argv) >= 5 else (100000, 50000, 512)))
want_kernel = sys.argv[5] if len(sys.argv) >= 6 else None   # e.g. "k_align_p<" : the dominant kernel of this workload

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "tools"))
import srchash
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")


def one(pattern):
    hits = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)   # (a re-collected tag: the newest run)
    if not hits:
        raise SystemExit("missing " + pattern)
    return hits[-1]


shutil.copy(one(f"{tag}_trace/**/*_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
with open(one(f"{tag}_trace/**/*_kernel_trace.csv")) as f, open(os.path.join(dst, f"{tag}_kernel_trace.csv"), "w") as g:
    rd = csv.reader(f)
    wr = csv.writer(g)
    wr.writerow(next(rd))
    for row in rd:
        if any("k_align" in c and (want_kernel is None or want_kernel in c.replace(" ", "")) for c in row):
            wr.writerow(row)
with open(os.path.join(src, f"{tag}_bench.log")) as f:
    lines = [l for l in f if l.startswith("{")]
with open(os.path.join(dst, f"{tag}_bench_under_rocprof.log"), "w") as g:
    g.write(lines[-1])

counters, kernel = {}, None
for name in ("fetch", "write", "sq", "sq2", "sq3"):
    try:
        path = one(f"{tag}_{name}/**/*_counter_collection.csv")
    except SystemExit:
        continue
    with open(path) as f:
        for row in csv.DictReader(f):
            if "k_align" not in row["Kernel_Name"] or (want_kernel and want_kernel not in row["Kernel_Name"].replace(" ", "")):
                continue
            kernel = row["Kernel_Name"]
            counters[row["Counter_Name"]] = counters.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
with open(os.path.join(dst, f"{tag}_pmc_counters.json"), "w") as g:
    json.dump({"kernel": kernel, "launches": 1, "counters": counters}, g, indent=1)

# MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 counts a 128-B read request as 64 B,
# so the read side is doubled; the write side is taken as is.
fetch = counters["FETCH_SIZE"] * 1024.0 * 2.0
write = counters["WRITE_SIZE"] * 1024.0
with open(os.path.join(dst, f"{tag}_traffic.json"), "w") as g:
    commit = open(os.path.join(src, f"{tag}_commit")).read().strip() if os.path.exists(os.path.join(src, f"{tag}_commit")) else None
    # the sources the data was collected on: recorded by the collecting side (gpurun_out/<tag>_source_hash, written on the GPU
    # box from the tree it ran) -- falls back to this tree, which is the same one when nothing was edited in between
    hpath = os.path.join(src, f"{tag}_source_hash")
    shash = open(hpath).read().strip() if os.path.exists(hpath) else srchash.source_hash(root)
    json.dump({"round": int(tag[1:3]) if tag[1:3].isdigit() else None, "commit": commit, "source_hash": shash, "kernel": kernel, "workload": {"pairs_per_launch": pairs, "len": length, "band": band},
               "FETCH_SIZE_KB": counters["FETCH_SIZE"], "WRITE_SIZE_KB": counters["WRITE_SIZE"],
               "fetch_bytes_corrected": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
               "note": "separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `python3 bench.py --steps 1 --warmup 0 "
                       "--no-cpu-baseline`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B read requests "
                       "at 64 B); WRITE_SIZE taken as is"}, g, indent=1)
print(json.dumps(counters, indent=1))
print("hbm bytes per launch %.4g (fetch %.4g, write %.4g)" % (fetch + write, fetch, write))
