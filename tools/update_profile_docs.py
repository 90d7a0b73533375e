#!/usr/bin/env python3
"""Re-writes the numbers of the Round-5 section of profiles/README.md from the summarised sets under profiles/ (after
tools/summarise_profiles.py / summarise_l1_profiles.py):   python3 tools/update_profile_docs.py"""
import csv, json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")

def load(tag):
    c = json.load(open(os.path.join(P, tag + "_pmc_counters.json")))["counters"]
    t = json.load(open(os.path.join(P, tag + "_traffic.json")))
    b = [json.loads(l) for l in open(os.path.join(P, tag + "_bench_under_rocprof.log")) if l.startswith("{")][-1]
    avg = None
    for row in csv.reader(open(os.path.join(P, tag + "_kernel_stats.csv"))):
        if row and row[0].startswith("void gamdp") and t["kernel"].split("(")[0] in row[0]:
            avg = float(row[3]) / 1e6
            break
    alg = b["roofline"]["algorithmic_bytes_per_launch"]
    return dict(c=c, t=t, b=b, avg=avg, cells=alg / 0.2507, alg=alg,
                valu=c["SQ_INSTS_VALU"] * 64 / (alg / 0.2507), busy=(c["SQ_ACTIVE_INST_VALU"] / 1024) / (c["SQ_WAVE_CYCLES"] / 4096),
                wait=c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"])

def sp(x):   # 14412 -> "14 412"
    s = "%d" % round(x)
    return s if len(s) < 4 else s[:-3] + " " + s[-3:]

h, b150, s5 = load("r05"), load("r05_band150"), load("r05_band150_5kb")
path = os.path.join(P, "README.md")
s = open(path).read()
a = s.index("## Round 5 (commit")
e = s.index("## Round 4 (commit")
sec = s[a:e]
sec = re.sub(r"## Round 5 \(commit \w+, source hash \w+;", "## Round 5 (commit %s, source hash %s;" % (h["t"]["commit"], h["t"]["source_hash"]), sec)
sec = re.sub(r"`k_align_p<17,4>` 4 launches, average \*\*[\d.]+ ms\*\*", "`k_align_p<17,4>` 4 launches, average **%.2f ms**" % h["avg"], sec)
sec = re.sub(r"\(\d+ \d+ GCUPS, kernel [\d.]+ ms by HIP events\)", "(%s GCUPS, kernel %.2f ms by HIP events)" % (sp(h["b"]["value"]), h["b"]["roofline"]["kernel_ms_per_launch"]), sec)
sec = re.sub(r"kernel `k_align_o<19,15>`: average \*\*[\d.]+ ms\*\* \(4 launches; bench.py's own line of that run: [\d ]+ GCUPS;",
             "kernel `k_align_o<19,15>`: average **%.2f ms** (4 launches; bench.py's own line of that run: %s GCUPS;" % (b150["avg"], sp(b150["b"]["value"])), sec)
sec = re.sub(r"`k_align_o<19,15>` \*\*[\d.]+ ms\*\* \([\d ]+ GCUPS whole step\)", "`k_align_o<19,15>` **%.2f ms** (%s GCUPS whole step)" % (s5["avg"], sp(s5["b"]["value"])), sec)
i = sec.index("Reading the counters.")
j = sec.index("What the calibration says")
para = ("Reading the counters.  **Headline** (%.3fe12 cell updates per launch in %.4f s): `roofline.achieved` = %s GB/s = %.1f %% of 8 TB/s\n"
        "(the round's other boxes: 44.7 - 46.2 %%); HBM traffic %.3f TB per launch = **%.2f x** the contract figure (%.3f written, %.3f read);\n"
        "`SQ_INSTS_VALU` x 64 / cells = **%.2f** vector instructions per cell update (round 4: 2.50), the vector pipe busy %.1f %%\n"
        "(`SQ_ACTIVE_INST_VALU` / 1 024 against `SQ_WAVE_CYCLES` / 4 096).  **Band 150** (%.3fe12 cell updates in %.4f s): `roofline_frac`\n"
        "%.3f here (0.305 - 0.324 over the boxes); HBM traffic **%.3f TB = %.2f x** the contract figure (%.3f read, %.3f written; round 4:\n"
        "0.773 TB = 2.05 x, 0.412 / 0.361); **%.2f** vector instructions per cell (2.95), the vector pipe busy **%.1f %%** (88 %%; 93 - 95.5 %% over\n"
        "the collections of the round, same kernels), `SQ_WAIT_ANY` %.0f %% of the wave-cycles (25 %%).  **Band 150, 5 kb calls** (%.3fe11 cell\n"
        "updates in %.1f ms: %s GCUPS inside the kernel): **%.2f** vector instructions per cell -- the top / end / ramp blocks and the side\n"
        "captures of 131 072 short tasks --, the vector pipe busy %.0f %%, `SQ_WAIT_ANY` %.0f %%; HBM traffic %.1f GB = %.2f x.  " % (
            h["cells"] / 1e12, h["avg"] / 1e3, sp(h["alg"] / (h["avg"] / 1e3) / 1e9), 100 * h["alg"] / (h["avg"] / 1e3) / 8e12,
            h["t"]["hbm_bytes_per_launch"] / 1e12, h["t"]["hbm_bytes_per_launch"] / h["alg"], h["t"]["write_bytes"] / 1e12, h["t"]["fetch_bytes_corrected"] / 1e12,
            h["valu"], 100 * h["busy"],
            b150["cells"] / 1e12, b150["avg"] / 1e3, b150["alg"] / (b150["avg"] / 1e3) / 8e12,
            b150["t"]["hbm_bytes_per_launch"] / 1e12, b150["t"]["hbm_bytes_per_launch"] / b150["alg"], b150["t"]["fetch_bytes_corrected"] / 1e12, b150["t"]["write_bytes"] / 1e12,
            b150["valu"], 100 * b150["busy"], 100 * b150["wait"],
            s5["cells"] / 1e11, s5["avg"], sp(s5["cells"] / (s5["avg"] / 1e3) / 1e9), s5["valu"], 100 * s5["busy"], 100 * s5["wait"],
            s5["t"]["hbm_bytes_per_launch"] / 1e9, s5["t"]["hbm_bytes_per_launch"] / s5["alg"]))
sec = sec[:i] + para + sec[j:]
open(path, "w").write(s[:a] + sec + s[e:])
print("headline %.2f ms %s GCUPS; band 150 %.2f ms %s GCUPS (traffic x%.3f, busy %.3f); 5 kb %.2f ms" % (
    h["avg"], sp(h["b"]["value"]), b150["avg"], sp(b150["b"]["value"]), b150["t"]["hbm_bytes_per_launch"] / b150["alg"], b150["busy"], s5["avg"]))
