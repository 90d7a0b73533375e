#!/bin/bash
# The round's closing check on the GPU box (from the repo root): the whole -m gpu suite, then the default bench line with its records.
set -u
OUT=gpurun_out/${1:-final}; mkdir -p $OUT
(time python -m pytest tests -m gpu -x -q) > $OUT/gputests.log 2>&1; tail -4 $OUT/gputests.log
python bench.py > $OUT/bench.log 2> $OUT/bench.err; tail -c 600 $OUT/bench.err
python - $OUT/bench.log <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print("value %.0f GCUPS, %.2f ms per step, roofline frac %.4f, kernel %.2f ms, profile_matches_source %s, VALU/cell %s (%s)" % (
    d["value"], d["ms_per_step"], r["frac"], r["kernel_ms_per_launch"], r["profile_matches_source"], r["valu"]["insts_per_cell"], r["valu"].get("profile_matches_source")))
print("cpu_baseline", d.get("cpu_baseline"), "verified", d.get("verified_pairs"))
b = d["band150"]
print("band150 %.0f GCUPS frac %.4f traffic/alg %s matches %s VALU/cell %s" % (b["gcups"], b["roofline_frac"], b["traffic_over_algorithmic"], b["profile_matches_source"], b["valu"]["insts_per_cell"]))
for k in ("strong8_proxy", "strong4_proxy"):
    print(k, {x: d[k][x] for x in d[k] if "gcups" in x or "factor" in x})
m = d["mixed150"]
print("mixed150 %.0f GCUPS, %.2f ms per step, %.2f ms kernels, packed-top share %s, verified %s (%s)" % (m["gcups"], m["ms_per_step"], m["kernel_ms_per_step"], m["packed_top_share"], m.get("verified_calls"), m.get("verified_against")))
l = d["l1"]
print("l1 %.3f ms per call, %.1f GCUPS, frac %.5f" % (l["ms_per_step"], l["gcups"], l["roofline_frac"]), l["counters"], l["cpu_baseline"])
PY
