#!/bin/bash
# round 4 soak (GPU box, from the repo root): the 30 Mb merge-block call 500 times (and 200 each with other host splits), every call's
# output bytes hashed; a long parity campaign on the final kernels
set -u
OUT=gpurun_out/${1:-r04soak}; mkdir -p $OUT
for v in default "GAMDP_L1_COHORTS=3" "GAMDP_L1_COHORTS=16" "GAMDP_L1_NO_TWINS=1"; do
  n=500; [ "$v" = default ] || n=200
  if [ "$v" = default ]; then python tests/test_gpu_l1_stress.py 30000000 $n > $OUT/l1_default.log 2>&1; f=$OUT/l1_default.log
  else env $v python tests/test_gpu_l1_stress.py 30000000 $n > $OUT/l1_$v.log 2>&1; f=$OUT/l1_$v.log; fi
  python - "$f" "$v" <<'PY'
import sys
lines = [l.split() for l in open(sys.argv[1]) if l[:1].isdigit()]
digs = {d for rc, d in lines}
print("%-22s %d calls, return codes %s, %d distinct digest(s): %s" % (sys.argv[2], len(lines), sorted({rc for rc, d in lines}), len(digs), sorted(digs)[0][:16] if digs else None))
PY
done
(time python tools/parity_campaign.py --seeds 200 --per-seed 1000 --long 1500 --first-seed 1000) > $OUT/campaign_default.log 2>&1; tail -n 5 $OUT/campaign_default.log | head -2
(time GAMDP_QUAD_MIN=1 python tools/parity_campaign.py --seeds 100 --per-seed 1000 --long 1500 --first-seed 1200) > $OUT/campaign_quad.log 2>&1; tail -n 5 $OUT/campaign_quad.log | head -2
(time python tools/parity_band512.py 150) > $OUT/band512.log 2>&1; tail -n 5 $OUT/band512.log | head -2
(time GAMDP_QUAD_MIN=1 python tools/parity_band512.py 150 150) > $OUT/band150.log 2>&1; tail -n 5 $OUT/band150.log | head -2
