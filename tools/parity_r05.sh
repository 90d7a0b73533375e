#!/bin/bash
# round 5: the parity campaign on the final kernels (GPU box, from the repo root): default kernel choice, every band-150 call through
# the four- / eight-task kernels (with and without the per-task packed top blocks), the window cases built for the direction-free /
# packed ranges at both bands.   SEED0=... tools/parity_r05.sh <tag>
set -u
OUT=gpurun_out/${1:-r05q}; mkdir -p $OUT
S=${SEED0:-900}
(time timeout 900 python3 tools/parity_campaign.py --seeds 21 --per-seed 1000 --long 300 --first-seed $S) > $OUT/campaign_default.log 2>&1; tail -3 $OUT/campaign_default.log
(time GAMDP_QUAD_MIN=1 timeout 900 python3 tools/parity_campaign.py --seeds 21 --per-seed 1000 --long 300 --first-seed $((S+40))) > $OUT/campaign_quad.log 2>&1; tail -3 $OUT/campaign_quad.log
(time GAMDP_QUAD_MIN=1 GAMDP_NO_PACKED_TOP_MIXED=1 timeout 600 python3 tools/parity_campaign.py --seeds 7 --per-seed 1000 --long 200 --first-seed $((S+70))) > $OUT/campaign_nomixed.log 2>&1; tail -2 $OUT/campaign_nomixed.log
timeout 600 python3 tools/parity_band512.py 6 > $OUT/band512.log 2>&1; tail -2 $OUT/band512.log
GAMDP_QUAD_MIN=1 timeout 600 python3 tools/parity_band512.py 9 150 > $OUT/band150.log 2>&1; tail -2 $OUT/band150.log
