#!/bin/bash
# round 4: the parity campaign on the final kernels (GPU box, from the repo root): default kernel choice, every band-150 call through
# the four- / eight-task kernels, the int32 top blocks, and the window cases built for the direction-free / packed ranges
set -u
OUT=gpurun_out/${1:-r04q}; mkdir -p $OUT
(time python tools/parity_campaign.py --seeds 40 --per-seed 1000 --long 400 --first-seed ${SEED0:-400}) > $OUT/campaign_default.log 2>&1; tail -3 $OUT/campaign_default.log
(time GAMDP_QUAD_MIN=1 python tools/parity_campaign.py --seeds 21 --per-seed 1000 --long 400 --first-seed $((${SEED0:-400}+40))) > $OUT/campaign_quad.log 2>&1; tail -3 $OUT/campaign_quad.log
(time GAMDP_NO_PACKED_TOP=1 python tools/parity_campaign.py --seeds 7 --per-seed 1000 --long 300 --first-seed $((${SEED0:-400}+70))) > $OUT/campaign_notop.log 2>&1; tail -2 $OUT/campaign_notop.log
python tools/parity_band512.py 6 > $OUT/band512.log 2>&1; tail -2 $OUT/band512.log
GAMDP_QUAD_MIN=1 python tools/parity_band512.py 9 150 > $OUT/band150.log 2>&1; tail -2 $OUT/band150.log
(time GAMDP_NO_STRIP_SHIFT=1 GAMDP_QUAD_MIN=1 python tools/parity_campaign.py --seeds 7 --per-seed 1000 --long 300 --first-seed $((${SEED0:-400}+80))) > $OUT/campaign_noshift.log 2>&1; tail -2 $OUT/campaign_noshift.log
