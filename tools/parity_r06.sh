#!/bin/bash
# round 6: the parity campaign on the final library (GPU box, from the repo root): default kernel choice, every band-150 call through the
# eight-task kernel, driver-shaped batches with many tail calls (the side-by-side N-aware launch is in their way too), and the bands beyond
# the systolic kernels (k_align_w).   SEED0=... tools/parity_r06.sh <tag>
set -u
OUT=gpurun_out/${1:-r06q}; mkdir -p $OUT
S=${SEED0:-1300}
(time timeout 900 python3 tools/parity_campaign.py --seeds 14 --per-seed 1000 --long 200 --first-seed $S) > $OUT/campaign_default.log 2>&1; tail -3 $OUT/campaign_default.log
(time GAMDP_QUAD_MIN=1 timeout 900 python3 tools/parity_campaign.py --seeds 14 --per-seed 1000 --long 200 --first-seed $((S+40))) > $OUT/campaign_quad.log 2>&1; tail -3 $OUT/campaign_quad.log
(time timeout 900 python3 tools/parity_campaign.py --wide --seeds 6 --per-seed 400 --long 60 --first-seed $((S+80))) > $OUT/campaign_wide.log 2>&1; tail -3 $OUT/campaign_wide.log
(time timeout 900 python3 tools/mixed_stress.py $((S+100)) 4 2048 0.5) > $OUT/mixed_stress.log 2>&1; tail -2 $OUT/mixed_stress.log
(time timeout 900 python3 tools/mixed_stress.py $((S+110)) 2 9000 0.1) > $OUT/mixed_stress_big.log 2>&1; tail -2 $OUT/mixed_stress_big.log
