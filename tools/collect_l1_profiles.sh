#!/bin/bash
# rocprofv3 --kernel-trace --stats of the merge-block record (bench_l1.py) on the GPU box, run from the repo root:
#   tools/collect_l1_profiles.sh <tag>        -> gpurun_out/<tag>_l1_{2p9mb,30mb}_{trace/,bench.log}
set -u
TAG=${1:-r03}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_l1_2p9mb_trace -- python3 bench_l1.py --steps 10 > $OUT/${TAG}_l1_2p9mb_bench.log 2> $OUT/${TAG}_l1_2p9mb_trace.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_l1_30mb_trace -- python3 bench_l1.py --genome 30000000 --steps 5 > $OUT/${TAG}_l1_30mb_bench.log 2> $OUT/${TAG}_l1_30mb_trace.log
for w in 2p9mb 30mb; do
  f=$(find $OUT/${TAG}_l1_${w}_trace -name '*kernel_stats.csv' | head -1)
  cp "$f" $OUT/${TAG}_l1_${w}_kernel_stats.csv
  head -6 "$f" | cut -c1-200
  tail -1 $OUT/${TAG}_l1_${w}_bench.log | cut -c1-300
done
