#!/bin/bash
# rocprofv3 evidence of the merge-block record (bench_l1.py) on the GPU box, run from the repo root:
#   tools/collect_l1_profiles.sh <tag>        -> gpurun_out/<tag>_l1_{2p9mb,30mb}_*
# Pass 1: --kernel-trace --stats.  Passes 2..4: PMC counters of the chain kernel (k_chain2), one set per pass, never combined
# with API traces, two calls each (--steps 1, one warm-up call).
set -u
TAG=${1:-r04}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python3 tools/srchash.py > $OUT/${TAG}_l1_source_hash
for w in 2p9mb 30mb; do
  if [ $w = 2p9mb ]; then G=2900000; S=10; else G=30000000; S=5; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_l1_${w}_trace -- python3 bench_l1.py --genome $G --steps $S --no-cpu-baseline > $OUT/${TAG}_l1_${w}_bench.log 2> $OUT/${TAG}_l1_${w}_trace.log
  f=$(find $OUT/${TAG}_l1_${w}_trace -name '*kernel_stats.csv' | head -1)
  cp "$f" $OUT/${TAG}_l1_${w}_kernel_stats.csv
  head -6 "$f" | cut -c1-200
  tail -1 $OUT/${TAG}_l1_${w}_bench.log | cut -c1-300
  pmc() { local name=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_l1_${w}_$name -- python3 bench_l1.py --genome $G --steps 1 --verify 0 --no-cpu-baseline > $OUT/${TAG}_l1_${w}_$name.log 2>&1; }
  pmc fetch FETCH_SIZE
  pmc write WRITE_SIZE
  pmc sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
done
