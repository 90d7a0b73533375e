#!/bin/bash
# A/B of product-build variants on the GPU box (run from the repo root):  tools/ab_r05.sh <tag> <variant> [<variant> ...]
# A variant is the NAME of gam_ngs_amd/libgamdp_<NAME>.so (make -C gam_ngs_amd/csrc pvariant NAME=.. FLAGS=..); "product" = libgamdp.so.
# Per variant: the whole step timed (bench.py, 2 steps), then one launch each under --pmc FETCH_SIZE and --pmc WRITE_SIZE (passes of
# their own, kernel-trace only).  BAND / PAIRS / LEN choose the workload (default: 100 000 x 50 kb at band 150).  PMC=0 skips the counters.
set -u
TAG=$1; shift
BAND=${BAND:-150}; PAIRS=${PAIRS:-100000}; LEN=${LEN:-50000}; PMC=${PMC:-1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--band $BAND --pairs $PAIRS --len $LEN --no-l1 --no-band150 --no-proxy --no-cpu-baseline --no-mixed150"
for v in "$@"; do
  if [ "$v" = product ]; then unset GAMDP_LIB; else export GAMDP_LIB=$PWD/gam_ngs_amd/libgamdp_$v.so; fi
  timeout 600 python3 bench.py $ARGS --steps 2 --warmup 1 > $OUT/${v}_bench.log 2> $OUT/${v}_bench.err
  if [ "$PMC" = 1 ]; then
    timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${v}_fetch -- python3 bench.py $ARGS --steps 1 --warmup 0 > $OUT/${v}_fetch.log 2>&1
    timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${v}_write -- python3 bench.py $ARGS --steps 1 --warmup 0 > $OUT/${v}_write.log 2>&1
  fi
done
python3 tools/ab_r05_summary.py $OUT "$@" | tee $OUT/summary.txt
