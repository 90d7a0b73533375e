#!/bin/bash
# round 4: where the HBM traffic of a kernel comes from -- FETCH_SIZE / WRITE_SIZE (separate rocprofv3 --pmc passes) of the diagnostics
# build in its three modes: fill only, fill + the strips of a walk down the middle, everything.
#   tools/pmc_traffic_split.sh <out-tag> <band> <pairs>
set -u
: "${GRAFT_REPO_ROOT:?run under gpurun}"
TAG=$1; BAND=$2; PAIRS=$3
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp GAMDP_LIB=$PWD/gam_ngs_amd/libgamdp_diag.so
for mode in fill fillmat full; do
  unset GAMDP_DIAG_SKIP_TRACEBACK GAMDP_DIAG_COUNT_MAT
  [ $mode = fill ] && export GAMDP_DIAG_SKIP_TRACEBACK=1
  [ $mode = fillmat ] && export GAMDP_DIAG_SKIP_TRACEBACK=1 GAMDP_DIAG_COUNT_MAT=1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d $OUT/${mode}_$ctr -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-l1 --no-band150 --no-proxy --band $BAND --pairs $PAIRS > $OUT/${mode}_$ctr.log 2>&1
  done
done
python3 - $OUT $BAND $PAIRS <<'PY'
import csv, glob, sys
out, band, pairs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cells = pairs * (2 * band + 1) * 50000
for mode in ("fill", "fillmat", "full"):
    v = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        tot = 0.0
        for f in glob.glob("%s/%s_%s/**/*_counter_collection.csv" % (out, mode, ctr), recursive=True):
            for row in csv.DictReader(open(f)):
                if "k_align" in row["Kernel_Name"] and row["Counter_Name"] == ctr: tot += float(row["Counter_Value"])
        v[ctr] = tot * 1024.0
    rd, wr = 2.0 * v["FETCH_SIZE"], v["WRITE_SIZE"]
    print("%-8s read %.4f TB  written %.4f TB  = %.3f B per cell update (%.2f x 0.2507)" % (mode, rd / 1e12, wr / 1e12, (rd + wr) / cells, (rd + wr) / cells / 0.2507))
PY
