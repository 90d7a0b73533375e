#!/bin/bash
# (lab notes: how a number quoted in DESIGN.md was measured; run under gpurun from the repo root)
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT = the repo copy on the GPU box)}"
# round 3: what a lone band-150 wavefront does with its cycles (256 calls of 50 kb, one per CU; fill only), direction-free and tagged
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03_pmc_lone150; mkdir -p $OUT
export GAMDP_LIB=$R/gam_ngs_amd/libgamdp_diag.so GAMDP_DIAG_SKIP_TRACEBACK=1
for V in df tagged; do
 if [ $V = tagged ]; then export GAMDP_NO_DF5=1; else unset GAMDP_NO_DF5; fi
 for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -s KILL 200 rocprofv3 --pmc $set --output-format csv -d $OUT/${V}_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-l1 --no-band150 --no-proxy --band 150 --pairs 256 > $OUT/${V}_$tag.log 2>&1
 done
done
python3 - $OUT <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for V in ("df", "tagged"):
    c = {}
    for f in glob.glob(os.path.join(out, V + "_*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_align" in row["Kernel_Name"]:
                c[row["Counter_Name"]] = c.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    print(V, " ".join("%s=%.4g" % (k, v) for k, v in sorted(c.items())))
PY
