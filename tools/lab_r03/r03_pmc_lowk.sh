#!/bin/bash
# (lab notes: how a number quoted in DESIGN.md was measured; run under gpurun from the repo root)
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT = the repo copy on the GPU box)}"
# round 3: why a SIMD with one or two wavefronts runs the fill at 55-72 %: SQ counters of the fill alone (diagnostics build,
# GAMDP_DIAG_SKIP_TRACEBACK) at 2048 pairs (1 wavefront per SIMD), 4096 (2) and 40960 (4, ten rounds)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03_pmc_lowk; mkdir -p $OUT
export GAMDP_LIB=$R/gam_ngs_amd/libgamdp_diag.so GAMDP_DIAG_SKIP_TRACEBACK=1
for P in 2048 4096 40960; do
 for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $OUT/p${P}_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-l1 --no-band150 --no-proxy --pairs $P > $OUT/p${P}_$tag.log 2>&1
 done
done
python3 - $OUT <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for P in (2048, 4096, 40960):
    c = {}
    for f in glob.glob(os.path.join(out, "p%d_*" % P, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_align" in row["Kernel_Name"]:
                c[row["Counter_Name"]] = c.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    print(P, " ".join("%s=%.4g" % (k, v) for k, v in sorted(c.items())))
PY
