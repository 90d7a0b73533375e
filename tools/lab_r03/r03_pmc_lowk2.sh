#!/bin/bash
# (lab notes: how a number quoted in DESIGN.md was measured; run under gpurun from the repo root)
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT = the repo copy on the GPU box)}"
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03_pmc_lowk2; mkdir -p $OUT
export GAMDP_LIB=$R/gam_ngs_amd/libgamdp_diag.so GAMDP_DIAG_SKIP_TRACEBACK=1
for P in 2048; do
 for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $OUT/p${P}_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-l1 --no-band150 --no-proxy --pairs $P > $OUT/p${P}_$tag.log 2>&1
 done
done
python3 - $OUT <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for P in (2048,):
    c = {}
    for f in glob.glob(os.path.join(out, "p%d_*" % P, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_align" in row["Kernel_Name"]:
                c[row["Counter_Name"]] = c.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    print(P, " ".join("%s=%.4g" % (k, v) for k, v in sorted(c.items())))
PY
cd $R
for P in 2048 4096; do python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-l1 --no-band150 --no-proxy --pairs $P 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('fill only', d['config']['pairs_total_per_step'], 'kernel_ms %.2f'%d['roofline']['kernel_ms_per_launch'])
"; done
