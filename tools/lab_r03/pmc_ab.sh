#!/bin/bash
# (lab notes: how a number quoted in DESIGN.md was measured; run under gpurun from the repo root)
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT = the repo copy on the GPU box)}"
# PMC comparison of library builds on one workload: tools/lab_r03/pmc_ab.sh "<bench args>" lib1.so lib2.so ...  ("-" = product)
ARGS="$1"; shift
mkdir -p gpurun_out/pmcab
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/pmcab/counters_avail.txt 2>&1
for lib in "$@"; do
 if [ "$lib" = "-" ]; then unset GAMDP_LIB; tag=prod; else export GAMDP_LIB=$PWD/$lib; tag=$(basename $lib .so); fi
 i=0
 for set in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_READ_sum" \
            "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmcab/${tag}_$i -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-l1 --no-band150 $ARGS > gpurun_out/pmcab/${tag}_$i.log 2>&1
 done
done
python3 - <<'PY'
import sqlite3,glob,os
for d in sorted(glob.glob("gpurun_out/pmcab/*_[0-9]")):
    for f in glob.glob(d+"/*.db"):
        db=sqlite3.connect(f); c=db.cursor()
        tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
        pmc=[t for t in tabs if 'pmc_event' in t]; info=[t for t in tabs if 'info_pmc' in t]
        try:
            for r in c.execute("select i.name, sum(e.value) from %s e join %s i on e.pmc_id=i.id group by i.name"%(pmc[0],info[0])): print(os.path.basename(d), r[0], "%.4g"%r[1])
        except Exception as ex: print(d, "ERR", ex)
PY
