#!/bin/bash
# Calibrates FETCH_SIZE / WRITE_SIZE for narrow accesses (tools/hbm_gran.hip) on the GPU box:  tools/hbm_gran.sh [outdir]
# One rocprofv3 --pmc pass per counter set (never combined with API traces); tools/hbm_gran_summary.py prints counter / payload.
set -u
OUT=${1:-gpurun_out/hbm_gran}
mkdir -p $OUT
export TMPDIR=/tmp
# (rebuilt whenever the source is newer: the binary is not tracked)
[ tools/hbm_gran -nt tools/hbm_gran.hip ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/hbm_gran tools/hbm_gran.hip
timeout 120 tools/hbm_gran > $OUT/plain.log 2>&1
pass() { local name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- tools/hbm_gran > $OUT/$name.log 2>&1; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass wrreq TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass hit TCC_HIT_sum TCC_MISS_sum
python3 tools/hbm_gran_summary.py $OUT | tee $OUT/summary.txt
