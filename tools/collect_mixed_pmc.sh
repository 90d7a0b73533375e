#!/bin/bash
# PMC counters of the driver-shaped batch (bench.py's mixed150 record; tools/mixed150.py), one rocprofv3 --pmc pass per set, never
# combined with API traces:   tools/collect_mixed_pmc.sh <tag>      -> gpurun_out/<tag>_mixed_*
set -u
TAG=${1:-r06}; OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 tools/srchash.py > $OUT/${TAG}_mixed_source_hash
pmc() { local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_mixed_$name -- python3 tools/mixed150.py --steps 1 > $OUT/${TAG}_mixed_$name.log 2>&1; }
pmc sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pmc sq3 SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
