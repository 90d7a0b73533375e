#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's numbers on the GPU box (run from the repo root):
#   tools/collect_profiles.sh <tag> [extra bench.py args]     e.g.  tools/collect_profiles.sh r02
#                                                                    tools/collect_profiles.sh r02_band150 --band 150
# Pass 1: --kernel-trace --stats of the bench.py run (default workload: 4 launches of the full 100 000-pair set; --no-proxy: the
# strong-scaling proxies launch the same kernel on other batch sizes and would be averaged into its statistics).
# Passes 2..6: PMC counters, one set per pass, kernel-trace/stats only (never combined with API traces), each on
# a single launch (--steps 1 --warmup 0).  Everything lands under gpurun_out/<tag>_*; tools/summarise_profiles.py
# turns it into the files committed under profiles/.  The commit is recorded next to the data (gpurun_out/<tag>_commit)
# by the caller, since the GPU box has no .git.
set -u
TAG=${1:-r02}; shift || true
EXTRA="$@"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python3 tools/srchash.py > $OUT/${TAG}_source_hash    # the sources these counters belong to (bench.py: profile_matches_source)
if [ -z "$EXTRA" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 bench.py --no-proxy > $OUT/${TAG}_bench.log 2> $OUT/${TAG}_trace.log
else
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 bench.py --no-l1 --no-band150 --no-proxy --no-mixed150 --no-cpu-baseline $EXTRA > $OUT/${TAG}_bench.log 2> $OUT/${TAG}_trace.log
fi
pmc() {  # name, counters...
    local name=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_$name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-l1 --no-band150 --no-proxy --no-mixed150 $EXTRA > $OUT/${TAG}_$name.log 2>&1
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pmc sq3 SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU
tail -1 $OUT/${TAG}_bench.log | cut -c1-400
