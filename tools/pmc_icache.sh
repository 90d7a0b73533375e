#!/bin/bash
# instruction-cache counters of the two throughput kernels (one rocprofv3 --pmc pass each, nothing else traced):   tools/pmc_icache.sh [tag]
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun}"
TAG=${1:-r04_icache}; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
for cfg in "b512:" "b150:--band 150"; do
  name=${cfg%%:*}; extra=${cfg#*:}
  rocprofv3 --pmc SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/$name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-l1 --no-band150 --no-proxy $extra > $OUT/$name.log 2>&1 || { echo "$name: rocprofv3 failed"; tail -5 $OUT/$name.log; continue; }
  python3 - $OUT/$name $name <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_align" in r["Kernel_Name"]:
            tot[(r["Kernel_Name"][:60], r["Counter_Name"])] += float(r["Counter_Value"]); n[(r["Kernel_Name"][:60], r["Counter_Name"])] += 1
for k in sorted(tot): print(sys.argv[2], k[0], k[1], "%.4g" % tot[k], "rows", n[k])
PY
done
