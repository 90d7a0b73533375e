#!/bin/bash
# (lab notes: how a number quoted in DESIGN.md was measured; run under gpurun from the repo root)
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT = the repo copy on the GPU box)}"
# round 3, probe 4: longest-remaining-first issue priority -- parity, then A/B at short launches (GAMDP_NO_PRIO=1 = off)
mkdir -p gpurun_out/r03_probe4
timeout 900 python -m pytest tests/test_gpu_l0_parity.py -x -q -m gpu -k "not fresh_process and not range_assertion" > gpurun_out/r03_probe4/pytest.log 2>&1; tail -3 gpurun_out/r03_probe4/pytest.log
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-l1 --no-band150 --no-proxy"
run() { name=$1; shift; $B "$@" > gpurun_out/r03_probe4/$name.log 2>&1; python - gpurun_out/r03_probe4/$name.log $name <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print("%-22s gcups %.0f kernel_ms %.1f ms_step %.1f kernel %s"%(sys.argv[2], d["value"], d["roofline"]["kernel_ms_per_launch"], d["ms_per_step"], d["roofline"]["kernel"]))
PY
}
for P in 100000 25000 12500 8192 4096 2048; do
  run prio_$P --pairs $P
  GAMDP_NO_PRIO=1 run noprio_$P --pairs $P
done
run prio_b150_100k --band 150
GAMDP_NO_PRIO=1 run noprio_b150_100k --band 150
run prio_b150_12500 --band 150 --pairs 12500
GAMDP_NO_PRIO=1 run noprio_b150_12500 --band 150 --pairs 12500
GAMDP_LIB=$PWD/gam_ngs_amd/libgamdp_diag_hwid.so python tools/hwid_hist.py 12500 2>&1 | cut -c1-400
python bench_l1.py --genome 2900000 --steps 5 --verify 0 | cut -c1-900
GAMDP_NO_PRIO=1 python bench_l1.py --genome 2900000 --steps 5 --verify 0 | cut -c1-900
