#!/bin/bash
cd $GRAFT_REPO_ROOT
B="timeout 600 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-l1 --no-band150 --no-proxy --no-mixed150 --band 150"
run() { name=$1; shift; "$@" > /tmp/$name.log 2>&1; python3 - /tmp/$name.log $name <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print("%-26s gcups %6.0f kernel_ms %7.1f ms_step %7.2f launches %d"%(sys.argv[2], d["value"], d["roofline"]["kernel_ms_per_launch"], d["ms_per_step"], d["roofline"]["launches"]))
PY
}
for shape in "131072 2000" "200000 2000" "262144 1000" "200000 5000" "300000 3000" "100000 5000"; do
  set -- $shape
  run p_$1_$2 $B --pairs $1 --len $2
  GAMDP_CHUNK_MIN=999999999 run w_$1_$2 env GAMDP_CHUNK_MIN=999999999 $B --pairs $1 --len $2
done
