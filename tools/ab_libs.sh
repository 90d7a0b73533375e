#!/bin/bash
# kernel time of several builds of the library on ONE box, alternating:   tools/ab_libs.sh "<lib> <lib> ..." [rounds] [shapes...]
set -u
: "${GRAFT_REPO_ROOT:?run under gpurun}"
LIBS=$1; R=${2:-2}; shift 2
[ $# -eq 0 ] && set -- "131072 5000" "131072 2000" "65536 20000" "32768 50000" "mixed 5"
for shape in "$@"; do
  for i in $(seq $R); do
    for l in $LIBS; do GAMDP_LIB=$PWD/gam_ngs_amd/$l timeout 300 python3 tools/ab_kernel.py $shape; done
  done
done
