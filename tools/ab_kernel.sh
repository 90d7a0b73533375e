#!/bin/bash
# A/B of two builds of the library on ONE box, alternating.   tools/ab_kernel.sh <libA> <libB> [rounds]
set -u
: "${GRAFT_REPO_ROOT:?run under gpurun}"
A=$PWD/gam_ngs_amd/$1; B=$PWD/gam_ngs_amd/$2; R=${3:-2}
for shape in "131072 5000" "131072 2000" "65536 20000" "32768 50000" "mixed 5" "16384 50000 512"; do
  for i in $(seq $R); do
    GAMDP_LIB=$A timeout 300 python3 tools/ab_kernel.py $shape
    GAMDP_LIB=$B timeout 300 python3 tools/ab_kernel.py $shape
  done
done
