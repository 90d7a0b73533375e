#!/bin/bash
cd $GRAFT_REPO_ROOT
D=$PWD/gam_ngs_amd/libgamdp_diag.so
for shape in "mixed 6" "131072 5000" "131072 2000" "32768 50000"; do
  GAMDP_LIB=$D python3 tools/ab_kernel.py $shape
  GAMDP_LIB=$D GAMDP_DIAG_SKIP_TRACEBACK=1 python3 tools/ab_kernel.py $shape | sed 's/^/SKIP_TB /'
done
