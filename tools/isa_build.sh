#!/bin/bash
# gfx950 assembly of the product kernels (-save-temps) + per-function register / scratch statistics.   tools/isa_build.sh [pattern ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
B=$R/gam_ngs_amd/csrc/build/isa; mkdir -p $B; cd $B
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$R/gam_ngs_amd/csrc/ -Wall -Wno-unused-result $ISA_FLAGS -save-temps -c $R/gam_ngs_amd/csrc/gamdp_kernel.hip -o $B/k.o
python3 $R/tools/isa_funcs.py $B/gamdp_kernel-hip-amdgcn-amd-amdhsa-gfx950.s "$@"
