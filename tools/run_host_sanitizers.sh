#!/bin/bash
# Host-only tests (no GPU needed) against the AddressSanitizer + UBSan build of the library's host code:
# FASTA loader, findHits, the post-alignment stage and its writers, the C-ABI symbol table.
set -e
cd "$(dirname "$0")/.."
make -C gam_ngs_amd/csrc asan -j8 > /dev/null
RT=$(find /opt/rocm/lib/llvm -name 'libclang_rt.asan-x86_64.so' | head -1)
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_PRELOAD=$RT \
GAMDP_LIB=$PWD/gam_ngs_amd/csrc/build/libgamdp_asan.so \
python -m pytest tests/test_pctg_stage.py tests/test_blocks_io.py tests/test_cabi_symbols.py tests/test_oracle_golden.py -x -q
