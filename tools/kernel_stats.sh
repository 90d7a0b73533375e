#!/bin/bash
# Registers, private memory (spills + out-of-line phase state) and LDS of every kernel in the built library:
#   tools/kernel_stats.sh [path/to/libgamdp.so]
set -eu
LIB=${1:-$(dirname "$0")/../gam_ngs_amd/libgamdp.so}
BIN=/opt/rocm/lib/llvm/bin
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
$BIN/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$LIB" --output="$TMP/dev.co" --unbundle 2>/dev/null || {
  # a shared library: the fat binary sits in .hip_fatbin
  $BIN/llvm-objcopy -O binary --only-section=.hip_fatbin "$LIB" "$TMP/fat.bin"
  $BIN/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$TMP/fat.bin" --output="$TMP/dev.co" --unbundle
}
$BIN/llvm-readelf --notes "$TMP/dev.co" | python3 -c '
import re, sys
txt = sys.stdin.read()
for m in re.finditer(r"\.name:\s+(\S+).*?(?=\n\s+- \.agpr_count|\Z)", txt, re.S):
    pass
cur = {}
rows = []
for line in txt.splitlines():
    line = line.strip()
    for key in (".name:", ".vgpr_count:", ".agpr_count:", ".sgpr_count:", ".private_segment_fixed_size:", ".group_segment_fixed_size:", ".vgpr_spill_count:"):
        if line.startswith(key) or line.startswith("- " + key):
            cur[key] = line.split(":", 1)[1].strip()
    if line.startswith(".wavefront_size:") or line.startswith("- .wavefront_size:"):
        if ".name:" in cur: rows.append(cur)
        cur = {}
import subprocess
for r in rows:
    name = subprocess.run(["c++filt", r[".name:"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"gamdp::\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "")
    print("%-28s vgpr %3s agpr %3s sgpr %3s private %5s B lds %6s B spills %s" % (name, r.get(".vgpr_count:"), r.get(".agpr_count:"), r.get(".sgpr_count:"), r.get(".private_segment_fixed_size:"), r.get(".group_segment_fixed_size:"), r.get(".vgpr_spill_count:")))
'
