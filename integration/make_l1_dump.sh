#!/bin/bash
# Produces a reference-held golden dump of the merge-block (L1) driver -- what PctgBuilder::alignMergeBlock
# (lib/src/pctg/PctgBuilder.cc:726-844) read and wrote on the REFERENCE's own CPU path -- and drops it where
# tests/test_l1_external_golden.py finds it.  Needs what the reference needs (cmake, Boost, sparsehash) and the example
# pipeline's outputs; no GPU, no hipcc.  Cannot run in the build container of this repository (no Boost there): L1 parity
# stays "unpinned" until somebody runs this on such a host and commits tests/golden/l1_reference_dump/<name>/.
#
#   integration/make_l1_dump.sh <reference checkout> <name> <example dir> [--check-only]
#
#   <reference checkout>  a clean tree of vice87/gam-ngs
#   <name>                dataset name, e.g. gage_saureus_allpaths_msrca (BASELINE.json config 1)
#   <example dir>         the directory example/gam-ngs_pipeline.sh was run in: holds Assembly/<master>/genome.ctg.fasta,
#                         Assembly/<slave>/genome.ctg.fasta and gam-ngs_merge/{out.blocks,*.PE.list.txt}; if
#                         gam-ngs_merge/out.blocks is missing the script runs download_dataset.sh + gam-ngs_pipeline.sh
#                         there first (network, bwa, samtools)
#   MASTER=Allpaths-LG SLAVE=MSR-CA   (environment) the two assemblies, as in example/gam-ngs_pipeline.sh:42-96
#   --check-only          stop after patching + building the stand-in library + compiling the bridge (what the CPU test
#                         suite exercises here)
set -euo pipefail
REF=${1:?reference checkout}; NAME=${2:?dataset name}; EX=${3:?example directory}; MODE=${4:-}
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(dirname "$HERE")
MASTER=${MASTER:-Allpaths-LG}; SLAVE=${SLAVE:-MSR-CA}
WORK=$(mktemp -d "${TMPDIR:-/tmp}/gamdp_l1dump.XXXXXX")
echo "[make_l1_dump] work directory $WORK"

# 1. a patched copy of the reference (the checkout itself is not touched)
cp -r "$REF" "$WORK/ref"
(cd "$WORK/ref" && patch -p1 --no-backup-if-mismatch < "$HERE/gam-merge-gamdp.patch")

# 2. the library the patched CMakeLists links: the real one if it was built here, else a stand-in that refuses every call
#    (gam-merge then runs its own CPU path, which is what the dump is about)
GR="$WORK/gamdp_root"; mkdir -p "$GR/gam_ngs_amd"; ln -s "$ROOT/include" "$GR/include"
if [ -f "$ROOT/gam_ngs_amd/libgamdp.so" ] && [ "${GAMDP_FORCE_STUB:-0}" != 1 ]; then
    cp "$ROOT/gam_ngs_amd/libgamdp.so" "$GR/gam_ngs_amd/libgamdp.so"
else
    ${CC:-gcc} -O1 -shared -fPIC -I"$ROOT/include" -o "$GR/gam_ngs_amd/libgamdp.so" "$HERE/gamdp_stub.c"
fi
${CXX:-g++} -std=c++98 -fPIC -Wall -Werror -I"$ROOT/include" -I"$WORK/ref/lib/include" -c "$WORK/ref/lib/src/pctg/GamdpBridge.cc" -o "$WORK/bridge.o"
${CXX:-g++} -shared -o "$WORK/bridge_link_check.so" "$WORK/bridge.o" -L"$GR/gam_ngs_amd" -lgamdp -Wl,--no-undefined
echo "[make_l1_dump] patch applied, bridge compiles and links against $GR/gam_ngs_amd/libgamdp.so"
if [ "$MODE" = "--check-only" ]; then rm -rf "$WORK"; exit 0; fi

# 3. build gam-create / gam-merge the reference's way
mkdir -p "$WORK/ref/build"
(cd "$WORK/ref/build" && cmake -DGAMDP_ROOT="$GR" .. && make -j"$(nproc)" gam-create gam-merge)

# 4. the example pipeline's inputs
if [ ! -f "$EX/gam-ngs_merge/out.blocks" ]; then
    echo "[make_l1_dump] $EX/gam-ngs_merge/out.blocks missing: running the example pipeline (needs network, bwa, samtools)"
    cp "$WORK/ref/example/download_dataset.sh" "$WORK/ref/example/gam-ngs_pipeline.sh" "$EX/" 2>/dev/null || true
    mkdir -p "$EX/../bin" && cp "$WORK/ref/bin/gam-create" "$WORK/ref/bin/gam-merge" "$EX/../bin/"
    (cd "$EX" && bash download_dataset.sh && bash gam-ngs_pipeline.sh)
fi
MF="$EX/Assembly/$MASTER/genome.ctg.fasta"; SF="$EX/Assembly/$SLAVE/genome.ctg.fasta"
OUT="$ROOT/tests/golden/l1_reference_dump/$NAME"; mkdir -p "$OUT"

# 5. gam-merge on its own CPU path (GAMDP_DEVICES unset), one thread (the dump follows graphs_list order), dump hook on
(cd "$EX" && env -u GAMDP_DEVICES GAMDP_DUMP_PREFIX="$OUT/dump" LD_LIBRARY_PATH="$GR/gam_ngs_amd:${LD_LIBRARY_PATH:-}" \
    "$WORK/ref/bin/gam-merge" --blocks-file gam-ngs_merge/out.blocks \
    --master-bam "gam-ngs_merge/$MASTER.PE.list.txt" --master-fasta "$MF" \
    --slave-bam "gam-ngs_merge/$SLAVE.PE.list.txt" --slave-fasta "$SF" \
    --min-block-size 10 --output "$WORK/out" --threads 1)
cp "$MF" "$OUT/master.fasta"; cp "$SF" "$OUT/slave.fasta"
cp "$WORK/out.gam.fasta" "$OUT/reference.gam.fasta" 2>/dev/null || true   # the artefact BASELINE configs 1-2 compare
wc -l "$OUT"/dump.mergeblocks.tsv "$OUT"/dump.mergeblocks.out.tsv

# 6. the oracle against it right away (the GPU half runs with `pytest -m gpu` on an MI355X)
(cd "$ROOT" && python -m pytest tests/test_l1_external_golden.py -q -m "not gpu")
echo "[make_l1_dump] done: commit $OUT"
rm -rf "$WORK"
