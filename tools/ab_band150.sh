# A/B timing of the band-150 throughput kernels on the GPU box: full / fill only / fill + strips (diagnostics builds;
# libgamdp_diag_sl4.so = the same with -DGAMDP_QUAD_STRIP_LANES=4, built by hand for this comparison)
mkdir -p gpurun_out/ab150
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-l1 --band 150"
for lib in diag diag_sl4; do
 D=$PWD/gam_ngs_amd/libgamdp_$lib.so
 [ -f $D ] || continue
 for len in 50000 5000; do
  if [ $len = 50000 ]; then P=100000; else P=400000; fi
  GAMDP_LIB=$D $B --len $len --pairs $P > gpurun_out/ab150/${lib}_full_$len.log 2>&1
  GAMDP_LIB=$D GAMDP_DIAG_SKIP_TRACEBACK=1 $B --len $len --pairs $P > gpurun_out/ab150/${lib}_fill_$len.log 2>&1
  GAMDP_LIB=$D GAMDP_DIAG_SKIP_TRACEBACK=1 GAMDP_DIAG_COUNT_MAT=1 $B --len $len --pairs $P > gpurun_out/ab150/${lib}_fillmat_$len.log 2>&1
 done
 GAMDP_LIB=$D GAMDP_DIAG_COUNT_MAT=1 python tools/count_materialise.py 150 40960 50000 > gpurun_out/ab150/${lib}_count.log 2>&1
 GAMDP_LIB=$D GAMDP_DIAG_COUNT_MAT=1 GAMDP_QUAD_MIN=1 python tools/count_materialise.py 150 64 50000 >> gpurun_out/ab150/${lib}_count.log 2>&1
done
for f in gpurun_out/ab150/*.log; do echo $f; python - "$f" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print("   gcups %.0f kernel_ms %.1f ms_step %.1f"%(d["value"], d["roofline"]["kernel_ms_per_launch"], d["ms_per_step"]))
    elif 'band' in l or 'first tasks' in l: print("  ", l.rstrip())
PY
done
