#!/bin/bash
# On the GPU box: BASELINE config 5's call (4096 x 256 x 200, bench.config5_score_1vK) for the product library and each variant, alternated
# usage: tools/dev/ab_onevk.sh OUTTAG tag1 tag2 ...
OUT=gpurun_out/ab_$1.log; shift
: > $OUT
for rep in 1 2 3; do
  for t in product "$@"; do
    if [ $t = product ]; then L=""; else L=graphembeddings_amd/_variants/libge_$t.so; fi
    GE_LIB=$L timeout -k 10 200 python - >> $OUT 2>/dev/null <<PY || exit 1
import os, sys
sys.path.insert(0, ".")
if os.environ.get("GE_LIB"):
    from graphembeddings_amd import _lib as _L
    _L.LIB_PATH = os.path.abspath(os.environ["GE_LIB"])
import bench
r = bench.config5_score_1vK(200)
print("$t rep $rep us_per_call %.2f err %.2e" % (r["us_per_call"], r["max_abs_diff_vs_per_triple_kernel"]))
PY
  done
done
cat $OUT
