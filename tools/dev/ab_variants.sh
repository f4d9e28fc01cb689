#!/bin/bash
# On the GPU box: the large-batch train step timed with the product library and each variant named (alternated twice).
# usage: tools/dev/ab_variants.sh OUTTAG tag1 tag2 ...
OUT=gpurun_out/ab_$1.log; shift
: > $OUT
for rep in 1 2; do
  for t in product "$@"; do
    if [ $t = product ]; then L=""; else L=graphembeddings_amd/_variants/libge_$t.so; fi
    echo "== $t (rep $rep)" >> $OUT
    GE_LIB=$L timeout -k 10 300 python tools/bigbatch_probe.py 1200000 65536 2>/dev/null | grep '^{' >> $OUT || exit 1
  done
done
cat $OUT
