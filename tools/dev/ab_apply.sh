#!/bin/bash
# On the GPU box: the update kernel of the large-batch step for the product library and each variant named, on the default workload
# (Zipf 0.8 ids, every pair hinge-active) and with no pair active (margin -1: what the kernel takes to walk its items and find nothing to add)
# usage: tools/dev/ab_apply.sh OUTTAG tag1 tag2 ...
OUT=gpurun_out/ab_$1.log; shift
: > $OUT
for rep in 1 2; do
  for t in product "$@"; do
    if [ $t = product ]; then L=""; else L=graphembeddings_amd/_variants/libge_$t.so; fi
    for a in "0.8 0.2" "0.8 -1"; do
      GE_LIB=$L timeout -k 10 300 python tools/bigbatch_probe.py 1200000 65536 $a 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$t rep $rep margin', d['margin'], 'step %.1f grad %.1f apply %.1f' % (d['us_per_step'], d['grad_us'], d['apply_us']))" >> $OUT || exit 1
    done
  done
done
cat $OUT
