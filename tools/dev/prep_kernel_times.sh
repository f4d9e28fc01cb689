#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of the large-batch probe for the product library and each variant; prints the prepare kernels' average durations
cd /tmp && export TMPDIR=/tmp
for t in product "$@"; do
  if [ $t = product ]; then L=""; else L=$GRAFT_REPO_ROOT/graphembeddings_amd/_variants/libge_$t.so; fi
  rm -rf /tmp/pk_$t
  GE_LIB=$L timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk_$t -- python3 $GRAFT_REPO_ROOT/tools/bigbatch_probe.py 1200000 65536 > /tmp/pk_$t.log 2>&1
  echo "== $t"
  python3 - /tmp/pk_$t <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "prep_big" in r["Name"] or "order_" in r["Name"]:
        print("  ", r["Name"][:44], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us avg")
PY
done
