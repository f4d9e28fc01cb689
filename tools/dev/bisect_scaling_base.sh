#!/bin/bash
# On the GPU box: which auxiliary leg of the default bench line slows the scaling_base leg that runs after it
ALL="--no-score-roofline --no-hbm-roofline --no-rank-roofline --no-extra-configs --no-cpu-baseline"
for keep in ${@:-none score-roofline hbm-roofline rank-roofline extra-configs cpu-baseline}; do
  FL=""
  for f in $ALL; do [ "$f" = "--no-$keep" ] || FL="$FL $f"; done
  timeout -k 10 280 python bench.py $FL 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$keep', 'scaling_base ms', round(d['scaling_base']['ms_per_step'],4), 'host', round(d['scaling_base']['host_enqueue_ms_per_step'],4), 'value', round(d['value']/1e6,1))" || exit 1
done
