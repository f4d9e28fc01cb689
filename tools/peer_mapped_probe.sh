#!/bin/bash
# Runs on a ONE-GPU box: the row-sharded bench at world 2 and 4 with all ranks on cuda:0 over gloo, once with the row
# all-to-all (host-staged under gloo!) and once with the peer-mapped fetch.  Mechanics and parity only: both
# "transports" are the same device here, nothing about xGMI can be read from the times.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd "$ROOT"
: > "$OUT/r03_peer_mapped_experiment.txt"
for W in 2 4; do for MODE in "" "--peer-mapped"; do
  GE_DIST_BACKEND=gloo GE_SINGLE_DEVICE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port $((29540 + W)) bench.py --gpus $W --steps 16 --warmup 16 --batch 16384 $MODE > "$OUT/peer_$W$MODE.log" 2>&1
  python3 - "$OUT/peer_$W$MODE.log" "$W" "$MODE" >> "$OUT/r03_peer_mapped_experiment.txt" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not line:
    print(json.dumps({"world": int(sys.argv[2]), "mode": sys.argv[3] or "all-to-all", "error": open(sys.argv[1]).read()[-400:]}))
else:
    d = json.loads(line[-1])
    print(json.dumps({"world": int(sys.argv[2]), "mode": sys.argv[3] or "all-to-all (gloo: staged through the host)", "batch_per_rank": d["config"]["batch_per_gpu"],
                      "ms_per_step": round(d["ms_per_step"], 3), "remote_rows_per_step": d["config"]["remote_rows_per_step"],
                      "final_mean_hinge": d["config"]["final_mean_hinge"]}))
PY
done; done
cat "$OUT/r03_peer_mapped_experiment.txt"
