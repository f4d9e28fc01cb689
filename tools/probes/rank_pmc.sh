#!/bin/bash
# GPU box: a few SQ counters for the rank sweep (tools/rank_only.py), one rocprofv3 --pmc pass per group
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/rank_pmc; rm -rf "$OUT"; mkdir -p "$OUT"
i=0
SETS=("SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum")
[ -n "$RANK_PMC_SETS" ] && SETS=("$RANK_PMC_SETS")      # one set of counters, space separated
for C in "${SETS[@]}"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 "$ROOT/tools/rank_only.py" 2 > "$OUT/p$i.log" 2>&1 || echo "pass $i failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for p in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "rank_f16_kernel" in r["Kernel_Name"] or "rank_pipe_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k}: {sum(v)/len(v):.4g} per launch ({len(v)} rows)")
PY
