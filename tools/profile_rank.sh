#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats + separate PMC passes (matrix-pipe busy cycles, GPU active cycles)
# for the fused rank sweep (tools/rank_only.py).  Output under gpurun_out/; the summary is gpurun_out/<tag>_rank_profile.txt
set -e
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; TAG=${1:-r02}
mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_rank_trace" -- python3 "$ROOT/tools/rank_only.py" 5 > "$OUT/${TAG}_rank_trace.log" 2>&1
for C in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU TCC_HIT_sum TCC_MISS_sum; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/${TAG}_rank_pmc_$C" -- python3 "$ROOT/tools/rank_only.py" 2 > "$OUT/${TAG}_rank_pmc_$C.log" 2>&1
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys
out, tag = sys.argv[1], sys.argv[2]
with open(f"{out}/{tag}_rank_profile.txt", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 tools/rank_only.py 5   (59,071 x 14,951 x d=200 sweep; candidate planes built once)\n")
    f.write("# counters: one rocprofv3 --pmc pass each, sweep kernel only; MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)\n")
    for p in glob.glob(f"{out}/{tag}_rank_trace/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(p)):
            if "rank" in row["Name"] or "planes" in row["Name"]:
                f.write(f"{row['Name'][:90]}  calls {row['Calls']}  avg_ns {row['AverageNs']}  min_ns {row['MinNs']}  max_ns {row['MaxNs']}\n")
    sweep = lambda name: any(k in name for k in ("rank_f16_kernel", "rank_pipe_kernel", "rank_1vK_kernel"))   # not the pre-pass
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "TCC_HIT_sum", "TCC_MISS_sum"):
        for p in glob.glob(f"{out}/{tag}_rank_pmc_{c}/**/*counter_collection.csv", recursive=True):
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(p)) if sweep(r["Kernel_Name"]) and r["Counter_Name"] == c]
            if vals:
                f.write(f"{c}: per launch {sum(vals) / len(vals):.4g}  ({len(vals)} launches)\n")
PY
cat "$OUT/${TAG}_rank_profile.txt"
