#!/bin/bash
# rocprofv3 kernel-trace + PMC (separate passes) for the forward score kernel; summary -> gpurun_out/<tag>_score_*.json
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_score_trace -- python3 $ROOT/tools/score_roofline.py 10 > $OUT/${TAG}_score_trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_score_pmc_$C -- python3 $ROOT/tools/score_roofline.py 4 > $OUT/${TAG}_score_pmc_$C.log 2>&1
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
res = {"_command": "rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/score_roofline.py",
       "_units": "counters in KB per launch; FETCH_SIZE doubled in hbm_bytes_corrected (gfx950: reports half of wide coalesced reads)"}
st = glob.glob(f"{out}/{tag}_score_trace/*/*_kernel_stats.csv")
if st:
    for r in csv.DictReader(open(st[0])):
        if "complex_score_kernel" in r["Name"]:
            res["kernel_trace"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"{out}/{tag}_score_pmc_{C}/*/*_counter_collection.csv")
    if fs:
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if r["Counter_Name"] == C and "complex_score_kernel" in r["Kernel_Name"]]
        if v: res[C + "_KB"] = sum(v) / len(v)
alg = (12 * 200 + 16) * (1 << 22)
res["algorithmic_bytes_per_launch"] = alg
if "FETCH_SIZE_KB" in res:
    res["hbm_bytes_corrected"] = (2 * res["FETCH_SIZE_KB"] + res.get("WRITE_SIZE_KB", 0)) * 1024
if "kernel_trace" in res:
    t = res["kernel_trace"]["avg_ns"] * 1e-9
    res["algorithmic_GBs"] = alg / t / 1e9
    res["frac_of_8TBs"] = alg / t / 8e12
    if "hbm_bytes_corrected" in res: res["measured_hbm_GBs"] = res["hbm_bytes_corrected"] / t / 1e9
for l in open(f"{out}/{tag}_score_trace.log"):
    if l.startswith("{"): res["bench_object_under_profiler"] = json.loads(l)
json.dump(res, open(f"{out}/{tag}_score_roofline.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
