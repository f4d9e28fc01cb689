#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats + separate PMC passes for `bench.py <args>`,
# condensed into gpurun_out/<tag>_{kernel_stats.csv,pmc.json}.  Usage: tools/profile_bench.sh TAG [bench args...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -- python3 "$ROOT/bench.py" "$@" --no-cpu-baseline --no-scaling-base --no-hbm-roofline --no-rank-roofline --no-extra-configs > "$OUT/${TAG}_trace.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_$C" -- python3 "$ROOT/bench.py" "$@" --no-cpu-baseline --no-scaling-base --no-hbm-roofline --no-rank-roofline --no-extra-configs > "$OUT/${TAG}_pmc_$C.log" 2>&1
done
python3 - "$OUT" "$TAG" "$*" <<'PY'
import csv, glob, json, re, sys, collections
out, tag, args = sys.argv[1], sys.argv[2], sys.argv[3]
stats = glob.glob(f"{out}/{tag}_trace/*/*_kernel_stats.csv")
with open(f"{out}/{tag}_kernel_stats.csv", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py {args} --no-cpu-baseline --no-scaling-base --no-hbm-roofline --no-rank-roofline --no-extra-configs\n")
    if stats:
        for i, row in enumerate(csv.reader(open(stats[0]))):
            row[0] = re.sub(r"\(.*", "", row[0])[:100]
            if i == 0 or "ge::" in row[0]:
                f.write(",".join(row) + "\n")
bench_line = [l for l in open(f"{out}/{tag}_trace.log") if l.startswith("{")]
pmc = {"_command": f"rocprofv3 --pmc <C> --kernel-trace -- python3 bench.py {args} --no-cpu-baseline --no-scaling-base --no-hbm-roofline --no-rank-roofline --no-extra-configs (one pass per counter)",
       "_units": "counter values in KB per launch (mean over launches); FETCH_SIZE doubled in hbm_bytes_corrected per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads)"}
agg = collections.defaultdict(dict)
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"{out}/{tag}_pmc_{C}/*/*_counter_collection.csv")
    if not fs: continue
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] == C and "ge::" in r["Kernel_Name"]:
            vals[re.sub(r"\(.*", "", r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in vals.items():
        agg[k][C + "_KB"] = round(sum(v) / len(v), 1); agg[k]["launches"] = len(v)
for k, v in agg.items():
    v["hbm_bytes_corrected"] = int((2 * v.get("FETCH_SIZE_KB", 0) + v.get("WRITE_SIZE_KB", 0)) * 1024)
pmc["kernels"] = agg
if bench_line: pmc["bench_line_under_profiler"] = json.loads(bench_line[-1])
json.dump(pmc, open(f"{out}/{tag}_pmc.json", "w"), indent=1)
print(open(f"{out}/{tag}_kernel_stats.csv").read())
print(json.dumps(agg, indent=1))
PY
