#!/bin/bash
# Runs on the GPU box: everything profiles/ holds for one round, under gpurun_out/ with the round tag.
# Usage: bash tools/profile_round.sh r04
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd "$ROOT"
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/${TAG}_smoke.txt" 2>&1
timeout -k 10 600 python bench.py > "$OUT/${TAG}_bench_default.log" 2>&1 && tail -1 "$OUT/${TAG}_bench_default.log" > "$OUT/${TAG}_bench_default_line.json"
echo "bench default done"
timeout -k 10 400 python bench.py --sharded --steps 128 --warmup 64 > "$OUT/${TAG}_sharded_world1.log" 2>&1 && tail -1 "$OUT/${TAG}_sharded_world1.log" > "$OUT/${TAG}_sharded_world1_line.json"
echo "sharded world1 done"
bash tools/profile_bench.sh ${TAG}_fb15k_d200_b4096 --no-score-roofline > "$OUT/${TAG}_prof_fb15k.log" 2>&1
echo "profile fb15k done"
bash tools/profile_bench.sh ${TAG}_synthetic_d200_b65536 --workload synthetic --batch 65536 --steps 64 --warmup 16 --no-score-roofline > "$OUT/${TAG}_prof_synth.log" 2>&1
echo "profile synthetic done"
bash tools/profile_bench.sh ${TAG}_fb15k_hole_d200_b4096 --model hole --no-score-roofline > "$OUT/${TAG}_prof_hole.log" 2>&1
echo "profile hole (config 3) done"
bash tools/profile_score.sh ${TAG} > "$OUT/${TAG}_prof_score.log" 2>&1
echo "profile score kernel done"
( cd /tmp && export TMPDIR=/tmp && B=4096 K=256 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_onevk_trace" -- python3 "$ROOT/tools/onevk_only.py" > "$OUT/${TAG}_onevk.log" 2>&1 )
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys
out, tag = sys.argv[1], sys.argv[2]
with open(f"{out}/{tag}_onevk_profile.txt", "w") as f:
    f.write("# B=4096 K=256 rocprofv3 --kernel-trace --stats -- python3 tools/onevk_only.py   (BASELINE config 5's own shape: 4096 x 256 x d=200, 419 MFLOP)\n")
    for p in glob.glob(f"{out}/{tag}_onevk_trace/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(p)):
            if "1vK" in row["Name"] or "score" in row["Name"]:
                f.write(f"{row['Name'][:70]}  calls {row['Calls']}  avg_ns {row['AverageNs']}  min_ns {row['MinNs']}  max_ns {row['MaxNs']}\n")
    f.write([l for l in open(f"{out}/{tag}_onevk.log").read().splitlines() if l.startswith("ok")][-1] + "\n")
PY
echo "onevk done"
bash tools/profile_rank.sh ${TAG} > "$OUT/${TAG}_prof_rank.log" 2>&1
echo "rank done"
timeout -k 10 300 python tools/rank_bench.py > "$OUT/${TAG}_rank_bench.json" 2>"$OUT/${TAG}_rank_bench.err"
echo "rank bench done"
# no launcher: bench.py starts its own two ranks (graphembeddings_amd/launch.py); both on this box's one GPU, collectives through gloo
GE_DIST_BACKEND=gloo GE_SINGLE_DEVICE=1 timeout -k 10 400 python bench.py --gpus 2 --steps 16 --warmup 16 --batch 16384 > "$OUT/${TAG}_gloo2_rehearsal.log" 2>&1
tail -1 "$OUT/${TAG}_gloo2_rehearsal.log" | cut -c1-300
echo "all done"
