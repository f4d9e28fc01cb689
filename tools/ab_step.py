"""A/B of the config-2 step between two checkouts on ONE box: runs each tree's bench (headline leg only) alternately."""
import json, os, subprocess, sys
trees = sys.argv[1:]
for rep in range(3):
    for t in trees:
        out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-scaling-base", "--no-score-roofline"] +
                             (["--no-hbm-roofline"] if os.path.exists(os.path.join(t, "graphembeddings_amd/csrc/ge_shard.hip")) else []),
                             cwd=t, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(t, "FAILED", out.stderr[-500:]); continue
        d = json.loads(line[-1])
        print(rep, t, "us/step", round(d["ms_per_step"] * 1e3, 2), "kernels_ms", d["roofline"].get("all_kernels_ms"), flush=True)
