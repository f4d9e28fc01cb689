"""What the product driver's loop costs per step at FB15k shape (483,142 synthetic train triples, B=4096, d=200):
graphembeddings_amd.train.run_training with its 16 validation ticks per epoch and its pocket checkpoint, against the
bare native loop (bench.py's number).  Two run lengths; the difference is the steady state (set-up cancels).
Usage (GPU box): python tools/train_loop_probe.py [short_epochs long_epochs]"""
import os, sys, time, tempfile, shutil, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphembeddings_amd import data as D, train as T

e0 = int(sys.argv[1]) if len(sys.argv) > 1 else 20
e1 = int(sys.argv[2]) if len(sys.argv) > 2 else 220
fb = D.fb15k_shape()
fb.triples = D.synthetic_fb15k_triples(fb)
fb.triple_count = len(fb.triples)
out = tempfile.mkdtemp()
res = {}
try:
    for model in ("complex", "hole"):
        t = {}
        for ep in (e0, e0, e1):                       # the first run also pays the process's one-time costs
            od = os.path.join(out, f"run_{model}_{ep}_{len(t)}")
            FLAGS = T.build_parser().parse_args(["--data_dir", "unused", "--output_dir", od, "--batch_size", "4096",
                                                 "--embedding_dim", "200", "--num_epochs", str(ep), "--model", model])
            logs = []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = T.run_training(fb, FLAGS, log=lambda *a: logs.append(a))
            torch.cuda.synchronize()
            t[ep] = (time.perf_counter() - t0, r["steps"], sum(1 for l in logs if "Model saved" in " ".join(str(x) for x in l)))
        (s0, n0, c0), (s1, n1, c1) = t[e0], t[e1]
        res[model] = {"steady_us_per_step": (s1 - s0) / (n1 - n0) * 1e6, "steps": n1 - n0, "seconds": s1 - s0,
                      "checkpoints_written_in_long_run": c1, "setup_ms": (s0 - n0 * (s1 - s0) / (n1 - n0)) * 1e3}
    print(json.dumps(res))
finally:
    shutil.rmtree(out, ignore_errors=True)
