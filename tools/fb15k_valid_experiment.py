#!/usr/bin/env python3
"""End-to-end run on REAL FB15k id files (the 50,000-triple validation split shipped with the
package; the train split is not available, .MISSING_LARGE_BLOBS:1-3): train on 45k, hold out 5k,
report filtered MRR / Hits@n with the reference's ranking semantics.  Writes a JSON line."""
import json, os, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphembeddings_amd import data as D, train as T, evaluate as E

fb = D.fb15k_shape()
rng = np.random.default_rng(0)
tri = fb.validation_triples.copy()
rng.shuffle(tri)
train, held = tri[:45000], tri[45000:]
seen = np.zeros(fb.entity_count, bool); seen[train[:, 0]] = True; seen[train[:, 1]] = True
rel_seen = np.zeros(fb.entity_count, bool); rel_seen[train[:, 2]] = True
held = held[seen[held[:, 0]] & seen[held[:, 1]] & rel_seen[held[:, 2]]]
tmp = tempfile.mkdtemp()
for f in ("entity_metadata.tsv.gz", "relation_ids.txt.gz"):
    os.symlink(os.path.join(D.PACKAGE_FB15K_DIR, f), os.path.join(tmp, f))
np.savetxt(os.path.join(tmp, "triples.txt"), train, fmt="%d", delimiter="\t")
np.savetxt(os.path.join(tmp, "triples-valid.txt"), train[:2048], fmt="%d", delimiter="\t")
np.savetxt(os.path.join(tmp, "test_positive_triples.txt"), held, fmt="%d", delimiter="\t")
out = os.path.join(tmp, "run")
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
lr = sys.argv[2] if len(sys.argv) > 2 else "0.1"
argv = ["--data_dir", tmp, "--output_dir", out, "--batch_size", "512", "--embedding_dim", "200",
        "--num_epochs", str(epochs), "--learning_rate", lr, "--margin", "0.2", "--seed", "0", "--learning_decay_steps", str(max(32, epochs // 4))]
FLAGS = T.build_parser().parse_args(argv)
data = D.init_data(tmp)
t0 = time.time()
res = T.run_training(data, FLAGS, log=lambda *a: None)
torch.cuda.synchronize()
train_s = time.time() - t0
emb, step = T.load_checkpoint(out)
inf = D.init_inference_data(tmp)
t1 = time.time()
m = E.evaluate_fb15k_style(emb, inf, both_sides=True, verbose=False)
eval_s = time.time() - t1
emb0 = torch.randn_like(emb) * 0.01
m0 = E.evaluate_fb15k_style(emb0, inf, both_sides=True, verbose=False)
print(json.dumps({"dataset": "FB15k validation split (real ids/types), 45,000 train / %d held out" % len(held),
                  "model": "complex d=200 B=512 hinge margin 0.2 lr %s type-safe negatives" % lr, "epochs": epochs,
                  "steps": res["steps"], "train_seconds": round(train_s, 2),
                  "scored_triples_per_s_incl_validation_and_checkpoints": round(2 * 512 * res["steps"] / train_s),
                  "pocket_validation_hinge": round(res["pocket_loss"], 5), "checkpoint_step": step,
                  "eval_seconds": round(eval_s, 2), "metrics": {k: round(v, 4) for k, v in m.items()},
                  "untrained_metrics": {k: round(v, 4) for k, v in m0.items()}}))
