"""Stress: the same 50-step ge_train_steps call repeated from the same state; reports, per repeat, the first
step whose loss vector deviates from repeat 0 by more than 1e-4 (hot rows use float atomics: ~1e-7 is normal)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphembeddings_amd import data as D, hole as H
model = sys.argv[1] if len(sys.argv) > 1 else "hole"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
kw = {}
if len(sys.argv) > 3 and sys.argv[3] == "nolook":
    kw["lookahead"] = False
fb = D.fb15k_shape()
names, id_to_type, offsets, ids = fb.type_arrays()
B, d, steps = 4096, 200, 50
tri = torch.as_tensor(D.synthetic_fb15k_triples(fb, n_triples=7 * B + 13, seed=17)).cuda()
base = H.init_embeddings(fb.entity_count, d, seed=8)
base[::4] *= 7.0
tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
ref = None
for r in range(reps):
    emb = base.clone()
    tr = H.Trainer(emb, tri, tt, B, margin=0.2, learning_rate=0.1, decay_steps=200.0, decay_rate=0.5, model=model, seed=33, **kw)
    losses = tr.run(steps, keep_losses=True)
    torch.cuda.synchronize()
    tr.close()
    if ref is None:
        ref, ref_emb = losses.clone(), emb.clone()
        continue
    dl = (losses - ref).abs().amax(1)
    bad = torch.nonzero(dl > 1e-4)
    print(f"rep {r}: max loss diff {dl.max().item():.3e} table diff {(emb - ref_emb).abs().max().item():.3e}",
          f"first bad step {int(bad[0])} ({int((dl > 1e-4).sum())} bad steps)" if len(bad) else "")
# per-step growth of the run-to-run difference (two more runs)
outs = []
for r in range(2):
    emb = base.clone()
    tr = H.Trainer(emb, tri, tt, B, margin=0.2, learning_rate=0.1, decay_steps=200.0, decay_rate=0.5, model=model, seed=33, **kw)
    outs.append(tr.run(steps, keep_losses=True).clone())
    torch.cuda.synchronize()
    tr.close()
dl = (outs[0] - outs[1]).abs().amax(1).cpu().numpy()
print("per-step run-to-run max loss diff:", " ".join(f"{v:.1e}" for v in dl))
