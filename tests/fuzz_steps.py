"""Randomised parity sweep of the native training loop's multi-tile steps against the C port (test infrastructure: uses the
oracle; not collected by pytest -- run `python tests/fuzz_steps.py [n_cases] [seed]` on a GPU box).  Shapes the fixed tests do not
pin one by one: embedding_dim from 8 to 512, batch sizes that are no multiple of anything, tables so small that a tile holds more
work items than the update kernel launches waves (its item loop takes several rounds), 1 ... 1,345 relations (a workgroup's
relation runs end on one relation or on many), margins that leave every pair / no pair hinge-active, the deterministic flag."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    import torch
    from graphembeddings_amd import data as D
    from graphembeddings_amd import hole as H
    from oracle import c_oracle as CO
    from oracle import hole_oracle as O
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261005)
    worst, fails = {"loss": 0.0, "table": 0.0}, []
    for case in range(n_cases):
        d = int(rng.choice([8, 50, 64, 104, 128, 200, 200, 200, 256, 300, 512]))
        n_ent = int(rng.choice([700, 20_000, 50_000, 300_000]))
        n_rel = int(rng.choice([1, 3, 18, 1345]))
        B = int(rng.integers(4097, 40_000 if d <= 256 else 12_000))
        margin = float(rng.choice([0.2, 0.2, -1.0, 5.0]))
        det = bool(rng.integers(0, 2)) and d <= 1024
        steps = 2
        data, tri = D.synthetic_large(n_entities=n_ent, n_relations=n_rel, n_types=int(rng.choice([2, 12])), n_triples=2 * B + 7,
                                      seed=int(rng.integers(1 << 30)), zipf_s=float(rng.choice([0.0, 0.8, 1.1])))
        names, id_to_type, offsets, ids = D.synthetic_large_type_arrays(data)
        N = data.entity_count
        table = (rng.standard_normal((N, d)) * 0.08).astype(np.float32)
        table[::3] *= 6.0                                            # rows on both sides of the max-norm clip
        tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
        emb = torch.as_tensor(table).cuda().clone()
        seed = int(rng.integers(1 << 20))
        tr = H.Trainer(emb, torch.as_tensor(tri).cuda(), tt, B, margin=margin, learning_rate=0.1, decay_steps=40.0, decay_rate=0.5,
                       seed=seed, deterministic=det)
        losses = tr.run(steps, keep_losses=True).cpu().numpy()
        ctab, row, dl = table.copy(), 0, 0.0
        t64 = table.astype(np.float64)
        for s in range(steps):
            if row + B > len(tri):
                row = 0
            pos = tri[row:row + B]
            neg = CO.corrupt_batch(pos, id_to_type, offsets, ids, seed, s, 1024, 0)
            lr = np.float32(0.1) / (np.float32(1.0) + np.float32(0.5) * (np.float32(s) / np.float32(40.0)))
            closs = CO.hinge_step(ctab, pos, neg, margin, float(lr), threads=16)
            t64, _ = O.sgd_step(t64, pos, neg, float(lr), margin)   # the same step in fp64 (NumPy restatement): the judge between two fp32 sums
            dl = max(dl, float(np.abs(losses[s] - closs).max()))
            row += B
        gtab = emb.cpu().numpy()
        dt = float(np.abs(gtab - ctab).max())
        # A row that took tens of thousands of gradient terms in a step (one relation for every pair) carries the rounding of
        # that fp32 sum: the C port adds the terms one after the other (ScatterSub's order), the kernels add partial sums of
        # partial sums.  Where the two fp32 tables differ by more than the usual bound, the fp64 restatement decides: the GPU
        # table must be at least as close to it as the C port's (tests/hot_row_rounding.py: 5e-8 / 2e-6 against 3e-5).
        err_g, err_c = float(np.abs(gtab - t64).max()), float(np.abs(ctab - t64).max())
        worst_row = int(np.abs(gtab - ctab).max(1).argmax())
        same_neg = bool(np.array_equal(tr._neg.cpu().numpy(), neg))
        tr.close()
        rec = {"case": case, "d": d, "B": B, "n_ent": n_ent, "n_rel": n_rel, "margin": margin, "det": det, "loss_diff": dl, "table_diff": dt,
               "gpu_vs_fp64": err_g, "c_port_vs_fp64": err_c, "worst_row": worst_row,
               "negatives_equal": same_neg}
        print(json.dumps(rec), flush=True)
        worst["loss"], worst["table"] = max(worst["loss"], dl), max(worst["table"], dt)
        if not (dl < 2e-5 and (dt < 1e-4 or err_g <= max(err_c, 2e-5)) and same_neg):
            fails.append(rec)
    print(json.dumps({"cases": n_cases, "worst": worst, "failed": fails}))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
