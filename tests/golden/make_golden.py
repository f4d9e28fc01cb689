"""Generates tests/golden/golden_v1.npz from the NumPy fp64 restatement (oracle/hole_oracle.py).

The reference (holE.py) cannot be imported here -- `import tensorflow` raises ModuleNotFoundError --
and holds no golden vectors of its own, so these fixtures are produced by the restatement and are
cross-checked against an independent torch-autograd implementation in tests/test_oracle.py.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import hole_oracle as O  # noqa: E402


def make_table(rng, n, d, relations):
    """fp32 table whose row norms span [0.3, 2.2] so the max-norm clip is exercised on ~half the
    rows (default init gives norms ~0.18 and never clips); row 0 is all-zero, row 1 has norm
    exactly 1 (MinimumGrad tie), row 2 is a tiny-norm row."""
    t = rng.standard_normal((n, d))
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    t *= rng.uniform(0.3, 2.2, size=(n, 1))
    t[0] = 0.0
    e = np.zeros(d); e[3] = 1.0
    t[1] = e
    t[2] *= 1e-3 / np.linalg.norm(t[2])
    t[3] = -2.0 * e                    # clipped to -e_3: triple (1,1,3) scores -1, (1,1,1) scores +1
    return t.astype(np.float32)


def make_triples(rng, n, relations, b):
    h = rng.integers(relations, n, size=b)
    t = rng.integers(relations, n, size=b)
    r = rng.integers(0, relations, size=b)
    tr = np.stack([h, t, r], axis=1).astype(np.int32)
    tr[0] = [0 + relations, 1, 2]      # includes the zero/unit/tiny rows via low ids
    tr[1] = [0, 1, 2]
    tr[2] = [1, 1, 1]
    tr[3] = tr[4]                      # duplicate triple
    tr[8] = [1, 1, 3]
    return tr


def main():
    out = {}
    rng = np.random.default_rng(20171)
    for d in (50, 128, 200):
        n, rel, b = 64, 8, 64
        table = make_table(rng, n, d, rel)
        pos = make_triples(rng, n, rel, b)
        neg = pos.copy()
        heads = rng.random(b) < 0.5
        neg[heads, 0] = rng.integers(rel, n, size=int(heads.sum()))
        neg[~heads, 1] = rng.integers(rel, n, size=int((~heads).sum()))
        neg[5] = pos[5]                # corrupted id equal to the original (holE.py:112 allows it)
        neg[6, 0] = pos[7, 0]          # duplicates across pairs
        t64 = table.astype(np.float64)
        out[f"d{d}_table"] = table
        out[f"d{d}_pos"] = pos
        out[f"d{d}_neg"] = neg
        out[f"d{d}_score_raw"] = O.complex_score(pos, t64)
        out[f"d{d}_sigma"] = O.evaluate_triples(pos, t64)[:, 0]
        out[f"d{d}_hole_raw"] = O.hole_score(pos, t64)
        for margin in (0.2, 0.0, -0.5):  # 0.0: mixed active/inactive + an exact tie; -0.5: all inactive (holE.py:231)
            key = f"d{d}_m{margin}"
            new, loss = O.sgd_step(t64, pos, neg, lr=0.05, margin=margin)
            out[key + "_loss"] = loss
            if margin != -0.5:
                out[key + "_table_after"] = new.astype(np.float32)
            newh, lossh = O.sgd_step(t64, pos, neg, lr=0.05, margin=margin, model="hole")
            out[key + "_hole_loss"] = lossh
            if margin != -0.5:
                out[key + "_hole_table_after"] = newh.astype(np.float32)
    # LR schedule (holE.py:291-295), defaults of holE.py:599-601 at FB15k scale
    steps = np.array([0, 1, 10, 943, 30176, 100000], dtype=np.int64)
    out["lr_steps"] = steps
    out["lr_values"] = np.array([O.inverse_time_decay(0.1, s, 32 * 943, 0.5) for s in steps])
    # sampler known answers
    n_types = 5
    sizes = [1, 7, 300, 2, 40]
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    ids = np.arange(8, 8 + offsets[-1], dtype=np.int32)
    rng.shuffle(ids)
    id_to_type = np.full(8 + offsets[-1] + 3, -1, dtype=np.int32)   # 3 trailing unknown ids
    for ty in range(n_types):
        id_to_type[ids[offsets[ty]:offsets[ty + 1]]] = ty
    b = 257
    pos = np.stack([rng.choice(ids, b), rng.choice(ids, b), rng.integers(0, 8, b)], axis=1).astype(np.int32)
    pos[10, 0] = 8 + offsets[-1] + 1   # unknown type -> -1
    pos[11, 1] = 8 + offsets[-1] + 2
    out["smp_id_to_type"], out["smp_offsets"], out["smp_ids"], out["smp_pos"] = id_to_type, offsets, ids, pos
    for mode in range(4):
        for (seed, step, padded) in ((0, 0, 1024), (0x1234567890ABCDEF, 77, 16), (5, 2**40 + 3, 0)):
            out[f"smp_neg_m{mode}_s{seed}_t{step}_p{padded}"] = O.corrupt_batch(
                pos, id_to_type, offsets, ids, seed, step, padded, mode)
    # Philox4x32-10 known answers (Random123 kat_vectors)
    kat_in = np.array([[0, 0, 0, 0, 0, 0],
                       [0xFFFFFFFF] * 6,
                       [0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0]], dtype=np.uint64)
    out["philox_kat_in"] = kat_in
    out["philox_kat_out"] = np.array([[int(v) for v in O.philox4x32_10(*[int(x) for x in row])] for row in kat_in],
                                     dtype=np.uint64)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
