"""Randomised check of the 1-vs-K contraction (ge_complex_score_1vK: MFMA tile kernels) against the per-triple score kernel -- which
the fixed tests pin to the fp64 oracle -- over embedding_dim 8 ... 288, ragged B and K, both corruption sides, raw scores and
sigmoid, invalid ids.  Not collected by pytest: `python tests/fuzz_onevk.py [n_cases] [seed]` on a GPU box."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    import torch
    from graphembeddings_amd import hole as H
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    worst, fails = 0.0, []
    for case in range(n_cases):
        d = int(rng.choice([8, 16, 40, 56, 64, 104, 120, 200, 200, 208, 224, 232, 256, 288]))
        N = int(rng.choice([300, 5000, 60000]))
        B = int(rng.integers(1, 700)) if rng.random() < 0.7 else int(rng.integers(3000, 9000))
        K = int(rng.integers(1, 300)) if rng.random() < 0.7 else int(rng.integers(1000, 4000))
        head = bool(rng.integers(0, 2))
        emb = torch.as_tensor((rng.standard_normal((N, d)) * rng.choice([0.05, 0.3])).astype(np.float32)).cuda()
        hr = torch.as_tensor(np.stack([rng.integers(0, N, B), rng.integers(0, N, B)], 1).astype(np.int32)).cuda()
        cand = torch.as_tensor(rng.integers(0, N, K).astype(np.int32)).cuda()
        bad_row, bad_col = int(rng.integers(0, B)), int(rng.integers(0, K))
        if rng.random() < 0.5:
            hr[bad_row, 0] = N + 3
            cand[bad_col] = -2
        out = H.score_candidates(emb, hr, cand, cand_is_head=head)
        # the same cells through the per-triple kernel: 4,000 random (row, candidate) pairs
        ii = torch.as_tensor(rng.integers(0, B, 4000)).cuda()
        jj = torch.as_tensor(rng.integers(0, K, 4000)).cuda()
        fixed, rel, c = hr[ii, 0], hr[ii, 1], cand[jj]
        tri = torch.stack([c, fixed, rel], 1) if head else torch.stack([fixed, c, rel], 1)
        valid = (fixed >= 0) & (fixed < N) & (c >= 0) & (c < N)
        ref = H.evaluate_triples(torch.where(valid.unsqueeze(1), tri, torch.zeros_like(tri)), emb)[:, 0]
        got = out[ii, jj]
        nan_ok = bool(torch.isnan(got[~valid]).all()) if (~valid).any() else True
        err = float((got[valid] - ref[valid]).abs().max()) if valid.any() else 0.0
        rec = {"case": case, "d": d, "N": N, "B": B, "K": K, "cand_is_head": head, "max_abs_diff": err, "invalid_cells_are_nan": nan_ok}
        print(json.dumps(rec), flush=True)
        worst = max(worst, err)
        if not (err < 1e-5 and nan_ok):
            fails.append(rec)
    print(json.dumps({"cases": n_cases, "worst_abs_diff": worst, "failed": fails}))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
