"""Rows whose fused rank counts differ between the rank-only sweep and the scores-storing sweep (debug aid)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graphembeddings_amd import hole as H
d = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(1)
N, B = 405, 128
table = (rng.standard_normal((N, d)) * 0.2).astype(np.float32)
emb = torch.as_tensor(table).cuda()
hr = torch.as_tensor(np.stack([rng.integers(5, N, B), rng.integers(0, 5, B)], 1).astype(np.int32)).cuda()
tid = torch.as_tensor(rng.integers(5, N, B).astype(np.int32)).cuda()
cand = torch.arange(5, N, dtype=torch.int32).cuda()
nb0, nk0 = H.rank_candidates(emb, hr, tid, cand)[:2]
nb1, nk1, sc = H.rank_candidates(emb, hr, tid, cand, return_scores=True)
col = (cand.view(1, -1) == tid.view(-1, 1)).float().argmax(1)
st = sc.gather(1, col.view(-1, 1))
ref = ((sc < st) | ((sc == st) & (cand.view(1, -1) < tid.view(-1, 1)))).sum(1).int()
print("scores-mode counts == counts from scores:", bool(torch.equal(nb1, ref)))
bad = (nb0 != ref).nonzero().flatten().cpu().numpy()
print("rank-only rows differing:", bad.tolist())
print("got", nb0.cpu().numpy()[bad][:16], "want", ref.cpu().numpy()[bad][:16])
