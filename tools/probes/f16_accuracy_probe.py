"""Accuracy of the split-precision sweep: losses from ge_rank_1vK(return_scores) at d=200 against the per-triple fp32
kernel and the fp64 oracle, on tables of different scales (incl. rows far outside the unit ball)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphembeddings_amd import hole as H
from oracle import hole_oracle as O
N, R, d, B = 4000, 50, 200, 256
g = torch.Generator(device="cuda").manual_seed(0)
for scale in (0.02, 0.1, 1.0, 30.0, 3000.0, 1e-6):
    emb = torch.randn(N, d, device="cuda", generator=g) * scale
    emb[::5] *= 0.05
    hr = torch.stack([torch.randint(R, N, (B,), device="cuda", generator=g), torch.randint(0, R, (B,), device="cuda", generator=g)], 1).int()
    c = torch.arange(R, N, dtype=torch.int32, device="cuda")
    tid = c[torch.randint(0, c.numel(), (B,), device="cuda", generator=g)]
    sc = H.rank_candidates(emb, hr, tid, c, return_scores=True)[2]
    K = c.numel()
    tr = torch.stack([hr[:, 0].repeat_interleave(K), c.repeat(B), hr[:, 1].repeat_interleave(K)], 1)
    per = H.evaluate_triples(tr, emb).view(B, K)
    raw_per = H.evaluate_triples(tr, emb, apply_sigmoid=False).view(B, K)
    t64 = emb.double().cpu().numpy()
    ref = O.evaluate_triples(tr[:20 * K].cpu().numpy(), t64)[:, 0].reshape(20, K)
    print(f"scale {scale}: max|sweep - per-triple fp32| = {(sc - per).abs().max().item():.2e}   max|sweep - fp64 oracle| = "
          f"{np.abs(sc[:20].cpu().numpy() - ref).max():.2e}   max|per-triple - oracle| = {np.abs(per[:20].cpu().numpy() - ref).max():.2e}   "
          f"score range [{raw_per.min().item():.3f}, {raw_per.max().item():.3f}]")
