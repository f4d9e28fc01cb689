"""Runs only the 1-vs-K sweep at the FB15k evaluation shape a few times (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphembeddings_amd import hole as H
N, d, B, K = 16296, 200, int(os.environ.get("B", 59071)), int(os.environ.get("K", 14951))
emb = H.init_embeddings(N, d)
g = torch.Generator().manual_seed(0)
hr = torch.stack([torch.randint(1345, N, (B,), generator=g), torch.randint(0, 1345, (B,), generator=g)], 1).int().cuda()
cand = torch.arange(1345, 1345 + K).int().cuda()
for _ in range(4):
    out = H.score_candidates(emb, hr, cand)
torch.cuda.synchronize()
print("ok", float(out[0, 0]))
