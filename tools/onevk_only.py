"""Runs only the 1-vs-K sweep a few times (for rocprofv3 passes) and prints its back-to-back time per call.
Default: the FB15k evaluation shape; B=4096 K=256 = BASELINE config 5's own shape."""
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
ev = H.Events(2)
ev.record(0)
for _ in range(50):
    out = H.score_candidates(emb, hr, cand)
ev.record(1)
torch.cuda.synchronize()
us = ev.elapsed_ms(0, 1) / 50 * 1e3
print("ok", float(out[0, 0]), f"{us:.2f} us per call back to back = {2.0 * B * K * d / us / 1e6:.1f} TFLOP/s")
