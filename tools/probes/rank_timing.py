import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from graphembeddings_amd import _lib
if os.environ.get("GE_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["GE_LIB"])
from graphembeddings_amd import hole as H
N, R, B = 16296, 1345, 59071
g = torch.Generator(device="cuda").manual_seed(0)
hr = torch.stack([torch.randint(R, N, (B,), device="cuda", generator=g), torch.randint(0, R, (B,), device="cuda", generator=g)], 1).int()
tid = torch.randint(R, N, (B,), device="cuda", generator=g).int()
c = torch.arange(R, N, dtype=torch.int32, device="cuda")
for d in (200,):
    emb = torch.randn(N, d, device="cuda", generator=g) * 0.1
    H.rank_candidates(emb, hr, tid, c)
    out = H.rank_candidates(emb, hr, tid, c, return_true_loss=True)
    torch.cuda.synchronize()
    v = out[2][:7].cpu().numpy()
    print(f"d={d} setup {v[0]:.0f} diag {v[1]:.0f} ({v[6]:.0f} segs) tiles {v[2]:.0f} ({v[2]/v[4]:.0f}/tile) epi {v[3]:.0f} ({v[3]/v[4]:.0f}/tile) epi2 {v[5]/v[4]:.0f}/tile n_tiles {v[4]:.0f}  [ticks]")
# the same with every true tail the best-scoring candidate (few scores inside a bracket)
d = 200
emb = torch.randn(N, d, device="cuda", generator=g) * 0.1
best = []
for s0 in range(0, B, 8192):
    sc = H.score_candidates(emb, hr[s0:s0 + 8192], c)
    best.append(c[sc.argmin(1)]); del sc
tid_top = torch.cat(best).int()
H.rank_candidates(emb, hr, tid_top, c)
out = H.rank_candidates(emb, hr, tid_top, c, return_true_loss=True)
torch.cuda.synchronize()
v = out[2][:7].cpu().numpy()
print(f"true-at-top d={d} tiles {v[2]/v[4]:.0f}/tile epi {v[3]/v[4]:.0f}/tile epi2 {v[5]/v[4]:.0f}/tile  [ticks]")
