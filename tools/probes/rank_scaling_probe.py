"""Time of the fused rank sweep vs embedding_dim (same 59,071 x 14,951 problem): per-tile fixed cost vs per-k cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphembeddings_amd import _lib
if os.environ.get("GE_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["GE_LIB"])
from graphembeddings_amd import hole as H
N, R, B = 16296, 1345, 59071
g = torch.Generator(device="cuda").manual_seed(0)
hr = torch.stack([torch.randint(R, N, (B,), device="cuda", generator=g), torch.randint(0, R, (B,), device="cuda", generator=g)], 1).int()
tid = torch.randint(R, N, (B,), device="cuda", generator=g).int()
c = torch.arange(R, N, dtype=torch.int32, device="cuda")
# the table's spread decides how often a candidate's raw score falls inside the bracket of a true score (the exact
# comparison path of ge_rank_pipe.hip): 0.1 is the tightest case (near-ties in most tiles), 1.0 a trained table's
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
for d in (8, 40, 104, 200, 232):
    emb = torch.randn(N, d, device="cuda", generator=g) * scale
    H.rank_candidates(emb, hr, tid, c)
    ev = H.Events(2); ev.record(0)
    for _ in range(3):
        H.rank_candidates(emb, hr, tid, c)
    ev.record(1); torch.cuda.synchronize()
    ms = ev.elapsed_ms(0, 1) / 3; ev.close()
    print(f"d={d:4d}  {ms:7.3f} ms  {2.0*B*(N-R)*d/(ms*1e-3)/1e12:6.1f} TFLOP/s")
