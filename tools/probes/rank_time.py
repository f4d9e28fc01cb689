"""Event-timed fused rank sweep (FB15k test-set shape, planes built once): python tools/probes/rank_time.py [d [table scale]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphembeddings_amd import data as D, hole as H
d = int(sys.argv[1]) if len(sys.argv) > 1 else 200
inf = D.init_inference_data(D.PACKAGE_FB15K_DIR)
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
emb = H.init_embeddings(inf.entity_count, d, seed=3) * scale
test = inf.test_array
hr = torch.as_tensor(np.stack([test[:, 0], test[:, 2]], 1).astype(np.int32)).cuda()
tid = torch.as_tensor(test[:, 1].astype(np.int32)).cuda()
c = torch.arange(inf.relation_count, inf.entity_count, dtype=torch.int32, device="cuda")
planes = H.RankPlanes(emb, c)
for _ in range(2):
    H.rank_candidates(emb, hr, tid, c, planes=planes)
ev = H.Events(2)
ts = []
for _ in range(5):
    ev.record(0)
    H.rank_candidates(emb, hr, tid, c, planes=planes)
    ev.record(1)
    torch.cuda.synchronize()
    ts.append(ev.elapsed_ms(0, 1))
print(f"d={d} sweep ms: min {min(ts):.3f} median {sorted(ts)[2]:.3f}  ({2.0 * len(test) * c.numel() * d / (sorted(ts)[2] * 1e-3) / 1e12:.0f} TFLOP/s fp32-equivalent)")
