"""Just the fused rank sweep (FB15k test-set shape), for rocprofv3: python tools/rank_only.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphembeddings_amd import data as D, hole as H
inf = D.init_inference_data(D.PACKAGE_FB15K_DIR)
emb = H.init_embeddings(inf.entity_count, 200, seed=3) * 4.0
test = inf.test_array
hr = torch.as_tensor(np.stack([test[:, 0], test[:, 2]], 1).astype(np.int32)).cuda()
tid = torch.as_tensor(test[:, 1].astype(np.int32)).cuda()
c = torch.arange(inf.relation_count, inf.entity_count, dtype=torch.int32, device="cuda")
planes = H.RankPlanes(emb, c)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    H.rank_candidates(emb, hr, tid, c, planes=planes)
torch.cuda.synchronize()
