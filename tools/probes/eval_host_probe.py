"""Where a two-sided filtered FB15k-shaped evaluation spends its time: known index build, per-side cell lists, sweeps, transfers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphembeddings_amd import data as D, evaluate as E, hole as H
inf = D.init_inference_data(D.PACKAGE_FB15K_DIR)
emb = H.init_embeddings(inf.entity_count, 200, seed=3) * 4.0
cand = np.arange(inf.relation_count, inf.entity_count, dtype=np.int32)
test, known = inf.test_array, inf.validation_triples
def t(f, n=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
dev = emb.device
print("known triples", len(known), "test", len(test))
print("KnownIndex build  ms", t(lambda: E.KnownIndex(known, emb.shape[0], "tail", dev)))
idx = E.KnownIndex(known, emb.shape[0], "tail", dev)
c = torch.as_tensor(cand).to(dev)
pos_of = torch.full((emb.shape[0],), -1, dtype=torch.int64, device=dev); pos_of[c.long()] = torch.arange(c.numel(), device=dev)
chunk = torch.as_tensor(test).to(dev)
print("test upload       ms", t(lambda: torch.as_tensor(test).to(dev)))
print("cells()           ms", t(lambda: idx.cells(chunk[:, 0], chunk[:, 2], pos_of, c.numel())))
planes = H.RankPlanes(emb, c)
print("RankPlanes        ms", t(lambda: H.RankPlanes(emb, c)))
off, rc = idx.cells(chunk[:, 0], chunk[:, 2], pos_of, c.numel())
hr = torch.stack([chunk[:, 0], chunk[:, 2]], 1).to(torch.int32); tid = chunk[:, 1]
print("rank_candidates   ms", t(lambda: H.rank_candidates(emb, hr, tid, c, known_off=off, known_rc=rc, planes=planes)))
print("one side, all     ms", t(lambda: E.link_prediction_ranks(emb, test, cand, known, side="tail", planes=planes)))
