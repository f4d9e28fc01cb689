"""Cycles per phase of the split-precision rank sweep (diagnostic build: GE_CXXFLAGS=-DGE_RANK_STAMPS python -m
graphembeddings_amd.build).  Sums over all waves of s_memtime differences; printed per tile per wave."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graphembeddings_amd import _lib, data as D, hole as H
lib = _lib.load()
inf = D.init_inference_data(D.PACKAGE_FB15K_DIR)
emb = H.init_embeddings(inf.entity_count, 200, seed=3) * 4.0
test = inf.test_array
cand = torch.arange(inf.relation_count, inf.entity_count, dtype=torch.int32).cuda()
hr = torch.as_tensor(np.stack([test[:, 0], test[:, 2]], 1).astype(np.int32)).cuda()
tid = torch.as_tensor(test[:, 1].astype(np.int32)).cuda()
planes = H.RankPlanes(emb, cand)
H.rank_candidates(emb, hr, tid, cand, planes=planes)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
lib.ge_debug_rank_stamps(buf, 1)
ev = H.Events(2); ev.record(0)
H.rank_candidates(emb, hr, tid, cand, planes=planes)
ev.record(1); torch.cuda.synchronize()
lib.ge_debug_rank_stamps(buf, 0)
v = list(buf)
print(f"kernel {ev.elapsed_ms(0, 1):.3f} ms")
names = ["MFMA loop (incl. its barriers)", "epilogue pieces + their barriers", "count, next chunk 0, barrier(s)"]
for g in (0, 1):
    waves = v[3 + 4 * g]
    print(f" group {g}: {waves} waves")
    for n, c in zip(names, v[4 * g:4 * g + 3]):
        print(f"  {n:36s} {c / max(waves, 1):12.0f} ticks per wave over the kernel")
