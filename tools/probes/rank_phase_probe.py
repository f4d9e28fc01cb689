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
H.rank_candidates(emb, hr, tid, cand)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
lib.ge_debug_rank_stamps(buf, 1)
ev = H.Events(2); ev.record(0)
H.rank_candidates(emb, hr, tid, cand)
ev.record(1); torch.cuda.synchronize()
lib.ge_debug_rank_stamps(buf, 0)
v = list(buf)
tiles = v[7]            # summed over waves: tiles x 8 waves x (blocks)
names = ["norm+stash slot 0", "barrier 1", "MFMA loop (incl. its barriers)", "barrier behind the loop", "epilogue", "barrier behind it", "count"]
print(f"kernel {ev.elapsed_ms(0, 1):.3f} ms; tile-waves {tiles}")
tot = 0
for n, c in zip(names, v[:7]):
    print(f"  {n:34s} {c / tiles:9.0f} cycles per tile per wave")
    tot += c / tiles
print(f"  {'sum':34s} {tot:9.0f}")
