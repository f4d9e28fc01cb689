import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphembeddings_amd import data as D, hole as H
fb = D.fb15k_shape()
tri = torch.as_tensor(D.synthetic_fb15k_triples(fb)).cuda()
names, id_to_type, offsets, ids = fb.type_arrays()
tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
emb = H.init_embeddings(fb.entity_count, 200)
tr = H.Trainer(emb, tri, tt, 4096)
gen = torch.Generator(device="cuda").manual_seed(0)
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("randperm ms", t(lambda: torch.randperm(tri.shape[0], device="cuda", generator=gen)))
perm = torch.randperm(tri.shape[0], device="cuda", generator=gen)
print("gather ms", t(lambda: tri[perm].contiguous()))
print("invalidate ms", t(lambda: tr.invalidate()))
print("reshuffle ms", t(lambda: tr.reshuffle(gen)))
tr.run(7)
print("run(7) after reshuffle ms", t(lambda: (tr.reshuffle(gen), tr.run(7))))
print("run(7) steady ms", t(lambda: tr.run(7)))
