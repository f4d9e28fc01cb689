"""Host time (no synchronisation) of the pieces of one validation tick of train.run_training."""
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
valid = torch.as_tensor(fb.validation_triples).cuda()
gen = torch.Generator(device="cuda").manual_seed(0)
pocket = torch.empty_like(emb); best = torch.full((), 2.0, device="cuda")
ring = torch.empty(4096, pin_memory=True)
acc = {}
def T(name, fn):
    t0 = time.perf_counter(); r = fn(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return r
n = 400
tr.run(7); torch.cuda.synchronize()
t_all = time.perf_counter()
for i in range(n):
    sel = T("randint", lambda: torch.randint(0, valid.shape[0], (4096,), device="cuda", generator=gen))
    vb = T("index", lambda: valid[sel])
    neg = T("corrupt", lambda: H.corrupt_batch(tt, fb.relation_count, vb, seed=1, step=i))
    hl = T("hinge_loss", lambda: H.hinge_loss(vb, neg, emb, margin=0.2))
    vl = T("mean", lambda: hl.mean())
    T("where", lambda: torch.where(vl < best, emb, pocket, out=pocket))
    T("minimum", lambda: torch.minimum(vl, best, out=best))
    T("host copy", lambda: ring[i].copy_(vl, non_blocking=True))
    T("event", lambda: torch.cuda.Event().record())
    T("run(7)", lambda: tr.run(7))
host = time.perf_counter() - t_all
torch.cuda.synchronize()
tot = time.perf_counter() - t_all
print({k: round(v / n * 1e6, 1) for k, v in acc.items()}, "us per tick (host)")
print(f"host loop {host/n*1e6:.1f} us per tick, with device {tot/n*1e6:.1f} us per tick = {tot/n/7*1e6:.1f} us/step")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(20):
        sel = torch.randint(0, valid.shape[0], (4096,), device="cuda", generator=gen)
        vb = valid[sel]
        neg = H.corrupt_batch(tt, fb.relation_count, vb, seed=1, step=i)
        hl = H.hinge_loss(vb, neg, emb, margin=0.2)
        vl = hl.mean()
        torch.where(vl < best, emb, pocket, out=pocket)
        torch.minimum(vl, best, out=best)
        ring[i].copy_(vl, non_blocking=True)
        tr.run(7)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70))
