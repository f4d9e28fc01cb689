import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from graphembeddings_amd import hole as H, data as D
fb = D.fb15k_shape()
d, B = 200, 4096
emb = H.init_embeddings(fb.entity_count, d)
tri = torch.as_tensor(D.synthetic_fb15k_triples(fb, n_triples=B, seed=0)).cuda()
names, id_to_type, offsets, ids = fb.type_arrays()
tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
neg = H.corrupt_batch(tt, fb.relation_count, tri, seed=0, step=0)
for margin in (1.0, 0.0, -1.0):
    for _ in range(5): out = H.hinge_grad(emb, tri, neg, 0.1, margin=margin, model="hole")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): out = H.hinge_grad(emb, tri, neg, 0.1, margin=margin, model="hole")
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    act = (out[1].view(-1, 6)[:, 0] >= 0).float().mean().item()
    print(f"margin {margin:+.1f}: active pairs {act:.2f}, hole hinge_grad {dt*1e6:.1f} us/call")
