#!/usr/bin/env python3
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphembeddings_amd import hole as H, data as D, _lib
d, B = 200, int(sys.argv[1]) if len(sys.argv) > 1 else 64
fb = D.fb15k_shape()
names, id_to_type, offsets, ids = fb.type_arrays()
tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
tri = D.synthetic_fb15k_triples(fb, n_triples=5 * B + 77, seed=11)
base = H.init_embeddings(fb.entity_count, d, seed=1)
_lib.load().ge_set_fused_step(1)
emb = base.clone()
tr = H.Trainer(emb, torch.as_tensor(tri).cuda(), tt, B, seed=21)
tr._ws.fill_(0x55)
tr.run(1)
torch.cuda.synchronize()
ws = tr._ws.cpu().numpy()
gs = 224
al = lambda v: (v + 255) // 256 * 256
gidx = ws[:6 * B * 4].view(np.int32).reshape(B, 6)
gval = ws[al(6 * B * 4):al(6 * B * 4) + 6 * B * gs * 4].view(np.float32).reshape(6 * B, gs)
part = ws[al(6 * B * 4) + al(6 * B * gs * 4):al(6 * B * 4) + al(6 * B * gs * 4) + 4 * B * gs * 4].view(np.float32).reshape(4 * B, gs)
print("gidx >=0 per slot column:", (gidx >= 0).sum(0), "of", B)
print("gidx sample:", gidx[:3])
nz = (np.abs(gval).max(1) > 0)
print("G rows nonzero:", nz.sum(), "expected ~", (gidx >= 0).sum())
print("G sentinel rows:", int((gval.view(np.uint32) == 0x55555555).all(1).sum()), "G zero rows:", int((gval == 0).all(1).sum()), "mixed:", int(((gval.view(np.uint32) != 0x55555555).any(1) & (gval != 0).any(1)).sum()))
print("G row 0 head:", gval[0, :4], "im head:", gval[0, 100:104], "pad:", gval[0, 200:204])
print("partials nonzero rows:", (np.abs(part).max(1) > 0).sum())
print("loss mean", float(tr.last_loss.mean()), "table changed rows:", int(((emb - base).abs().amax(1) > 0).sum()))
