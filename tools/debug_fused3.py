#!/usr/bin/env python3
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphembeddings_amd import hole as H, data as D, _lib
d, B = 200, int(sys.argv[1]) if len(sys.argv) > 1 else 2
fb = D.fb15k_shape()
names, id_to_type, offsets, ids = fb.type_arrays()
tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
tri = D.synthetic_fb15k_triples(fb, n_triples=5 * B + 77, seed=11)
base = H.init_embeddings(fb.entity_count, d, seed=1)
_lib.load().ge_set_fused_step(1)
emb = base.clone()
tr = H.Trainer(emb, torch.as_tensor(tri).cuda(), tt, B, seed=21)
tr._ws.fill_(0x55)
torch.cuda.synchronize()
tr.run(1)
torch.cuda.synchronize()
ws = tr._ws.cpu().numpy()
ch = np.nonzero(ws != 0x55)[0]
print("workspace bytes", len(ws), "changed bytes", len(ch))
# print changed ranges
if len(ch):
    starts = [ch[0]]; prev = ch[0]
    ranges = []
    for x in ch[1:]:
        if x != prev + 1:
            ranges.append((starts[-1], prev)); starts.append(x)
        prev = x
    ranges.append((starts[-1], prev))
    print("ranges:", ranges[:60], "n_ranges", len(ranges))
al = lambda v: (v + 255) // 256 * 256
gs = 224
print("expected regions: gidx [0,%d) gval [%d,%d) partials [%d,%d) prep from %d" % (6*B*4, al(6*B*4), al(6*B*4)+6*B*gs*4, al(6*B*4)+al(6*B*gs*4), al(6*B*4)+al(6*B*gs*4)+4*B*gs*4, al(6*B*4)+al(6*B*gs*4)+al(4*B*gs*4)))
print("table changed rows:", int(((emb - base).abs().amax(1) > 0).sum()), "loss", tr.last_loss.cpu().numpy())
