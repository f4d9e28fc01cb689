#!/usr/bin/env python3
"""Debug: run a few two-kernel (non-fused) steps and validate the prepared index on the host."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphembeddings_amd import hole as H, data as D, _lib

B, d = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 200
fused = int(sys.argv[2]) if len(sys.argv) > 2 else 0
_lib.load().ge_set_fused_step(fused)
fb = D.fb15k_shape()
names, id_to_type, offsets, ids = fb.type_arrays()
tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
tri = D.synthetic_fb15k_triples(fb, n_triples=5 * B + 77, seed=11)
emb = H.init_embeddings(fb.entity_count, d, seed=1)
tr = H.Trainer(emb, torch.as_tensor(tri).cuda(), tt, B, seed=21)
nsteps = 3
tr.run(nsteps)
torch.cuda.synchronize()
ws = tr._ws.cpu().numpy()
gs = (d + 31) // 32 * 32
al = lambda v: (v + 255) // 256 * 256
grad_bytes = al(6 * B * 4) + al(6 * B * gs * 4) + al(4 * B * gs * 4)
stride = 41 * B + 64
prep = ws[grad_bytes:grad_bytes + 2 * 32 * stride * 4].view(np.int32)
ok = True
for s in range(nsteps):
    p = prep[s * stride:(s + 1) * stride]
    neg = p[:3 * B].reshape(B, 3); occ = p[3 * B:7 * B]; items = p[7 * B:27 * B].reshape(4 * B, 5)
    n_items = p[27 * B]; slot_item = p[27 * B + 64:33 * B + 64]; item_cnt = p[33 * B + 64:37 * B + 64]; row_cnt = p[37 * B + 64:41 * B + 64]
    row = (s * B) % (5 * B + 77 - B + 1)
    print(f"step {s}: n_items={n_items} valid_slots={(occ>=0).sum()} slot_item_set={(slot_item>=0).sum()} item_cnt_sum={item_cnt.sum()} row_cnt_sum={row_cnt.sum()}")
    it = items[:n_items]
    cnt = it[:, 2] & 0x3FFFFFFF; multi = (it[:, 2] >> 30) & 1
    assert cnt.sum() == (occ >= 0).sum(), "cnt sum"
    assert (cnt >= 1).all() and (cnt <= 16).all()
    # occ sorted by row, each item's slots map back via slot_item
    for k in range(n_items):
        sl = occ[it[k, 1]:it[k, 1] + cnt[k]]
        assert (slot_item[sl] == k).all(), ("slot_item", k)
    assert ((slot_item >= 0) | (slot_item == -2)).sum() == (occ >= 0).sum()   # -2: sole-contributor slots applied by the grad kernel
    # rows: consecutive items with same row share row_first; n_row_items at first item
    rf = it[:, 3]
    for k in range(n_items):
        f = rf[k]
        assert it[f, 0] == it[k, 0] and f <= k
        nri = it[f, 4]
        assert f + nri > k, ("n_row_items", k, f, nri)
        if multi[k] == 0:
            assert f == k and nri == 1, ("single", k, f, nri)
    # rows distinct across different row_first
    firsts = np.unique(rf)
    assert len(np.unique(it[firsts, 0])) == len(firsts)
    assert (it[firsts, 4].sum() == n_items), ("sum of n_row_items", it[firsts, 4].sum(), n_items)
    if fused:
        assert (item_cnt[:n_items] == cnt).all(), ("arrivals", np.nonzero(item_cnt[:n_items] != cnt)[0][:10])
        mf = np.unique(rf[multi == 1])
        assert (row_cnt[mf] == it[mf, 4]).all(), "row arrivals"
print("prepare outputs consistent; finite:", bool(torch.isfinite(emb).all()))
