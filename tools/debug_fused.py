#!/usr/bin/env python3
"""Debug: one fused step vs one two-kernel step from the same state; report which rows differ."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphembeddings_amd import hole as H, data as D, _lib

d = 200
fb = D.fb15k_shape()
names, id_to_type, offsets, ids = fb.type_arrays()
tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
for B in [int(x) for x in sys.argv[1:]] or [8, 64, 1024]:
    tri = D.synthetic_fb15k_triples(fb, n_triples=5 * B + 77, seed=11)
    base = H.init_embeddings(fb.entity_count, d, seed=1)
    base[::4] *= 7.0
    outs = {}
    for fused in (0, 1):
        _lib.load().ge_set_fused_step(fused)
        emb = base.clone()
        tr = H.Trainer(emb, torch.as_tensor(tri).cuda(), tt, B, seed=21)
        tr.run(1)
        torch.cuda.synchronize()
        outs[fused] = emb.cpu().numpy()
        neg = tr._neg.cpu().numpy()
    diff = np.abs(outs[0] - outs[1]).max(1)
    upd = np.abs(outs[0] - base.cpu().numpy()).max(1)
    pos = tri[:B]
    rows = np.concatenate([pos.ravel(), neg.ravel()])
    occ = np.bincount(rows, minlength=len(diff))
    bad = np.nonzero(diff > 1e-6 * np.maximum(upd, 1e-6) + 1e-7)[0]
    print(f"B={B}: rows updated={int((upd>0).sum())} differing={len(bad)} max_diff={diff.max():.3e} max_update={upd.max():.3e}")
    for r in bad[:12]:
        print(f"   row {r}: diff {diff[r]:.3e} update {upd[r]:.3e} occurrences(pos+neg cols) {occ[r]} fused_update {np.abs(outs[1][r]-base.cpu().numpy()[r]).max():.3e}")
