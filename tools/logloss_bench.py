"""--log_loss step rate: the native loop (ge_train_steps_logloss) vs the round-1 route (Python per step: K sampler
launches, ge_complex_logloss_step with two dense table passes and an atomic scatter).  FB15k-shaped table and the
960 MB synthetic table.  Usage (GPU box): python tools/logloss_bench.py"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphembeddings_amd import data as D, hole as H

def case(name, n_rows, tt, tri, B, K, d=200, steps=100, l2=1e-6):
    out = {"case": name, "B": B, "K": K, "table_mb": round(n_rows * d * 4 / 1e6, 1)}
    emb = H.init_embeddings(n_rows, d, seed=1)
    tr = H.Trainer(emb, tri, tt, B, learning_rate=0.05, seed=1).enable_log_loss(K, l2)
    tr.run(steps); torch.cuda.synchronize()
    t0 = time.perf_counter(); tr.run(steps); torch.cuda.synchronize()
    out["native_us_per_step"] = (time.perf_counter() - t0) / steps * 1e6
    tr.close()
    emb = H.init_embeddings(n_rows, d, seed=1)
    opt = H.LogLossSGD(emb, l2)
    def py_steps(n):
        for s in range(n):
            pos = tri[(s * B) % (tri.shape[0] - B):][:B]
            negs = [H.corrupt_batch(tt, 0, pos, seed=1, step=s * K + i) for i in range(K)]
            opt.step(pos, negs, 0.05)
    py_steps(10); torch.cuda.synchronize()
    t0 = time.perf_counter(); py_steps(30); torch.cuda.synchronize()
    out["python_per_step_us"] = (time.perf_counter() - t0) / 30 * 1e6
    out["scored_triples_per_s_native"] = (1 + K) * B / (out["native_us_per_step"] * 1e-6)
    return out

fb = D.fb15k_shape()
names, id_to_type, offsets, ids = fb.type_arrays()
tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
tri = torch.as_tensor(D.synthetic_fb15k_triples(fb, n_triples=483142, seed=0)).cuda()
res = [case("fb15k", fb.entity_count, tt, tri, 512, 1), case("fb15k", fb.entity_count, tt, tri, 4096, 1), case("fb15k", fb.entity_count, tt, tri, 1024, 8)]
data, tl = D.synthetic_large(n_entities=1_200_000, n_triples=2_000_000, seed=1234)
n2, i2, o2, d2 = D.synthetic_large_type_arrays(data)
tt2 = H.TypeTables.from_host(i2, o2, d2, padded_size=1024)
res.append(case("synthetic-1.2M", data.entity_count, tt2, torch.as_tensor(tl).cuda(), 4096, 1))
for r in res:
    print(json.dumps(r))
