"""The ComplEx hinge SGD step (holE.py:161-234, 296) written with stock PyTorch-ROCm ops (gather, clip,
complex product, sigmoid, hinge, autograd, index_add_) on the same GPU, against ge_train_steps.
Shows what the hand-written path buys over an eager framework translation; also a parity check of the loss."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphembeddings_amd import hole as H, data as D

fb = D.fb15k_shape()
d, B = 200, 4096
names, id_to_type, offsets, ids = fb.type_arrays()
tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
tri = torch.as_tensor(D.synthetic_fb15k_triples(fb, n_triples=483142, seed=0)).cuda()
emb0 = H.init_embeddings(fb.entity_count, d)


def clip_rows(x):
    return x * torch.clamp(torch.rsqrt((x * x).sum(1, keepdim=True)), max=1.0)


def eager_step(table, pos, neg, lr, margin=0.2):
    idx = torch.cat([pos, neg], 0).long()
    rows = [table[idx[:, c]].detach().requires_grad_(True) for c in range(3)]      # gathered leaves
    k = d // 2
    y = [clip_rows(r) for r in rows]
    h, t, r = [torch.complex(v[:, :k], v[:, k:]) for v in y]
    e = torch.sigmoid((h * r * torch.conj(t)).real.sum(1))
    loss = torch.clamp(e[:B] - e[B:] + margin, min=0.0)
    loss.sum().backward()
    with torch.no_grad():
        for c in range(3):
            table.index_add_(0, idx[:, c], rows[c].grad, alpha=-lr)
    return loss.detach()


res = {}
pos = tri[:B].contiguous()
neg = H.corrupt_batch(tt, fb.relation_count, pos, seed=0, step=0)
t_e = emb0.clone()
for _ in range(3): l_e = eager_step(t_e, pos, neg, 0.1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): l_e = eager_step(t_e, pos, neg, 0.1)
torch.cuda.synchronize(); res["torch_eager_us_per_step"] = round((time.perf_counter() - t0) / 50 * 1e6, 1)

# same first step through the library: the losses must agree
t_a, t_b = emb0.clone(), emb0.clone()
l_ours = H.HingeSGD(t_a, B, margin=0.2).step(pos, neg, 0.1)[:, 0]
l_ref = eager_step(t_b, pos, neg, 0.1)
res["first_step_loss_max_abs_diff"] = float((l_ours - l_ref).abs().max())
res["first_step_table_max_abs_diff"] = float((t_a - t_b).abs().max())

tr = H.Trainer(emb0.clone(), tri, tt, B, seed=0)
tr.run(64); torch.cuda.synchronize(); t0 = time.perf_counter()
tr.run(400); torch.cuda.synchronize()
res["ge_train_steps_us_per_step"] = round((time.perf_counter() - t0) / 400 * 1e6, 1)
res["speedup"] = round(res["torch_eager_us_per_step"] / res["ge_train_steps_us_per_step"], 1)
res["note"] = "eager step excludes the sampler (negatives given); ge_train_steps includes it"
print(json.dumps(res))
