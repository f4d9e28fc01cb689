"""Where a row-sharded step spends its time at world_size 1 (one MI355X): plan_chunk vs the per-step
phases.  Usage (GPU box): python tools/sharded_breakdown.py [B] [S]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphembeddings_amd import data as D, hole as H, sharded as S

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
SS = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda", 0)
n_rel, n_ent, d = 18, 1_200_000, 200
N = n_rel + n_ent
data, _ = D.synthetic_large(n_entities=n_ent, n_triples=1, seed=1234)
names, id_to_type, offsets, ids = D.synthetic_large_type_arrays(data)
tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024, device=dev)
rng = np.random.default_rng(1)
n_loc = B * 64
tri = np.stack([n_rel + D._zipf_sample(rng, n_ent, n_loc, 0.8), n_rel + D._zipf_sample(rng, n_ent, n_loc, 0.8),
                rng.integers(0, n_rel, n_loc)], 1).astype(np.int32)
dtri = torch.as_tensor(tri).to(dev)
shard = torch.randn(N, d, device=dev) * 0.04
tr = S.ShardedTrainer(shard, N, tt, seed=0)


def sync_time(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6, out


pos = torch.stack([dtri[j * B:(j + 1) * B] for j in range(SS)], 0)
t_neg, neg = sync_time(lambda: tr.sample_negatives(pos))
t_plan, plan = sync_time(lambda: tr.plan_chunk(pos, neg.to(torch.int32)))
print(f"B={B} S={SS}: sample_negatives {t_neg/SS:.1f} us/step, plan_chunk {t_plan/SS:.1f} us/step ({t_plan:.0f} us per chunk)")
t_steps, _ = sync_time(lambda: [tr.step_planned(plan, s, 0.05) for s in range(SS)])
print(f"step_planned {t_steps/SS:.1f} us/step")

# phases of one step, each synchronised
s = 0
sc, rc = plan.sc[s], plan.rc[s]
req = plan.req_all[plan.req_start[s]:plan.req_start[s + 1]]
k = tr.k
ph = {}
ph["gather"], rows_out = sync_time(lambda: k.gather_rows(tr.shard, req), 20)
ph["a2a rows (copy at world 1)"], staged = sync_time(lambda: tr._a2a(rows_out, rc, sc), 20)
remap = plan.remap[s]
ph["remap slices"], (p_, n_) = sync_time(lambda: (remap[:B].contiguous(), remap[B:].contiguous()), 20)
ph["hinge_grad"], (loss, gi, gv) = sync_time(lambda: k.hinge_grad(staged, p_, n_, 0.05, 0.2, "complex", 1.0), 20)
ri = plan.reduce_items
i0, i1 = ri.item_start[s], ri.item_start[s + 1]
gsum = torch.empty_like(staged)
ph["segment_sum reduce"], _ = sync_time(lambda: k.segment_sum_rows(gv, gi, ri.order, ri.begin[i0:i1], ri.length[i0:i1], ri.target[i0:i1], gsum, False), 20)
g2 = torch.zeros_like(staged)
ph["(old) zeros + atomic scatter reduce"], _ = sync_time(lambda: (g2.zero_(), k.scatter_add_rows(g2, gi, gv)), 20)
ph["a2a grads (copy at world 1)"], recv_g = sync_time(lambda: tr._a2a(gsum, sc, rc), 20)
ai = plan.apply_items
j0, j1 = ai.item_start[s], ai.item_start[s + 1]
ph["segment_sum apply"], _ = sync_time(lambda: k.segment_sum_rows(recv_g, None, ai.order, ai.begin[j0:j1], ai.length[j0:j1], ai.target[j0:j1], tr.shard, True), 20)
ph["(old) atomic scatter apply"], _ = sync_time(lambda: k.scatter_add_rows(tr.shard, req, recv_g), 20)
print(f"unique rows {staged.shape[0]}, reduce items {i1 - i0} (split rows {ri.split_start[s+1]-ri.split_start[s]}), apply items {j1 - j0}")
for n_, v in ph.items():
    print(f"  {n_:40s} {v:8.1f} us")
