"""Does building chunk c+1's exchange plan on a side stream hide it behind chunk c's steps?  One MI355X,
world_size 1, config-4 workload.  Usage (GPU box): python tools/probes/sharded_pipeline_probe.py [B] [S] [chunks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphembeddings_amd import data as D, hole as H, sharded as S

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
SS = int(sys.argv[2]) if len(sys.argv) > 2 else 16
NC = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda", 0)
n_rel, n_ent, d = 18, 1_200_000, 200
N = n_rel + n_ent
data, _ = D.synthetic_large(n_entities=n_ent, n_triples=1, seed=1234)
names, id_to_type, offsets, ids = D.synthetic_large_type_arrays(data)
tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024, device=dev)
rng = np.random.default_rng(1)
n_loc = B * SS * 2
tri = np.stack([n_rel + D._zipf_sample(rng, n_ent, n_loc, 0.8), n_rel + D._zipf_sample(rng, n_ent, n_loc, 0.8),
                rng.integers(0, n_rel, n_loc)], 1).astype(np.int32)
dtri = torch.as_tensor(tri).to(dev)
shard = torch.randn(N, d, device=dev) * 0.04
tr = S.ShardedTrainer(shard, N, tt, seed=0)
chunks = [torch.stack([dtri[((c * SS + j) * B) % (n_loc - B):][:B] for j in range(SS)], 0) for c in range(NC)]
lr = lambda gs: 0.05

def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (NC * SS) * 1e6

print(f"B={B} S={SS} chunks={NC}")
print(f"sequential (plan, then steps, per chunk): {timed(lambda: [tr.run(c, lr) for c in chunks]):8.1f} us/step")
print(f"pipelined  (plan c+1 on the side stream): {timed(lambda: tr.run_pipelined(chunks, lr)):8.1f} us/step")
plans = [tr._plan_ahead(c, 0) for c in chunks]
torch.cuda.synchronize()
def steps_only():
    for p in plans:
        tr._adopt(p)
        for s in range(p.S):
            tr.step_planned(p, s, 0.05)
print(f"steps only (plans prebuilt):              {timed(steps_only):8.1f} us/step")
t0 = time.perf_counter(); steps_only(); host = time.perf_counter() - t0; torch.cuda.synchronize()
print(f"host time to ENQUEUE the steps:           {host / (NC * SS) * 1e6:8.1f} us/step")
def plans_only():
    for c in chunks:
        tr._plan_ahead(c, 0)
print(f"plans only (side stream, device idle):    {timed(plans_only):8.1f} us/step")
