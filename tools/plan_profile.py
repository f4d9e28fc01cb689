import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from graphembeddings_amd import data as D, hole as H, sharded as S
B, SS = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
n_rel, n_ent, d = 18, 1_200_000, 200
N = n_rel + n_ent
data, _ = D.synthetic_large(n_entities=n_ent, n_triples=1, seed=1234)
names, id_to_type, offsets, ids = D.synthetic_large_type_arrays(data)
tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024, device=dev)
rng = np.random.default_rng(1)
n_loc = B * SS
tri = np.stack([n_rel + D._zipf_sample(rng, n_ent, n_loc, 0.8), n_rel + D._zipf_sample(rng, n_ent, n_loc, 0.8), rng.integers(0, n_rel, n_loc)], 1).astype(np.int32)
dtri = torch.as_tensor(tri).to(dev)
shard = torch.zeros(N, d, device=dev)
tr = S.ShardedTrainer(shard, N, tt, seed=0)
pos = dtri.view(SS, B, 3)
neg = tr.sample_negatives(pos).to(torch.int32)
for _ in range(2): plan = tr.plan_chunk(pos, neg)
torch.cuda.synchronize()
t0 = time.perf_counter(); plan = tr.plan_chunk(pos, neg); torch.cuda.synchronize(); print("plan_chunk ms", (time.perf_counter() - t0) * 1e3, "per step us", (time.perf_counter() - t0) * 1e6 / SS)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    plan = tr.plan_chunk(pos, neg); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
