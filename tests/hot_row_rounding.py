"""One step in which ONE relation row takes 79,668 gradient terms (39,834 pairs, 700 entities, a single relation): how far the fp32
sums of the C port (terms added one after the other, ScatterSub's order), of the kernels (partial sums of partial sums, hot rows
by float atomics) and of the deterministic kernels end from the fp64 restatement.  Measured on MI355X (round 4): C port 2.8e-5, kernels
1.8e-6, deterministic kernels 5e-8.  Test infrastructure (uses the oracle); `python tests/hot_row_rounding.py` on a GPU box."""
import numpy as np, sys, json
sys.path.insert(0,'.')
import torch
from graphembeddings_amd import data as D, hole as H
from oracle import c_oracle as CO
from oracle import hole_oracle as O
rng=np.random.default_rng(1)
B,d=39834,200
data, tri = D.synthetic_large(n_entities=700, n_relations=1, n_types=2, n_triples=2*B+7, seed=5, zipf_s=0.8)
names, id_to_type, offsets, ids = D.synthetic_large_type_arrays(data)
N=data.entity_count
table=(rng.standard_normal((N,d))*0.08).astype(np.float32); table[::3]*=6.0
tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
out={}
for det in (False, True):
    emb=torch.as_tensor(table).cuda().clone()
    tr=H.Trainer(emb, torch.as_tensor(tri).cuda(), tt, B, margin=5.0, learning_rate=0.1, decay_steps=1e9, seed=3, deterministic=det)
    tr.run(1); torch.cuda.synchronize()
    neg=tr._neg.cpu().numpy()
    out[det]=emb.cpu().numpy(); tr.close()
pos=tri[:B]
assert np.array_equal(neg, CO.corrupt_batch(pos,id_to_type,offsets,ids,3,0,1024,0))
c=table.copy(); CO.hinge_step(c,pos,neg,5.0,0.1,threads=16)
t64,_=O.sgd_step(table.astype(np.float64),pos,neg,0.1,5.0)
res={"row0_abs_err_vs_fp64": {"c_port_fp32": float(np.abs(c[0]-t64[0]).max()), "gpu_atomics": float(np.abs(out[False][0]-t64[0]).max()), "gpu_deterministic": float(np.abs(out[True][0]-t64[0]).max())},
     "all_rows_abs_err_vs_fp64": {"c_port_fp32": float(np.abs(c-t64).max()), "gpu_atomics": float(np.abs(out[False]-t64).max()), "gpu_deterministic": float(np.abs(out[True]-t64).max())},
     "gpu_vs_c_port": float(np.abs(out[False]-c).max()), "row0_moved": float(np.abs(t64[0]-table[0]).max()), "terms_in_row0": int(2*B)}
print(json.dumps(res))
