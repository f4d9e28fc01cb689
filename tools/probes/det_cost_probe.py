"""What GE_STEP_DETERMINISTIC (Trainer(deterministic=True): rows with > 16 gradient slots reduced in a fixed order, one more
launch per step) costs: us per step with and without it, alternated, at BASELINE config 2 (FB15k-shaped, B = 4096) and at
config 4's workload on one GPU (1.2 M rows, B = 65,536)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graphembeddings_amd import data as D
from graphembeddings_amd import hole as H


def time_steps(emb, tri, tt, B, det, steps):
    tr = H.Trainer(emb, tri, tt, B, margin=0.2, learning_rate=0.1, decay_steps=1e6, seed=0, deterministic=det)
    tr.run(steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.run(steps)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    tr.close()
    return el * 1e6


def main():
    out = {}
    fb = D.fb15k_shape()
    names, id_to_type, offsets, ids = fb.type_arrays()
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
    tri = torch.as_tensor(D.synthetic_fb15k_triples(fb, n_triples=483142, seed=0)).cuda()
    res = {False: [], True: []}
    for rep in range(3):
        for det in (False, True):
            res[det].append(time_steps(H.init_embeddings(fb.entity_count, 200, seed=0), tri, tt, 4096, det, 400))
    out["config2 fb15k-shaped B=4096"] = {"us_per_step": float(np.median(res[False])), "us_per_step_deterministic": float(np.median(res[True])),
                                           "all": {"off": res[False], "on": res[True]}}
    data, tri2 = D.synthetic_large(n_entities=1_200_000, n_triples=4_000_000, seed=1234)
    names, id_to_type, offsets, ids = D.synthetic_large_type_arrays(data)
    tt2 = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
    dtri = torch.as_tensor(tri2).cuda()
    res = {False: [], True: []}
    for rep in range(3):
        for det in (False, True):
            res[det].append(time_steps(H.init_embeddings(data.entity_count, 200, seed=0), dtri, tt2, 65536, det, 32))
    out["config4 workload on one GPU, B=65536"] = {"us_per_step": float(np.median(res[False])), "us_per_step_deterministic": float(np.median(res[True])),
                                                    "all": {"off": res[False], "on": res[True]}}
    for v in out.values():
        v["cost_pct"] = 100.0 * (v["us_per_step_deterministic"] / v["us_per_step"] - 1.0)
    print(json.dumps(out, indent=1))
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out", "r04_deterministic_hot_rows_cost.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
