"""How far the FB15k material the reference ships (the 50,000-triple validation split as training set) can be trained: filtered
MRR of held-out test triples after n steps of the native loop, GPU only (tests/mrr_parity.py picks its "trained" length from
this).  usage: python tools/probes/mrr_trajectory_probe.py [init_scale] [margin] [lr]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graphembeddings_amd import data as D
from graphembeddings_amd import evaluate as E
from graphembeddings_amd import hole as H


def main():
    init_scale = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
    margin = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
    lr0 = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
    B, d, seed = 4096, 200, 7
    fb = D.fb15k_shape()
    names, id_to_type, offsets, ids = fb.type_arrays()
    inf = D.init_inference_data(D.PACKAGE_FB15K_DIR)
    rng = np.random.default_rng(seed)
    train = fb.validation_triples.astype(np.int32).copy()
    rng.shuffle(train)
    T = len(train)
    seen_e = np.zeros(fb.entity_count, bool); seen_e[train[:, 0]] = True; seen_e[train[:, 1]] = True
    seen_r = np.zeros(fb.entity_count, bool); seen_r[train[:, 2]] = True
    test = inf.test_array[seen_e[inf.test_array[:, 0]] & seen_e[inf.test_array[:, 1]] & seen_r[inf.test_array[:, 2]]]
    known = {tuple(x) for x in train.tolist()}
    test = np.array([x for x in test.tolist() if tuple(x) not in known][:2000], np.int64)
    cand = np.arange(fb.relation_count, fb.entity_count)
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
    emb = H.init_embeddings(fb.entity_count, d, seed=seed)
    emb.mul_(init_scale)
    tr = H.Trainer(emb, torch.as_tensor(train).cuda(), tt, B, margin=margin, learning_rate=lr0, decay_steps=32.0 * (T // B),
                   decay_rate=0.5, seed=seed)
    out, done = [], 0
    for n in (400, 1200, 5000, 20000, 50000, 100000):
        tr.run(n - done)
        done = n
        torch.cuda.synchronize()
        ranks = [E.link_prediction_ranks(emb, test, cand, train, side=s) for s in ("tail", "head")]
        m = E.mrr_and_hits(np.concatenate([np.asarray(r[0]) for r in ranks]), np.concatenate([np.asarray(r[1]) for r in ranks]))
        out.append({"steps": n, "mean_hinge": float(tr.last_loss.mean()), "filtered_mrr": m["filtered_mrr"], "hits10": m["hits10"],
                    "mean_filtered_pos": m["mean_filtered_pos"]})
        print(json.dumps(out[-1]), flush=True)
    tr.close()


if __name__ == "__main__":
    main()
