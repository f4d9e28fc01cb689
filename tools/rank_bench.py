"""Full FB15k-shaped two-sided link-prediction evaluation (59,071 test triples x 14,951 candidates, d=200):
ranks counted in the GEMM epilogue (ge_complex_rank_1vK) vs scores stored by ge_complex_score_1vK and ranked
with tensor ops.  Usage (GPU box): python tools/rank_bench.py"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphembeddings_amd import data as D, evaluate as E, hole as H

inf = D.init_inference_data(D.PACKAGE_FB15K_DIR)
emb = H.init_embeddings(inf.entity_count, 200, seed=3) * 4.0
R, N = inf.relation_count, inf.entity_count
cand = np.arange(R, N, dtype=np.int32)
test, known = inf.test_array, inf.validation_triples
out = {"test_triples": int(len(test)), "candidates": int(len(cand)), "d": 200}
for fused in (True, False):
    E.link_prediction_ranks(emb, test[:4096], cand, known, side="tail", fused=fused)   # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = [E.link_prediction_ranks(emb, test, cand, known, side=s, fused=fused) for s in ("tail", "head")]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    m = E.mrr_and_hits(np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res]))
    out["fused" if fused else "stored_scores"] = {"seconds_two_sided": el, "filtered_mrr": m["filtered_mrr"],
                                                   "sweep_tflops_incl_host": 2 * 2.0 * len(test) * len(cand) * 200 / el / 1e12}
# the sweep kernel alone
hr = torch.as_tensor(np.stack([test[:, 0], test[:, 2]], 1).astype(np.int32)).cuda()
tid = torch.as_tensor(test[:, 1].astype(np.int32)).cuda()
c = torch.as_tensor(cand).cuda()
planes = H.RankPlanes(emb, c)        # built once per evaluation (evaluate_fb15k_style does the same)
H.rank_candidates(emb, hr, tid, c, planes=planes)
ev = H.Events(2)
ev.record(0)
for _ in range(3):
    H.rank_candidates(emb, hr, tid, c, planes=planes)
ev.record(1)
torch.cuda.synchronize()
ms = ev.elapsed_ms(0, 1) / 3
out["rank_kernel_ms"] = ms
out["rank_kernel_tflops"] = 2.0 * len(test) * len(cand) * 200 / (ms * 1e-3) / 1e12
# The same sweep when the model is good: every true tail is the best-scoring candidate of its (head, relation), so
# few candidates score within rounding of it and the exact-comparison path of ge_rank_pipe.hip is rarely taken.
# (With random true tails, above, near-ties occur in most 32 x 64 blocks: the worst case for that path.)
best = []
for s0 in range(0, len(test), 8192):
    sc = H.score_candidates(emb, hr[s0:s0 + 8192], c)
    best.append(c[sc.argmin(1)])                      # loss = sigmoid(score): holE.py ranks ascending
    del sc
tid_top = torch.cat(best).int()
nb, _ = H.rank_candidates(emb, hr, tid_top, c, planes=planes)
assert int(nb.max()) == 0                             # nothing pops before the best candidate
ev.record(0)
for _ in range(3):
    H.rank_candidates(emb, hr, tid_top, c, planes=planes)
ev.record(1)
torch.cuda.synchronize()
ms = ev.elapsed_ms(0, 1) / 3
out["rank_kernel_true_at_top_ms"] = ms
out["rank_kernel_true_at_top_tflops"] = 2.0 * len(test) * len(cand) * 200 / (ms * 1e-3) / 1e12
print(json.dumps(out))
