"""Link-prediction ranking and MRR / Hits@n (holE.py:427-490), with the candidate sweep on the GPU.

Orientation: E = sigmoid(score) is a loss, candidates are ranked ASCENDING by it (min-heap pop order,
holE.py:446-447); ties are broken by the (head, tail, relation) tuple (holE.py:434).
raw_rank counts every popped candidate; filtered_rank skips candidates that are known-true
train/valid triples (holE.py:454-463); ranks are recorded for tails in the test set
(holE.py:464-466).  The reference evaluator is wired to Diffbot-specific candidates
(holE.py:534-541); here the same ranking semantics run 1-vs-all over a candidate list through
ge_complex_score_1vK (fp32-MFMA GEMM), for tails and -- as FB15k protocols need -- heads.
"""
from __future__ import annotations

from collections import defaultdict

import numpy as np
import torch

from . import hole as H


def mrr_and_hits(raw_ranks, filtered_ranks) -> dict:
    """The summary numbers of holE.py:475-490 from rank arrays: reciprocal-rank means, mean ranks and the
    share of filtered ranks within 1 / 3 / 10 (percent)."""
    raw = np.asarray(raw_ranks, dtype=np.float64)
    fil = np.asarray(filtered_ranks, dtype=np.float64)
    pct = lambda n: 100.0 * float(np.count_nonzero(fil <= n)) / max(1, fil.size)
    return {"raw_mrr": float((1.0 / raw).mean()), "mean_raw_pos": float(raw.mean()),
            "filtered_mrr": float((1.0 / fil).mean()), "mean_filtered_pos": float(fil.mean()),
            "hits1": pct(1), "hits3": pct(3), "hits10": pct(10)}


def _known_lists(known_triples: np.ndarray, side: str):
    """{(fixed entity, relation): sorted candidate ids known to be true} from [T,3] (h,t,r)."""
    d = defaultdict(set)
    if known_triples is None:
        return d
    for h, t, r in np.asarray(known_triples):
        if side == "tail":
            d[(int(h), int(r))].add(int(t))
        else:
            d[(int(t), int(r))].add(int(h))
    return {k: sorted(v) for k, v in d.items()}


@torch.no_grad()
def link_prediction_ranks(embeddings: torch.Tensor, test_triples: np.ndarray, candidates: np.ndarray,
                          known_triples: np.ndarray = None, side: str = "tail", batch: int = 2048,
                          max_norm: float = 1.0):
    """Raw and filtered rank of every test triple's true entity among `candidates`, with the
    semantics of holE.py:446-469.  side="tail": candidates replace the tail; "head": the head.
    Returns (raw_ranks, filtered_ranks) int64 arrays.  The true entity must be a candidate."""
    assert side in ("tail", "head")
    test = np.asarray(test_triples, dtype=np.int64)
    cand = torch.as_tensor(np.asarray(candidates, dtype=np.int32)).to(embeddings.device)
    cand64 = cand.to(torch.int64)
    # position of every row id in the candidate list (-1: not a candidate)
    pos_of = torch.full((embeddings.shape[0],), -1, dtype=torch.int64, device=embeddings.device)
    pos_of[cand64] = torch.arange(cand.numel(), device=embeddings.device)
    known = _known_lists(known_triples, side)
    raw_all, fil_all = [], []
    fixed_col, true_col = (0, 1) if side == "tail" else (1, 0)
    for s in range(0, len(test), batch):
        chunk = test[s:s + batch]
        hr = torch.as_tensor(np.stack([chunk[:, fixed_col], chunk[:, 2]], 1).astype(np.int32)).to(embeddings.device)
        true_id = torch.as_tensor(chunk[:, true_col]).to(embeddings.device)
        scores = H.score_candidates(embeddings, hr, cand, cand_is_head=(side == "head"), max_norm=max_norm)
        tpos = pos_of[true_id]
        if (tpos < 0).any():
            raise ValueError("a test triple's true entity is not in the candidate list")
        s_true = scores.gather(1, tpos.view(-1, 1))
        # ascending by (loss, triple tuple): among equal losses the smaller entity id pops first
        before = (scores < s_true) | ((scores == s_true) & (cand64.view(1, -1) < true_id.view(-1, 1)))
        raw = before.sum(1) + 1
        # filtered: known-true candidates popped before the target do not advance the rank
        rows, cols = [], []
        for i, (f, r) in enumerate(zip(chunk[:, fixed_col], chunk[:, 2])):
            lst = known.get((int(f), int(r)))
            if lst:
                rows.extend([i] * len(lst))
                cols.extend(lst)
        if rows:
            rows_t = torch.as_tensor(rows, device=embeddings.device)
            cpos = pos_of[torch.as_tensor(cols, device=embeddings.device)]
            ok = cpos >= 0
            rows_t, cpos = rows_t[ok], cpos[ok]
            hit = before[rows_t, cpos] & (cpos != tpos[rows_t])
            skipped = torch.zeros(len(chunk), dtype=torch.int64, device=embeddings.device)
            skipped.index_add_(0, rows_t, hit.to(torch.int64))
            fil = raw - skipped
        else:
            fil = raw.clone()
        raw_all.append(raw.cpu().numpy())
        fil_all.append(fil.cpu().numpy())
    return np.concatenate(raw_all), np.concatenate(fil_all)


def evaluate_fb15k_style(embeddings: torch.Tensor, data, both_sides: bool = True, batch: int = 2048,
                         verbose: bool = True) -> dict:
    """Filtered link prediction over all entities (rows >= relation_count) for
    data.test_array, filtering train+valid triples as the reference does (holE.py:413-422)."""
    R, N = data.relation_count, data.entity_count
    cand = np.arange(R, N, dtype=np.int32)
    parts = [a for a in (data.triples, data.validation_triples) if a is not None]
    known = np.concatenate(parts, 0) if parts else None
    raw_t, fil_t = link_prediction_ranks(embeddings, data.test_array, cand, known, "tail", batch)
    raw, fil = [raw_t], [fil_t]
    if both_sides:
        raw_h, fil_h = link_prediction_ranks(embeddings, data.test_array, cand, known, "head", batch)
        raw.append(raw_h); fil.append(fil_h)
    out = mrr_and_hits(np.concatenate(raw), np.concatenate(fil))
    if verbose:
        print("raw MRR {raw_mrr:.6f} (mean rank {mean_raw_pos:.1f}); filtered MRR {filtered_mrr:.6f} "
              "(mean rank {mean_filtered_pos:.1f}); hits@1/3/10 {hits1:.2f} / {hits3:.2f} / {hits10:.2f} %".format(**out))
    return out
