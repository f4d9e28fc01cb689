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
from heapq import heappop, heappush

import numpy as np
import torch

from . import hole as H


def eval_link_prediction(scores, id_to_metadata, true_triples, test_triples, max_triples,
                         raw_positions, filtered_positions, infer_threshold=None, output=None):
    """Host version with the reference's signature (holE.py:427-428) for one candidate list:
    `scores` iterates (loss_row, triple) pairs as `zip(batch_loss, triples)` does (holE.py:571).
    Printing is dropped; `output` (file-like, optional) receives the inference_results.tsv lines
    (holE.py:456).  infer_threshold=None disables the is_confident gate (holE.py:438)."""
    heap = []
    min_loss = 100
    for pair in scores:
        loss = float(np.asarray(pair[0]).reshape(-1)[0])
        min_loss = min(min_loss, loss)
        heappush(heap, (loss, tuple(int(v) for v in pair[1])))
    is_confident = True if infer_threshold is None else (min_loss < infer_threshold)
    raw_rank = 0
    filtered_rank = 0
    while heap:
        loss, (head_id, tail_id, relation_id) = heappop(heap)
        raw_rank += 1
        in_sample = tail_id in true_triples[head_id][relation_id]
        if output is not None and is_confident and filtered_rank < max_triples:
            output.write('{:.6f}\t{}\t{}\t{}\t{}\n'.format(loss, head_id, tail_id, relation_id, in_sample))
        if is_confident and in_sample:
            continue
        filtered_rank += 1
        if is_confident and tail_id in test_triples[head_id][relation_id]:
            raw_positions.append(raw_rank)
            filtered_positions.append(filtered_rank)


def score_mrr(raw_positions, filtered_positions, verbose: bool = True) -> dict:
    """holE.py:475-490: raw / filtered MRR, mean positions, Hits@1/3/10 (percent)."""
    raw = np.array(raw_positions, dtype=np.float64)
    fil = np.array(filtered_positions, dtype=np.float64)
    out = {
        "raw_mrr": float(np.mean(1.0 / raw)), "mean_raw_pos": float(np.mean(raw)),
        "filtered_mrr": float(np.mean(1.0 / fil)), "mean_filtered_pos": float(np.mean(fil)),
        "hits1": float(np.mean(fil <= 1).sum() * 100), "hits3": float(np.mean(fil <= 3).sum() * 100),
        "hits10": float(np.mean(fil <= 10).sum() * 100),
    }
    if verbose:
        print('Raw MRR: {} (mean position: {})'.format(out["raw_mrr"], out["mean_raw_pos"]))
        print('Filtered MRR: {} (mean position: {})'.format(out["filtered_mrr"], out["mean_filtered_pos"]))
        print('Hits at 1: {}, 3: {}, 10: {}'.format(out["hits1"], out["hits3"], out["hits10"]))
    return out


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
    return score_mrr(np.concatenate(raw), np.concatenate(fil), verbose=verbose)
