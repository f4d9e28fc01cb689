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


class KnownIndex:
    """Known-true triples as a device-side sorted index: key = fixed entity * n_rows + relation -> the entities
    that complete a known triple (tails for side="tail", heads for "head")."""

    def __init__(self, known_triples, n_rows: int, side: str, device):
        self.n_rows = int(n_rows)
        if known_triples is None or len(known_triples) == 0:
            self.key = torch.empty(0, dtype=torch.int64, device=device)
            self.ent = torch.empty(0, dtype=torch.int64, device=device)
            return
        k = torch.as_tensor(np.asarray(known_triples, dtype=np.int64)).to(device)
        fixed, other = (k[:, 0], k[:, 1]) if side == "tail" else (k[:, 1], k[:, 0])
        # a set, like the reference's dict of sets (holE.py:413-422), sorted by (fixed, relation, other)
        if self.n_rows ** 3 < 2 ** 63:      # one radix sort of the triple packed into an int64
            packed = torch.unique_consecutive(torch.sort((fixed * self.n_rows + k[:, 2]) * self.n_rows + other)[0])
            self.key, self.ent = (packed // self.n_rows).contiguous(), (packed % self.n_rows).contiguous()
        else:
            pairs = torch.unique(torch.stack([fixed * self.n_rows + k[:, 2], other], 1), dim=0)
            self.key, self.ent = pairs[:, 0].contiguous(), pairs[:, 1].contiguous()

    def cells(self, fixed: torch.Tensor, rel: torch.Tensor, pos_of: torch.Tensor, n_cand: int):
        """(row, candidate position) of every known-true candidate of the query rows, as the per-tile lists
        ge_complex_rank_1vK takes: (known_off int32 [tiles+1], known_rc int16) -- ge_known_cells: a counting pass, the total
        read back (the call's one synchronisation), a filling pass."""
        B = fixed.numel()
        dev = fixed.device
        if dev.type != "cuda" or pos_of.dtype != torch.int64 or not pos_of.is_cuda:
            raise ValueError("KnownIndex.cells runs on the GPU (ge_known_cells): CUDA tensors, pos_of int64")
        n_tiles = ((B + 127) // 128) * ((n_cand + 127) // 128)
        off = torch.empty(n_tiles + 1, dtype=torch.int32, device=dev)
        scratch = torch.empty(max(n_tiles, 1), dtype=torch.int32, device=dev)
        fixed, rel = fixed.to(torch.int64).contiguous(), rel.to(torch.int64).contiguous()
        args = (self.key.data_ptr(), self.ent.data_ptr(), self.key.numel(), fixed.data_ptr(), rel.data_ptr(), B,
                pos_of.data_ptr(), self.n_rows, n_cand, scratch.data_ptr(), off.data_ptr())
        H._lib.call("ge_known_cells", 0, *args, None, H._stream())
        total = int(off[-1])
        rc = torch.empty(max(total, 1), dtype=torch.int16, device=dev)
        if total:
            H._lib.call("ge_known_cells", 1, *args, rc.data_ptr(), H._stream())
        else:
            rc.zero_()
        return off, rc


@torch.no_grad()
def link_prediction_ranks(embeddings: torch.Tensor, test_triples: np.ndarray, candidates: np.ndarray,
                          known_triples: np.ndarray = None, side: str = "tail", batch: int = None,
                          max_norm: float = 1.0, fused: bool = None, model: str = "complex", planes=None,
                          infer_threshold: float = None, return_confident: bool = False):
    """Raw and filtered rank of every test triple's true entity among `candidates`, with the
    semantics of holE.py:446-469.  side="tail": candidates replace the tail; "head": the head.
    known_triples: an [n,3] array, or a KnownIndex built for this side (evaluate_fb15k_style keeps the two it builds).
    Returns (raw_ranks, filtered_ranks) int64 arrays.  The true entity must be a candidate.
    fused (default: whenever the kernel supports embedding_dim): the ranks are counted in the candidate
    GEMM's epilogue (ge_complex_rank_1vK) and no [B,K] score matrix is materialised; otherwise the scores
    of ge_complex_score_1vK are ranked with tensor ops.
    model: "complex"; "hole" (README.md:42 on a real-valued table: a copy is taken to the frequency domain once,
    where HolE is the ComplEx-shaped form the sweep computes) or "hole_spectral" (table already there).  HolE
    needs the fused sweep.
    planes: H.RankPlanes of (embeddings, candidates) shared between calls (tails then heads: evaluate_fb15k_style); built
    here otherwise -- once for all the batches of the call.
    infer_threshold: the reference's gate (holE.py:436-438, flag holE.py:616): a sweep is `is_confident` when the lowest
    loss among its candidates is below the threshold, and ONLY confident sweeps record their positions (holE.py:464-466)
    -- the returned arrays then hold the confident rows' ranks only (return_confident=True adds the bool mask over all
    test rows).  None: every row is recorded (the reference with a threshold above every loss)."""
    assert side in ("tail", "head")
    if model not in ("complex", "hole", "hole_spectral"):
        raise ValueError(f"unknown model {model!r}")
    dev = embeddings.device
    d = embeddings.shape[1]
    can_fuse = d % 8 == 0 and d <= H.rank_max_dim() and (d <= 232 or max_norm <= 8.0)   # (fp32 kernels: up to 232)
    if fused is None:
        fused = can_fuse
    if model != "complex":
        if not (fused and can_fuse):
            raise ValueError("HolE link prediction needs the fused sweep: embedding_dim a multiple of 8, <= %d" % H.rank_max_dim())
        if model == "hole":
            embeddings = H.hole_to_spectral(embeddings.detach().clone())
        model = "hole_spectral"
    test = np.asarray(test_triples, dtype=np.int64)
    cand = torch.as_tensor(np.asarray(candidates, dtype=np.int32)).to(dev)
    cand64 = cand.to(torch.int64)
    # position of every row id in the candidate list (-1: not a candidate)
    pos_of = torch.full((embeddings.shape[0],), -1, dtype=torch.int64, device=dev)
    pos_of[cand64] = torch.arange(cand.numel(), device=dev)
    index = known_triples if isinstance(known_triples, KnownIndex) else KnownIndex(known_triples, embeddings.shape[0], side, dev)
    if batch is None:
        # test rows per call: the stored-scores path holds a [batch, K] fp32 matrix; the fused sweep holds nothing
        # per row, and longer calls amortise its per-row-block set-up (the whole FB15k test set is one call)
        batch = 1 << 17 if fused else 16384
    raw_all, fil_all, conf_all = [], [], []
    fixed_col, true_col = (0, 1) if side == "tail" else (1, 0)
    if fused and planes is None:
        planes = H.RankPlanes(embeddings, cand, max_norm=max_norm, model=model)
    if planes is not None:
        # the planes' own id tensor is handed to the kernel: it has to BE this candidate list (pos_of and the known cells
        # below are built from the caller's), not merely as long
        if planes.cand.numel() != cand.numel() or not bool(torch.equal(planes.cand.to(dev), cand)):
            raise ValueError("`planes` were built for another candidate list")
        cand = planes.cand
    for s in range(0, len(test), batch):
        chunk = torch.as_tensor(test[s:s + batch]).to(dev)
        fixed, rel, true_id = chunk[:, fixed_col], chunk[:, 2], chunk[:, true_col]
        hr = torch.stack([fixed, rel], 1).to(torch.int32)
        tpos = pos_of[true_id]
        if (tpos < 0).any():
            raise ValueError("a test triple's true entity is not in the candidate list")
        off, rc = index.cells(fixed, rel, pos_of, cand.numel())
        if fused:
            n_before, n_known = H.rank_candidates(embeddings, hr, true_id, cand, known_off=off, known_rc=rc,
                                                  cand_is_head=(side == "head"), max_norm=max_norm, model=model, planes=planes)
            raw = n_before.to(torch.int64) + 1
            fil = raw - n_known.to(torch.int64)
        else:
            scores = H.score_candidates(embeddings, hr, cand, cand_is_head=(side == "head"), max_norm=max_norm)
            s_true = scores.gather(1, tpos.view(-1, 1))
            # ascending by (loss, triple tuple): among equal losses the smaller entity id pops first
            before = (scores < s_true) | ((scores == s_true) & (cand64.view(1, -1) < true_id.view(-1, 1)))
            raw = before.sum(1) + 1
            # filtered: known-true candidates popped before the target do not advance the rank
            n_ct = (cand.numel() + 127) // 128
            tiles = torch.repeat_interleave(torch.arange(off.numel() - 1, device=dev), (off[1:] - off[:-1]).to(torch.int64))
            rcv = rc[:tiles.numel()].to(torch.int64)
            rows_t = (tiles // n_ct) * 128 + rcv // 128
            cols_t = (tiles % n_ct) * 128 + rcv % 128
            skipped = torch.zeros(chunk.shape[0], dtype=torch.int64, device=dev)
            if rows_t.numel():
                skipped.index_add_(0, rows_t, before[rows_t, cols_t].to(torch.int64))
            fil = raw - skipped
        if infer_threshold is not None:
            # is_confident: the sweep's lowest loss is below the threshold <=> some candidate ranks before a loss equal to it
            if fused:
                conf = H.confident_rows(embeddings, hr, cand, infer_threshold, cand_is_head=(side == "head"), max_norm=max_norm,
                                        model=model, planes=planes)
            else:
                conf = scores.min(dim=1).values < infer_threshold
            raw, fil = raw[conf], fil[conf]
            conf_all.append(conf.cpu().numpy())
        raw_all.append(raw.cpu().numpy())
        fil_all.append(fil.cpu().numpy())
    out = (np.concatenate(raw_all), np.concatenate(fil_all))
    if return_confident:
        out += (np.concatenate(conf_all) if conf_all else np.ones(len(test), dtype=bool),)
    return out


def evaluate_fb15k_style(embeddings: torch.Tensor, data, both_sides: bool = True, batch: int = None,
                         verbose: bool = True, model: str = "complex", infer_threshold: float = None) -> dict:
    """Filtered link prediction over all entities (rows >= relation_count) for
    data.test_array, filtering train+valid triples as the reference does (holE.py:413-422).
    infer_threshold: the reference's is_confident gate (holE.py:436-438); the summary then covers the confident sweeps
    only and carries their number (`recorded` of `sweeps`)."""
    R, N = data.relation_count, data.entity_count
    cand = np.arange(R, N, dtype=np.int32)
    parts = [a for a in (data.triples, data.validation_triples) if a is not None]
    known = np.concatenate(parts, 0) if parts else None
    if model == "hole":                      # one transform for both sides
        embeddings, model = H.hole_to_spectral(embeddings.detach().clone()), "hole_spectral"
    d = embeddings.shape[1]
    planes = None                            # the candidates' fp16 planes: one build for tails and heads
    if d % 8 == 0 and d <= H.rank_max_dim():
        planes = H.RankPlanes(embeddings, torch.as_tensor(cand).to(embeddings.device), model=model)
    # the known-triple indexes (a sort each) are kept on `data`: the train / valid splits do not change between the
    # evaluations of a training run
    cache = data.__dict__.setdefault("_known_index_cache", {}) if hasattr(data, "__dict__") else {}
    def known_index(side):
        # keyed on the IDENTITY of the split arrays (kept alive by `data`) as well as their length: a data object whose
        # train / valid triples are replaced by others of the same count must not reuse the old filter
        key = (side, str(embeddings.device), N, 0 if known is None else len(known), tuple(id(a) for a in parts))
        if key not in cache:
            cache[key] = KnownIndex(known, N, side, embeddings.device)
        return cache[key]
    raw_t, fil_t = link_prediction_ranks(embeddings, data.test_array, cand, known_index("tail"), "tail", batch, model=model,
                                         planes=planes, infer_threshold=infer_threshold)
    raw, fil = [raw_t], [fil_t]
    if both_sides:
        raw_h, fil_h = link_prediction_ranks(embeddings, data.test_array, cand, known_index("head"), "head", batch, model=model,
                                             planes=planes, infer_threshold=infer_threshold)
        raw.append(raw_h); fil.append(fil_h)
    raw, fil = np.concatenate(raw), np.concatenate(fil)
    sweeps = len(data.test_array) * (2 if both_sides else 1)
    if raw.size == 0:      # no sweep was confident: the reference would take the mean of nothing (holE.py:477)
        out = {k: float("nan") for k in ("raw_mrr", "mean_raw_pos", "filtered_mrr", "mean_filtered_pos", "hits1", "hits3", "hits10")}
    else:
        out = mrr_and_hits(raw, fil)
    out["recorded"], out["sweeps"] = int(raw.size), int(sweeps)
    if verbose and infer_threshold is not None:
        print(f"is_confident (lowest loss < {infer_threshold}): {raw.size} of {sweeps} sweeps recorded")
    if verbose:
        print("raw MRR {raw_mrr:.6f} (mean rank {mean_raw_pos:.1f}); filtered MRR {filtered_mrr:.6f} "
              "(mean rank {mean_filtered_pos:.1f}); hits@1/3/10 {hits1:.2f} / {hits3:.2f} / {hits10:.2f} %".format(**out))
    return out
