"""Host-side mirror of the holE.py operator interface for the hot path, on libge_hip.so.

Same names, argument meaning and orientation as the reference functions (cited per function);
tensors are PyTorch-ROCm CUDA tensors instead of TF graph nodes, and every call enqueues hand-written
HIP kernels on torch's current stream through the C ABI of include/ge_hip.h.  PyTorch is plumbing
here (device memory, streams); there is no torch / CPU compute fallback.

Orientation reminder: E(h,t,r) = sigmoid(score) is a LOSS -- lower is more
plausible; the hinge max(E(pos) - E(neg) + margin, 0) pushes positive scores down.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _lib

MODEL_COMPLEX, MODEL_HOLE, MODEL_HOLE_SPECTRAL, MODEL_HOLE_DIRECT = 0, 1, 2, 3
STEP_DETERMINISTIC = 0x100      # GE_STEP_DETERMINISTIC: OR-ed into the model code of ge_train_steps
_MODELS = {"complex": MODEL_COMPLEX, "hole": MODEL_HOLE, "hole_spectral": MODEL_HOLE_SPECTRAL,
           "hole_direct": MODEL_HOLE_DIRECT, 0: 0, 1: 1, 2: 2, 3: 3}

CORRUPT_BATCH_COIN, CORRUPT_ROW_COIN, CORRUPT_HEADS, CORRUPT_TAILS = 0, 1, 2, 3


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(t: torch.Tensor, name: str):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA (ROCm) tensor: graphembeddings_amd has no CPU path")


def _table(embeddings: torch.Tensor):
    _need_cuda(embeddings, "embeddings")
    if embeddings.dtype != torch.float32 or embeddings.dim() != 2 or not embeddings.is_contiguous():
        raise ValueError("embeddings must be a contiguous float32 [entity_count, embedding_dim] tensor")
    return embeddings


def _triples(t: torch.Tensor, name="triple_batch") -> torch.Tensor:
    """[B,3] (head, tail, relation) (holE.py:181-185). int64 accepted (holE.py:547) and narrowed."""
    _need_cuda(t, name)
    if t.dim() != 2 or t.shape[1] != 3:
        raise ValueError(f"{name} must have shape [B, 3] (head, tail, relation)")
    if t.dtype != torch.int32:
        t = t.to(torch.int32)
    return t.contiguous()


def init_embeddings(entity_count: int, embedding_dim: int, device="cuda", seed: Optional[int] = None):
    """The `embeddings` variable of holE.py:263-264: xavier_initializer(uniform=False) =
    truncated normal (re-drawn beyond 2 sigma) with stddev sqrt(2.6 / (entity_count + dim)), drawn on the
    device (a 960 MB table is not built on the host and copied)."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    if seed is not None:
        gen.manual_seed(seed)
    std = float(np.sqrt(2.6 / (entity_count + embedding_dim)))
    t = torch.empty(entity_count, embedding_dim, dtype=torch.float32, device=dev)
    torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=gen)
    return t


def evaluate_triples(triple_batch: torch.Tensor, embeddings: torch.Tensor, label=None, *,
                     model="complex", max_norm: float = 1.0, apply_sigmoid: bool = True,
                     l2_regularization: float = 0.1) -> torch.Tensor:
    """holE.py:179-202 as [B,1].  label=None (hinge / inference branch): sigmoid(sum_k Re(h_k r_k conj(t_k))).
    label=+1 / -1 (the --log_loss branch, holE.py:194-196; FLAGS.l2_regularization becomes a keyword):
    log(1 + exp(-label * score)) + l2_regularization * l2_loss(embeddings), ComplEx score only.
    model="hole" gives the README.md:42 HolE score instead ("hole_spectral": on a spectral table)."""
    emb = _table(embeddings)
    tb = _triples(triple_batch)
    out = torch.empty(tb.shape[0], dtype=torch.float32, device=emb.device)
    if label is not None:
        if _MODELS[model] != MODEL_COMPLEX:
            raise NotImplementedError("the log-loss branch is defined for the ComplEx score (holE.py:191-196)")
        ws = torch.empty(256, dtype=torch.uint8, device=emb.device)
        _lib.call("ge_complex_logloss", emb.data_ptr(), emb.shape[0], emb.shape[1], tb.data_ptr(), tb.shape[0],
                  float(label), float(l2_regularization), max_norm, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
        return out.view(-1, 1)
    fn = {MODEL_COMPLEX: "ge_complex_score", MODEL_HOLE: "ge_hole_score", MODEL_HOLE_DIRECT: "ge_hole_score",
          MODEL_HOLE_SPECTRAL: "ge_hole_spectral_score"}[_MODELS[model]]
    _lib.call(fn, emb.data_ptr(), emb.shape[0], emb.shape[1], tb.data_ptr(), tb.shape[0],
              max_norm, int(apply_sigmoid), out.data_ptr(), _stream())
    return out.view(-1, 1)


def hole_to_spectral(embeddings: torch.Tensor) -> torch.Tensor:
    """In place: every row of a HolE table -> its packed half spectrum (ge_hole_to_spectral).  On the
    result, model="hole_spectral" scores / trains README.md:42's HolE without any transform."""
    emb = _table(embeddings)
    _lib.call("ge_hole_to_spectral", emb.data_ptr(), emb.shape[0], emb.shape[1], _stream())
    return emb


def hole_from_spectral(embeddings: torch.Tensor) -> torch.Tensor:
    """In place inverse of hole_to_spectral."""
    emb = _table(embeddings)
    _lib.call("ge_hole_from_spectral", emb.data_ptr(), emb.shape[0], emb.shape[1], _stream())
    return emb


@dataclass
class TypeTables:
    """Device form of the two corruption tables of holE.py:267-277.

    id_to_type : int32 [entity_count] type code per table row (-1 unknown -> '?' default, holE.py:39)
    type_offsets / type_ids : CSR of type code -> ids (the full lists of data.type_to_ids; the
    per-batch padded_size subsample of holE.py:343-347 is drawn inside the kernel).
    """
    id_to_type: torch.Tensor
    type_offsets: torch.Tensor
    type_ids: torch.Tensor
    padded_size: int = 1024

    @staticmethod
    def from_host(id_to_type, type_offsets, type_ids, padded_size=1024, device="cuda") -> "TypeTables":
        return TypeTables(
            torch.as_tensor(np.ascontiguousarray(id_to_type, dtype=np.int32)).to(device),
            torch.as_tensor(np.ascontiguousarray(type_offsets, dtype=np.int64)).to(device),
            torch.as_tensor(np.ascontiguousarray(type_ids, dtype=np.int32)).to(device),
            int(padded_size))

    @property
    def n_types(self) -> int:
        return int(self.type_offsets.numel()) - 1


def corrupt_batch(type_tables: TypeTables, relation_count: int, triples: torch.Tensor, *, seed: int = 0,
                  step: int = 0, mode: int = CORRUPT_BATCH_COIN) -> torch.Tensor:
    """holE.py:152-153 (-> corrupt_entities 136-140 -> corrupt_heads/tails 97-133): type-safe
    corruption, [B,3] int32.  relation_count is unused, as in the reference (relation corruption is
    commented out, holE.py:154-158).  (seed, step) key the counter-based random stream."""
    tb = _triples(triples, "triples")
    _need_cuda(type_tables.id_to_type, "type tables")
    neg = torch.empty_like(tb)
    _lib.call("ge_corrupt_batch", tb.data_ptr(), tb.shape[0], type_tables.id_to_type.data_ptr(),
              type_tables.id_to_type.numel(), type_tables.type_offsets.data_ptr(), type_tables.n_types,
              type_tables.type_ids.data_ptr(), int(seed) & (2**64 - 1), int(step) & (2**64 - 1),
              type_tables.padded_size, int(mode), neg.data_ptr(), _stream())
    return neg


def hinge_loss(pos: torch.Tensor, neg: torch.Tensor, embeddings: torch.Tensor, *, margin: float = 0.2,
               model="complex", max_norm: float = 1.0, return_sigmoids: bool = False):
    """max(E(pos) - E(neg) + margin, 0) as [B,1] (holE.py:231) for given negatives."""
    emb = _table(embeddings)
    p, n = _triples(pos, "pos"), _triples(neg, "neg")
    if p.shape != n.shape:
        raise ValueError("pos and neg must have the same shape")
    B = p.shape[0]
    loss = torch.empty(B, dtype=torch.float32, device=emb.device)
    sig = torch.empty(2 * B, dtype=torch.float32, device=emb.device) if return_sigmoids else None
    _lib.call("ge_hinge_loss", emb.data_ptr(), emb.shape[0], emb.shape[1], p.data_ptr(), n.data_ptr(), B,
              margin, max_norm, _MODELS[model], loss.data_ptr(), sig.data_ptr() if sig is not None else None,
              _stream())
    if return_sigmoids:
        return loss.view(-1, 1), sig[:B].view(-1, 1), sig[B:].view(-1, 1)
    return loss.view(-1, 1)


def evaluate_batch(triple_batch: torch.Tensor, embeddings: torch.Tensor, type_to_ids_table: TypeTables,
                   id_to_type_table=None, relation_count: int = 0, *, margin: float = 0.2,
                   model="complex", seed: int = 0, step: int = 0, mode: int = CORRUPT_BATCH_COIN,
                   max_norm: float = 1.0) -> torch.Tensor:
    """holE.py:205-234 (hinge branch): corrupt the batch type-safely, score both, return the hinge
    [B,1].  The reference passes two hash tables; here both live in one TypeTables object
    (`id_to_type_table` is accepted and ignored)."""
    neg = corrupt_batch(type_to_ids_table, relation_count, triple_batch, seed=seed, step=step, mode=mode)
    return hinge_loss(triple_batch, neg, embeddings, margin=margin, model=model, max_norm=max_norm)


def inverse_time_decay(learning_rate: float, global_step: int, decay_steps: float, decay_rate: float) -> float:
    """tf.train.inverse_time_decay as called at holE.py:292-294 (no staircase)."""
    return learning_rate / (1.0 + decay_rate * (float(global_step) / float(decay_steps)))


class HingeSGD:
    """GradientDescentOptimizer(lr).minimize(loss) of holE.py:296 for the hinge of holE.py:231,
    fused on the GPU: gradient of SUM_i loss_i, then ScatterSub with duplicates accumulating.
    Owns the scratch workspace the C ABI needs."""

    def __init__(self, embeddings: torch.Tensor, batch_size: int, *, margin: float = 0.2,
                 model="complex", max_norm: float = 1.0):
        self.embeddings = _table(embeddings)
        self.margin, self.max_norm = float(margin), float(max_norm)
        self.model = _MODELS[model]
        self._ws = None
        self._turn = 0
        self._reserve(batch_size)

    # The gradient rows are written by one kernel and read by the next on other XCDs; rewriting a buffer whose
    # lines still sit in another XCD's L2 is the slow path (profiles/r01_xcd_locality_probe.txt), so small
    # workspaces are rotated (4 regions, the same trick as ge_train_steps); large ones evict themselves.
    _RING, _RING_MAX_BYTES = 4, 48 << 20

    def _reserve(self, B: int):
        need = max(_lib.load().ge_hinge_step_workspace_bytes(B, self.embeddings.shape[1]), 256)
        if self._ws is None or self._ws.shape[1] < need:
            ring = self._RING if need <= self._RING_MAX_BYTES else 1
            self._ws = torch.empty(ring, (need + 255) // 256 * 256, dtype=torch.uint8, device=self.embeddings.device)

    def step(self, pos: torch.Tensor, neg: torch.Tensor, lr: float) -> torch.Tensor:
        """One training step in place on the table; returns the per-pair hinge [B,1]."""
        emb = self.embeddings
        p, n = _triples(pos, "pos"), _triples(neg, "neg")
        if p.shape != n.shape:
            raise ValueError("pos and neg must have the same shape")
        B = p.shape[0]
        self._reserve(B)
        loss = torch.empty(B, dtype=torch.float32, device=emb.device)
        ws = self._ws[self._turn % self._ws.shape[0]]
        self._turn += 1
        if self.model == MODEL_HOLE_SPECTRAL:    # the two halves of the step through their own entry points
            gi = ws[:24 * B].view(torch.int32)
            off = (24 * B + 255) // 256 * 256
            gv = ws[off:off + 24 * B * emb.shape[1]].view(torch.float32).view(6 * B, emb.shape[1])
            _lib.call("ge_hinge_grad", emb.data_ptr(), emb.shape[0], emb.shape[1], p.data_ptr(), n.data_ptr(), B,
                      self.margin, float(lr), self.max_norm, self.model, loss.data_ptr(), gi.data_ptr(), gv.data_ptr(),
                      _stream())
            _lib.call("ge_scatter_add_rows", emb.data_ptr(), emb.shape[0], emb.shape[1], gi.data_ptr(), gv.data_ptr(),
                      6 * B, _stream())
            return loss.view(-1, 1)
        fn = "ge_complex_hinge_step" if self.model == MODEL_COMPLEX else "ge_hole_hinge_step"
        _lib.call(fn, emb.data_ptr(), emb.shape[0], emb.shape[1], p.data_ptr(), n.data_ptr(), B,
                  self.margin, float(lr), self.max_norm, loss.data_ptr(), ws.data_ptr(),
                  ws.numel(), _stream())
        return loss.view(-1, 1)


class LogLossSGD:
    """The --log_loss branch (holE.py:194-196, 206-220): logistic loss over the positives (label +1)
    and `negative_ratio` corrupted batches (label -1) plus l2_regularization * l2_loss(whole table),
    minimised by plain SGD on the SUM of the loss vector (holE.py:296)."""

    def __init__(self, embeddings: torch.Tensor, l2_regularization: float = 0.1, max_norm: float = 1.0):
        self.embeddings = _table(embeddings)
        self.l2, self.max_norm = float(l2_regularization), float(max_norm)
        self._ws = None

    def step(self, pos: torch.Tensor, negs, lr: float) -> torch.Tensor:
        """pos [B,3]; negs: [K,B,3] tensor or list of K [B,3] tensors.  Returns the loss vector
        [(1+K)*B, 1] in the reference's concat order (holE.py:220)."""
        emb = self.embeddings
        p = _triples(pos, "pos")
        if isinstance(negs, (list, tuple)):
            negs = torch.stack([_triples(n, "neg") for n in negs], 0)
        negs = negs.to(torch.int32).reshape(-1, 3)
        tri = torch.cat([p, negs], 0).contiguous()
        M, B = tri.shape[0], p.shape[0]
        labels = torch.ones(M, dtype=torch.float32, device=emb.device)
        labels[B:] = -1.0
        need = _lib.load().ge_logloss_step_workspace_bytes(M, emb.shape[1])
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(need, 256), dtype=torch.uint8, device=emb.device)
        loss = torch.empty(M, dtype=torch.float32, device=emb.device)
        _lib.call("ge_complex_logloss_step", emb.data_ptr(), emb.shape[0], emb.shape[1], tri.data_ptr(),
                  labels.data_ptr(), M, float(lr), self.l2, self.max_norm, loss.data_ptr(), self._ws.data_ptr(),
                  self._ws.numel(), _stream())
        return loss.view(-1, 1)


def hinge_grad(rows: torch.Tensor, pos: torch.Tensor, neg: torch.Tensor, lr: float, *, margin=0.2,
               model="complex", max_norm=1.0):
    """First half of the step (ge_hinge_grad): returns (loss [B], grad_idx [6B] int32, grad_val [6B,d])
    with grad_val already multiplied by -lr.  `rows` may be a staging buffer of fetched rows."""
    rows = _table(rows)
    p, n = _triples(pos, "pos"), _triples(neg, "neg")
    B, d = p.shape[0], rows.shape[1]
    loss = torch.empty(B, dtype=torch.float32, device=rows.device)
    gi = torch.empty(6 * B, dtype=torch.int32, device=rows.device)
    gv = torch.empty(6 * B, d, dtype=torch.float32, device=rows.device)
    _lib.call("ge_hinge_grad", rows.data_ptr(), rows.shape[0], d, p.data_ptr(), n.data_ptr(), B, margin,
              float(lr), max_norm, _MODELS[model], loss.data_ptr(), gi.data_ptr(), gv.data_ptr(), _stream())
    return loss, gi, gv


def scatter_add_rows(table: torch.Tensor, idx: torch.Tensor, val: torch.Tensor) -> None:
    """table[idx[i]] += val[i] for idx[i] >= 0 (ScatterSub with the sign folded into val)."""
    table = _table(table)
    _need_cuda(idx, "idx"); _need_cuda(val, "val")
    idx = idx.to(torch.int32).contiguous()
    val = val.to(torch.float32).contiguous()
    if val.shape != (idx.numel(), table.shape[1]):
        raise ValueError("val must be [len(idx), embedding_dim]")
    _lib.call("ge_scatter_add_rows", table.data_ptr(), table.shape[0], table.shape[1], idx.data_ptr(),
              val.data_ptr(), idx.numel(), _stream())


def gather_rows(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """out[i] = table[idx[i]] (zeros where idx[i] < 0)."""
    table = _table(table)
    _need_cuda(idx, "idx")
    idx = idx.to(torch.int32).contiguous()
    out = torch.empty(idx.numel(), table.shape[1], dtype=torch.float32, device=table.device)
    _lib.call("ge_gather_rows", table.data_ptr(), table.shape[0], table.shape[1], idx.data_ptr(),
              idx.numel(), out.data_ptr(), _stream())
    return out


def score_candidates(embeddings: torch.Tensor, fixed_and_relation: torch.Tensor, candidates: torch.Tensor, *,
                     cand_is_head: bool = False, max_norm: float = 1.0, apply_sigmoid: bool = True) -> torch.Tensor:
    """The candidate sweep of holE.py:564-569 in one call: E for every (fixed_i, cand_j, rel_i)
    (or (cand_j, fixed_i, rel_i) with cand_is_head) as [B,K]; fp32-MFMA GEMM on the GPU."""
    emb = _table(embeddings)
    _need_cuda(fixed_and_relation, "fixed_and_relation"); _need_cuda(candidates, "candidates")
    hr = fixed_and_relation.to(torch.int32).contiguous()
    cand = candidates.to(torch.int32).contiguous().view(-1)
    if hr.dim() != 2 or hr.shape[1] != 2:
        raise ValueError("fixed_and_relation must be [B,2] (entity, relation)")
    out = torch.empty(hr.shape[0], cand.numel(), dtype=torch.float32, device=emb.device)
    _lib.call("ge_complex_score_1vK", emb.data_ptr(), emb.shape[0], emb.shape[1], hr.data_ptr(), hr.shape[0],
              cand.data_ptr(), cand.numel(), max_norm, int(apply_sigmoid), int(cand_is_head), out.data_ptr(),
              _stream())
    return out


def rank_max_dim() -> int:
    return int(_lib.load().ge_rank_max_dim())


class RankPlanes:
    """The candidates of a ranking sweep as the split-precision kernel reads them (ge_rank_planes: fp16 high halves and
    remainders of row * clip scale * 2^8 per 64-candidate tile, plus the entity -> position map), built ONCE for all the
    batches -- tails and heads -- ranked against the same (table, candidates, max_norm, model).  `buffer` is None when
    embedding_dim has no split-precision sweep (rank_candidates then ignores it).  Rebuild after the table changes."""

    def __init__(self, embeddings: torch.Tensor, candidates: torch.Tensor, *, max_norm: float = 1.0, model: str = "complex"):
        if model not in ("complex", "hole_spectral"):
            raise ValueError("RankPlanes: model must be 'complex' or 'hole_spectral'")
        emb = _table(embeddings)
        _need_cuda(candidates, "candidates")
        self.cand = candidates.to(torch.int32).contiguous().view(-1)
        self.key = (emb.data_ptr(), emb.shape[0], emb.shape[1], self.cand.data_ptr(), self.cand.numel(), float(max_norm), model)
        nbytes = int(_lib.load().ge_rank_planes_bytes(emb.shape[0], emb.shape[1], self.cand.numel())) if max_norm <= 8.0 else 0
        self.buffer = None
        if nbytes > 0:
            self.buffer = torch.empty(nbytes, dtype=torch.uint8, device=emb.device)     # (the allocator aligns to 256 bytes and more)
            _lib.call("ge_rank_planes", emb.data_ptr(), emb.shape[0], emb.shape[1], self.cand.data_ptr(), self.cand.numel(),
                      max_norm, _MODELS[model], self.buffer.data_ptr(), _stream())


def rank_candidates(embeddings: torch.Tensor, fixed_and_relation: torch.Tensor, true_ids: torch.Tensor,
                    candidates: torch.Tensor, *, known_off: Optional[torch.Tensor] = None,
                    known_rc: Optional[torch.Tensor] = None, cand_is_head: bool = False, max_norm: float = 1.0,
                    return_true_loss: bool = False, return_scores: bool = False, model: str = "complex",
                    planes: Optional[RankPlanes] = None):
    """The candidate sweep of holE.py:564-569 with the ranking of holE.py:427-472 as its epilogue
    (ge_complex_rank_1vK): per test row the number of candidates that pop from the reference's heap before the
    true one (n_before; raw rank = 1 + n_before) and how many of those are known-true (n_known_before; filtered
    rank = raw - n_known_before).  No [B,K] score matrix exists unless return_scores asks for it (tests).
    known_off / known_rc: the per-(128 rows x 128 candidates)-tile lists of known-true cells (evaluate.py).
    model: "complex", or "hole_spectral" for a table held in the frequency domain (hole_to_spectral).
    planes: RankPlanes(embeddings, candidates, ...) built once for many calls (otherwise the kernel rebuilds them)."""
    if model not in ("complex", "hole_spectral"):
        raise ValueError("rank_candidates: model must be 'complex' or 'hole_spectral' (transform a real HolE table first)")
    emb = _table(embeddings)
    for name, t in (("fixed_and_relation", fixed_and_relation), ("true_ids", true_ids), ("candidates", candidates)):
        _need_cuda(t, name)
    hr = fixed_and_relation.to(torch.int32).contiguous()
    tid = true_ids.to(torch.int32).contiguous().view(-1)
    cand = candidates.to(torch.int32).contiguous().view(-1)
    B, K = hr.shape[0], cand.numel()
    if hr.dim() != 2 or hr.shape[1] != 2 or tid.numel() != B:
        raise ValueError("fixed_and_relation must be [B,2] (entity, relation) and true_ids [B]")
    n_before = torch.empty(B, dtype=torch.int32, device=emb.device)
    n_known = torch.empty(B, dtype=torch.int32, device=emb.device)
    tl = torch.empty(B, dtype=torch.float32, device=emb.device) if return_true_loss else None
    sc = torch.empty(B, K, dtype=torch.float32, device=emb.device) if return_scores else None
    if (known_off is None) != (known_rc is None):
        raise ValueError("known_off and known_rc come together")
    if known_off is not None:
        n_tiles = ((B + 127) // 128) * ((K + 127) // 128)
        if known_off.dtype != torch.int32 or known_off.numel() != n_tiles + 1 or known_rc.dtype != torch.int16:
            raise ValueError("known_off must be int32 [tiles+1], known_rc int16 (row%128 << 7 | col%128)")
    pl = None
    if planes is not None and planes.buffer is not None:
        same = planes.key == (emb.data_ptr(), emb.shape[0], emb.shape[1], planes.cand.data_ptr(), K, float(max_norm), model)
        if same and cand.data_ptr() != planes.cand.data_ptr():      # another tensor: compare the ids (one synchronisation)
            same = bool(torch.equal(planes.cand, cand))
        if not same:
            raise ValueError("rank_candidates: `planes` were built for another table / candidate list / max_norm / model")
        cand, pl = planes.cand, planes.buffer.data_ptr()
    _lib.call("ge_rank_1vK_planes", emb.data_ptr(), emb.shape[0], emb.shape[1], hr.data_ptr(), B, tid.data_ptr(),
              cand.data_ptr(), K, max_norm, _MODELS[model], int(cand_is_head),
              known_off.data_ptr() if known_off is not None else None,
              known_rc.data_ptr() if known_rc is not None else None, n_before.data_ptr(), n_known.data_ptr(),
              tl.data_ptr() if tl is not None else None, sc.data_ptr() if sc is not None else None, pl, _stream())
    out = (n_before, n_known)
    if return_true_loss:
        out += (tl,)
    if return_scores:
        out += (sc,)
    return out


def rank_candidates_vs_loss(embeddings: torch.Tensor, fixed_and_relation: torch.Tensor, ref_ids: torch.Tensor,
                            ref_losses: torch.Tensor, candidates: torch.Tensor, *, known_off: Optional[torch.Tensor] = None,
                            known_rc: Optional[torch.Tensor] = None, cand_is_head: bool = False, max_norm: float = 1.0,
                            model: str = "complex", planes: Optional[RankPlanes] = None):
    """The sweep of rank_candidates against GIVEN losses (ge_rank_1vK_vs_loss): per row how many of `candidates` pop from
    the reference's heap (holE.py:427-472: ascending loss, ties by id) before a triple of loss ref_losses[i] and id
    ref_ids[i] -- which need not be among them -- and how many of those are known-true.  What a rank of a row-sharded
    evaluation computes over its own candidates (the counts add across ranks), and what the reference's is_confident
    gate (holE.py:436-438) reduces to: min_loss < threshold  <=>  n_before(threshold, id = INT32_MIN) > 0."""
    if model not in ("complex", "hole_spectral"):
        raise ValueError("rank_candidates_vs_loss: model must be 'complex' or 'hole_spectral'")
    emb = _table(embeddings)
    for name, t in (("fixed_and_relation", fixed_and_relation), ("ref_ids", ref_ids), ("ref_losses", ref_losses),
                    ("candidates", candidates)):
        _need_cuda(t, name)
    hr = fixed_and_relation.to(torch.int32).contiguous()
    rid = ref_ids.to(torch.int32).contiguous().view(-1)
    rl = ref_losses.to(torch.float32).contiguous().view(-1)
    cand = candidates.to(torch.int32).contiguous().view(-1)
    B, K = hr.shape[0], cand.numel()
    if hr.dim() != 2 or hr.shape[1] != 2 or rid.numel() != B or rl.numel() != B:
        raise ValueError("fixed_and_relation must be [B,2] (entity, relation), ref_ids and ref_losses [B]")
    if (known_off is None) != (known_rc is None):
        raise ValueError("known_off and known_rc come together")
    pl = None
    if planes is not None and planes.buffer is not None:
        same = planes.key == (emb.data_ptr(), emb.shape[0], emb.shape[1], planes.cand.data_ptr(), K, float(max_norm), model)
        if same and cand.data_ptr() != planes.cand.data_ptr():
            same = bool(torch.equal(planes.cand, cand))
        if not same:
            raise ValueError("rank_candidates_vs_loss: `planes` were built for another table / candidate list / max_norm / model")
        cand, pl = planes.cand, planes.buffer.data_ptr()
    n_before = torch.empty(B, dtype=torch.int32, device=emb.device)
    n_known = torch.empty(B, dtype=torch.int32, device=emb.device)
    _lib.call("ge_rank_1vK_vs_loss", emb.data_ptr(), emb.shape[0], emb.shape[1], hr.data_ptr(), B, rid.data_ptr(),
              rl.data_ptr(), cand.data_ptr(), K, max_norm, _MODELS[model], int(cand_is_head),
              known_off.data_ptr() if known_off is not None else None,
              known_rc.data_ptr() if known_rc is not None else None, n_before.data_ptr(), n_known.data_ptr(), pl, _stream())
    return n_before, n_known


def confident_rows(embeddings: torch.Tensor, fixed_and_relation: torch.Tensor, candidates: torch.Tensor, infer_threshold: float,
                   **kw) -> torch.Tensor:
    """is_confident of holE.py:436-438 for every row of a sweep: min over the candidates of the loss < infer_threshold,
    as a bool tensor [B] -- one more sweep, counting the candidates below the threshold (no [B,K] matrix)."""
    B = fixed_and_relation.shape[0]
    dev = embeddings.device
    thr = torch.full((B,), float(infer_threshold), dtype=torch.float32, device=dev)
    lowest = torch.full((B,), -2 ** 31, dtype=torch.int32, device=dev)        # no candidate id is smaller: strict "<" only
    n_before, _ = rank_candidates_vs_loss(embeddings, fixed_and_relation, lowest, thr, candidates, **kw)
    return n_before > 0


class ValidationPocket:
    """The validation ticks of the training loop (holE.py:299-304, 351-360) without a host round trip per tick
    (ge_validation_tick): each tick draws a validation batch on the device, corrupts it, takes the mean hinge into
    `hist`, and -- the reference's "pocket" -- copies the table into `pocket` iff that mean is the best so far.
    The host calls read() when it wants the numbers (one synchronisation for all ticks since the last read)."""

    def __init__(self, embeddings: torch.Tensor, validation_triples: torch.Tensor, type_tables: "TypeTables",
                 batch_size: int, *, margin: float = 0.2, model="complex", max_norm: float = 1.0, seed: int = 0,
                 mode: int = CORRUPT_BATCH_COIN, keep_table: bool = True, capacity: int = 4096, log_loss=None):
        """log_loss = (negative_ratio, l2_regularization): the tick evaluates the --log_loss objective instead of the hinge
        (ge_validation_tick_logloss; ComplEx score)."""
        self.emb = _table(embeddings)
        _need_cuda(validation_triples, "validation_triples")
        self.valid = validation_triples.to(torch.int32).contiguous()
        if self.valid.dim() != 2 or self.valid.shape[1] != 3 or self.valid.shape[0] == 0:
            raise ValueError("validation_triples must be [V,3] with V > 0")
        self.tt, self.B = type_tables, int(batch_size)
        self.margin, self.max_norm, self.model = float(margin), float(max_norm), _MODELS[model]
        self.seed, self.mode = int(seed), int(mode)
        dev = self.emb.device
        self.best = torch.full((), 2.0, dtype=torch.float32, device=dev)     # holE.py:336 pocket_loss = 2.
        self.pocket = torch.empty_like(self.emb) if keep_table else None
        self.hist = torch.zeros(int(capacity), dtype=torch.float32, device=dev)
        self.log_loss = None if log_loss is None else (int(log_loss[0]), float(log_loss[1]))
        if self.log_loss is not None and self.model != MODEL_COMPLEX:
            raise NotImplementedError("--log_loss is defined for the ComplEx score (holE.py:191-196)")
        need = (_lib.load().ge_validation_workspace_bytes(self.B) if self.log_loss is None
                else _lib.load().ge_validation_logloss_workspace_bytes(self.B, self.log_loss[0]))
        self._ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
        self._steps = []        # global step of each tick since the last read()

    def tick(self, counter: int, global_step: int) -> None:
        if len(self._steps) >= self.hist.numel():
            raise RuntimeError("ValidationPocket: read() the pending ticks first (capacity %d)" % self.hist.numel())
        tt, i = self.tt, len(self._steps)
        if self.log_loss is not None:
            _lib.call("ge_validation_tick_logloss", self.emb.data_ptr(), self.emb.shape[0], self.emb.shape[1],
                      self.valid.data_ptr(), self.valid.shape[0], self.B, tt.id_to_type.data_ptr(), tt.type_offsets.data_ptr(),
                      tt.n_types, tt.type_ids.data_ptr(), self.seed & (2**64 - 1), int(counter) & (2**64 - 1), tt.padded_size,
                      self.mode, self.log_loss[0], self.log_loss[1], self.max_norm, self._ws.data_ptr(), self._ws.numel(),
                      self.hist.data_ptr() + 4 * i, self.best.data_ptr(),
                      self.pocket.data_ptr() if self.pocket is not None else None, _stream())
            self._steps.append(int(global_step))
            return
        _lib.call("ge_validation_tick", self.emb.data_ptr(), self.emb.shape[0], self.emb.shape[1], self.valid.data_ptr(),
                  self.valid.shape[0], self.B, tt.id_to_type.data_ptr(), tt.type_offsets.data_ptr(), tt.n_types,
                  tt.type_ids.data_ptr(), self.seed & (2**64 - 1), int(counter) & (2**64 - 1), tt.padded_size, self.mode, self.margin, self.max_norm,
                  self.model, self._ws.data_ptr(), self._ws.numel(), self.hist.data_ptr() + 4 * i, self.best.data_ptr(),
                  self.pocket.data_ptr() if self.pocket is not None else None, _stream())
        self._steps.append(int(global_step))

    def read(self):
        """[(global_step, mean validation hinge)] of the ticks since the last read, in order (synchronises)."""
        n = len(self._steps)
        if n == 0:
            return []
        vals = self.hist[:n].cpu().tolist()
        out = list(zip(self._steps, vals))
        self._steps = []
        return out


def _i32(t: torch.Tensor, name: str) -> torch.Tensor:
    _need_cuda(t, name)
    if t.dtype != torch.int32 or not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous int32")
    return t


# ---------------------------------------------------------------- the row-sharded step (csrc/ge_shard.hip)
def shard_plan(pos: torch.Tensor, neg: torch.Tensor, n_rows: int, world: int, rank: int, peer_mapped: bool = False):
    """ge_shard_plan for S steps: pos, neg [S,B,3] int32 global ids -> (records [S,W], pos_src [S,B,3], neg_src [S,B],
    req_row [S,cap], counts [S,world]) -- see include/ge_hip.h."""
    _i32(pos, "pos"), _i32(neg, "neg")
    S, B = int(pos.shape[0]), int(pos.shape[1])
    lay = prepared_layout(B)
    dev = pos.device
    records = torch.empty(S, lay[0], dtype=torch.int32, device=dev)
    pos_src = torch.empty(S, B, 3, dtype=torch.int32, device=dev)
    neg_src = torch.empty(S, B, dtype=torch.int32, device=dev)
    req_row = torch.empty(S, 4 * lay[1] * lay[2], dtype=torch.int32, device=dev)
    counts = torch.empty(S, world, dtype=torch.int32, device=dev)
    need = int(_lib.load().ge_shard_plan_workspace_bytes(B, S))
    ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
    _lib.call("ge_shard_plan", pos.data_ptr(), neg.data_ptr(), S, B, int(n_rows), int(world), int(rank), records.data_ptr(),
              pos_src.data_ptr(), neg_src.data_ptr(), req_row.data_ptr(), counts.data_ptr(), ws.data_ptr(), ws.numel(),
              int(bool(peer_mapped)), _stream())
    return records, pos_src, neg_src, req_row, counts


def shard_grad(shard: torch.Tensor, staged, pos_src, neg_src, record, B: int, n_rows: int, world: int, lr: float,
               margin: float, model, max_norm: float, grad_idx: torch.Tensor, grad_val: torch.Tensor, gsum,
               peer_shards=None) -> torch.Tensor:
    """ge_shard_grad: one step's fused gather/score/hinge/grad on the shard (in place) + the staged rows; returns loss [B].
    peer_shards (experiment): the `world` shards as tensors mapped into this process -- the other owners' rows are
    then read in place and `staged` is None (the plan must have been built with peer_mapped=True)."""
    import ctypes as C
    shard = _table(shard)
    n_staged = (0 if gsum is None else int(gsum.shape[0])) if peer_shards is not None else (0 if staged is None else int(staged.shape[0]))
    loss = torch.empty(B, dtype=torch.float32, device=shard.device)
    peers = None
    if peer_shards is not None:
        peers = (C.c_void_p * len(peer_shards))(*[t.data_ptr() for t in peer_shards])
    _lib.call("ge_shard_grad", shard.data_ptr(), shard.shape[0], shard.shape[1],
              staged.data_ptr() if (peers is None and n_staged) else None,
              n_staged, pos_src.data_ptr(), neg_src.data_ptr(), record.data_ptr(), int(B), int(n_rows), int(world),
              float(margin), float(lr), float(max_norm), _MODELS[model], loss.data_ptr(), grad_idx.data_ptr(),
              grad_val.data_ptr(), gsum.data_ptr() if n_staged else None, peers, _stream())
    return loss


def shard_apply(shard: torch.Tensor, record, B: int, n_rows: int, world: int, grad_idx, grad_val, gsum) -> None:
    """ge_shard_apply: own rows updated in place, staged rows' gradient sums into gsum."""
    shard = _table(shard)
    _lib.call("ge_shard_apply", shard.data_ptr(), shard.shape[0], shard.shape[1], record.data_ptr(), int(B), int(n_rows),
              int(world), grad_idx.data_ptr(), grad_val.data_ptr(), gsum.data_ptr() if gsum is not None and gsum.numel() else None,
              _stream())


def shard_owner_plan(req_all: torch.Tensor, req_start, rows_local: int):
    """ge_shard_owner_plan: (records [S,W], cap) for the chunk's received request lists; req_start: host list [S+1]."""
    S = len(req_start) - 1
    cap = max([req_start[i + 1] - req_start[i] for i in range(S)] + [0])
    dev = req_all.device
    lib = _lib.load()
    words = int(lib.ge_shard_owner_record_words(cap))
    records = torch.empty(S, max(words, 1), dtype=torch.int32, device=dev)
    if cap:
        rs = torch.tensor(req_start, dtype=torch.int64).to(dev, non_blocking=False)
        ws = torch.empty(max(int(lib.ge_shard_owner_workspace_bytes(cap, S)), 256), dtype=torch.uint8, device=dev)
        _lib.call("ge_shard_owner_plan", _i32(req_all, "req_all").data_ptr(), rs.data_ptr(), S, cap, int(rows_local),
                  records.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    return records, cap


def shard_owner_apply(shard: torch.Tensor, oplan, s: int, recv: torch.Tensor) -> None:
    records, cap = oplan
    if cap == 0 or recv.shape[0] == 0:
        return
    shard = _table(shard)
    _lib.call("ge_shard_owner_apply", shard.data_ptr(), shard.shape[0], shard.shape[1], records[s].data_ptr(), int(cap),
              recv.data_ptr(), _stream())


class Trainer:
    """The inner loop of run_training (holE.py:340-362, minus validation) enqueued natively by
    ge_train_steps: per step a batch of the device-resident shuffled triple array, type-safe
    negatives, lr from inverse_time_decay, one fused hinge SGD step.  No host work per step."""

    def __init__(self, embeddings: torch.Tensor, triples: torch.Tensor, type_tables: TypeTables,
                 batch_size: int, *, margin: float = 0.2, learning_rate: float = 0.1,
                 decay_steps: float = 0.0, decay_rate: float = 0.5, model="complex", max_norm: float = 1.0,
                 seed: int = 0, corrupt_mode: int = CORRUPT_BATCH_COIN, prepared: bool = True,
                 lookahead: bool = True, spectral_resident: bool = False, deterministic: bool = False):
        # deterministic (GE_STEP_DETERMINISTIC): rows with more than 16 gradient slots in a step are reduced in a fixed
        # order instead of by float atomics -- every row bitwise reproducible run to run; one more launch per step
        self.deterministic = bool(deterministic)
        self.embeddings = _table(embeddings)
        self.triples = _triples(triples, "triples")
        if self.triples.shape[0] < batch_size:
            raise ValueError("fewer triples than batch_size (the reference never yields a short batch)")
        self.tt = type_tables
        self.B = int(batch_size)
        self.margin, self.lr0 = float(margin), float(learning_rate)
        self.decay_steps, self.decay_rate = float(decay_steps), float(decay_rate)
        self.model, self.max_norm, self.seed, self.mode = _MODELS[model], float(max_norm), int(seed), int(corrupt_mode)
        self.global_step = 0
        self.row = 0
        dev = self.embeddings.device
        # model="hole" hands ge_train_steps the real-valued table: every run() transforms it to the frequency
        # domain and back (two O(N d^2) passes per call).  spectral_resident=True transforms ONCE, here, keeps
        # `embeddings` spectral between calls (score it with model="hole_spectral") and to_real() undoes it.
        self.spectral = False
        if spectral_resident:
            if self.model != MODEL_HOLE or (self.embeddings.shape[1] & 1):
                raise ValueError("spectral_resident needs model='hole' and an even embedding_dim")
            hole_to_spectral(self.embeddings)
            self.model, self.spectral = MODEL_HOLE_SPECTRAL, True
        # prepared=False: only the single-step workspace -> ge_train_steps takes its fallback branch
        # (per-step sampler launch + float-atomic scatter); lookahead=False: no pipeline handle, the
        # prepare launches go to the caller's stream and nothing survives between run() calls
        lib = _lib.load()
        d = self.embeddings.shape[1]
        need = lib.ge_train_workspace_bytes(self.B, d) if prepared else lib.ge_hinge_step_workspace_bytes(self.B, d)
        self._ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
        self._neg = torch.empty(self.B, 3, dtype=torch.int32, device=dev)
        self.last_loss = torch.zeros(self.B, dtype=torch.float32, device=dev)
        # the prepared-steps pipeline (side stream + look-ahead records that survive between run() calls)
        self._pipe = None
        if prepared and lookahead:
            import ctypes as C
            h = C.c_void_p()
            with torch.cuda.device(dev):
                _lib.call("ge_train_pipeline_create", C.byref(h))
            self._pipe = h.value

    def close(self):
        """Release the pipeline handle (waits for its side stream)."""
        if getattr(self, "_pipe", None):
            _lib.load().ge_train_pipeline_destroy(self._pipe)
            self._pipe = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def to_real(self):
        """Leave the resident-spectral mode: `embeddings` becomes the real-valued HolE table again."""
        if self.spectral:
            hole_from_spectral(self.embeddings)
            self.model, self.spectral = MODEL_HOLE, False

    def real_embeddings(self) -> torch.Tensor:
        """A real-valued copy of the table (checkpoints, evaluation), whatever domain it is held in."""
        return hole_from_spectral(self.embeddings.clone()) if self.spectral else self.embeddings

    def invalidate(self):
        """Forget the records prepared ahead (call after changing `triples` or the type tables in place)."""
        if self._pipe:
            _lib.call("ge_train_pipeline_reset", self._pipe)

    def learning_rate(self, step=None) -> float:
        s = self.global_step if step is None else step
        return inverse_time_decay(self.lr0, s, self.decay_steps, self.decay_rate) if self.decay_steps > 0 else self.lr0

    def prepare_reshuffle(self, generator=None):
        """Draw the NEXT pass order now, on a side stream, beside the steps of the current pass (a device randperm is a
        30-launch radix sort: ~0.35 ms, a fifth of a 116-step FB15k epoch at B = 4096 when it sits between two epochs);
        the next reshuffle() call adopts it.  Same generator, same draw order as reshuffle(generator): identical orders."""
        dev = self.triples.device
        if getattr(self, "_shuffle_stream", None) is None:
            self._shuffle_stream = torch.cuda.Stream(device=dev)
        side, cur = self._shuffle_stream, torch.cuda.current_stream(dev)
        side.wait_stream(cur)                                  # `triples` as it is now (an order is a permutation of it)
        with torch.cuda.stream(side):
            perm = torch.randperm(self.triples.shape[0], device=dev, generator=generator)
            nxt = self.triples[perm].contiguous()
            ready = torch.cuda.Event()
            ready.record(side)
        self.triples.record_stream(side)
        self._next_order = (nxt, ready)

    def reshuffle(self, generator=None):
        """New pass order (the shuffle queue of holE.py:281-283), done on the device -- the one prepare_reshuffle() drew
        ahead, if there is one."""
        ahead, self._next_order = getattr(self, "_next_order", None), None
        if ahead is not None:
            nxt, ready = ahead
            cur = torch.cuda.current_stream(nxt.device)
            cur.wait_event(ready)
            nxt.record_stream(cur)
            self.triples = nxt
        else:
            perm = torch.randperm(self.triples.shape[0], device=self.triples.device, generator=generator)
            self.triples = self.triples[perm].contiguous()
        self.row = 0
        self.invalidate()     # the allocator may hand the new order the address of an older one

    def enable_log_loss(self, negative_ratio: int = 1, l2_regularization: float = 0.1):
        """Switch the native loop to the --log_loss objective (holE.py:194-196, 206-220): run() then enqueues
        ge_train_steps_logloss -- negatives of all `negative_ratio` corrupted batches drawn in the prepare launch,
        row-sorted update, the dense L2 decay carried as one scalar.  last_loss becomes [(1+K)*B]."""
        if self.model != MODEL_COMPLEX:
            raise NotImplementedError("--log_loss is defined for the ComplEx score (holE.py:191-196)")
        self.K, self.l2 = int(negative_ratio), float(l2_regularization)
        dev = self.embeddings.device
        need = _lib.load().ge_train_logloss_workspace_bytes(self.B, self.K, self.embeddings.shape[1])
        self._ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
        self._neg = torch.empty(self.K, self.B, 3, dtype=torch.int32, device=dev)
        self.last_loss = torch.zeros((1 + self.K) * self.B, dtype=torch.float32, device=dev)
        self.invalidate()
        return self

    def _run_logloss(self, n_steps: int, keep_losses: bool):
        emb, T = self.embeddings, self.triples.shape[0]
        M = (1 + self.K) * self.B
        loss = torch.empty(n_steps * M, dtype=torch.float32, device=emb.device) if keep_losses else self.last_loss
        _lib.call("ge_train_steps_logloss", emb.data_ptr(), emb.shape[0], emb.shape[1], self.triples.data_ptr(), T,
                  self.row, self.B, n_steps, self.tt.id_to_type.data_ptr(), self.tt.type_offsets.data_ptr(),
                  self.tt.n_types, self.tt.type_ids.data_ptr(), self.seed & (2**64 - 1), self.global_step,
                  self.tt.padded_size, self.mode, self.K, self.l2, self.lr0, self.decay_steps, self.decay_rate,
                  self.max_norm, loss.data_ptr(), int(keep_losses), self._neg.data_ptr(), self._ws.data_ptr(),
                  self._ws.numel(), self._pipe, _stream())
        row = self.row % T
        for _ in range(n_steps):
            if row + self.B > T:
                row = 0
            row += self.B
        self.row = row
        self.global_step += n_steps
        return loss.view(n_steps, M) if keep_losses else loss

    def run(self, n_steps: int, *, keep_losses: bool = False, events=None, ev_kernel: int = 2):
        """Enqueue n_steps training steps on the current stream; returns the loss tensor
        ([n_steps,B] if keep_losses else the last step's [B])."""
        import ctypes as C
        if getattr(self, "K", 0):
            return self._run_logloss(n_steps, keep_losses)
        emb, T = self.embeddings, self.triples.shape[0]
        loss = (torch.empty(n_steps * self.B, dtype=torch.float32, device=emb.device)
                if keep_losses else self.last_loss)
        evp = None
        if events is not None:
            assert len(events) == 2 * n_steps
            evp = (C.c_void_p * len(events))(*events)
        _lib.call("ge_train_steps", emb.data_ptr(), emb.shape[0], emb.shape[1], self.triples.data_ptr(), T,
                  self.row, self.B, n_steps, self.tt.id_to_type.data_ptr(), self.tt.type_offsets.data_ptr(),
                  self.tt.n_types, self.tt.type_ids.data_ptr(), self.seed & (2**64 - 1), self.global_step,
                  self.tt.padded_size, self.mode, self.margin, self.lr0, self.decay_steps, self.decay_rate,
                  self.max_norm, self.model | (STEP_DETERMINISTIC if self.deterministic else 0), loss.data_ptr(), int(keep_losses),
                  self._neg.data_ptr(), self._ws.data_ptr(), self._ws.numel(), evp, int(ev_kernel), self._pipe, _stream())
        # mirror the C loop's row bookkeeping
        row = self.row % T
        for _ in range(n_steps):
            if row + self.B > T:
                row = 0
            row += self.B
        self.row = row
        self.global_step += n_steps
        return loss.view(n_steps, self.B) if keep_losses else loss


def prepared_layout(batch_size: int):
    """Word offsets of a prepared step record (ge_train_prepared_layout): (words per step, sub-batches,
    pairs per sub-batch, slot_item offset, first sub-batch offset, words per sub-batch, items offset,
    islots offset)."""
    import ctypes as C
    out = (C.c_int64 * 8)()
    _lib.call("ge_train_prepared_layout", int(batch_size), out)
    return tuple(int(v) for v in out)


def prepare_steps(triples: torch.Tensor, type_tables: TypeTables, batch_size: int, n_steps: int, *, first_row: int = 0,
                  seed: int = 0, global_step: int = 0, mode: int = CORRUPT_BATCH_COIN, direct: bool = False) -> torch.Tensor:
    """The prepare launch of ge_train_steps on its own (ge_train_prepare_steps): negatives plus the
    row-sorted work items of `n_steps` consecutive steps as an int32 [n_steps, words] tensor."""
    tb = _triples(triples, "triples")
    lay = prepared_layout(batch_size)
    nbytes = int(_lib.load().ge_train_prepare_bytes(int(batch_size), int(n_steps)))   # records + (B > 4096) sort scratch
    buf = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=tb.device)
    out = buf[:n_steps * lay[0]].view(n_steps, lay[0])
    tt = type_tables
    _lib.call("ge_train_prepare_steps", tb.data_ptr(), tb.shape[0], int(first_row), int(batch_size), int(n_steps),
              tt.id_to_type.data_ptr(), tt.id_to_type.numel(), tt.type_offsets.data_ptr(), tt.n_types,
              tt.type_ids.data_ptr(), int(seed) & (2**64 - 1), int(global_step), tt.padded_size, int(mode),
              int(bool(direct)), buf.data_ptr(), buf.numel() * 4, _stream())
    return out


class Events:
    """hipEvent_t handles from the C ABI (ge_event_*), recorded on the launch stream."""

    def __init__(self, n: int):
        import ctypes as C
        self.handles = []
        for _ in range(n):
            h = C.c_void_p()
            _lib.call("ge_event_create", C.byref(h))
            self.handles.append(h.value)

    def record(self, i: int):
        _lib.call("ge_event_record", self.handles[i], _stream())

    def elapsed_ms(self, i: int, j: int) -> float:
        import ctypes as C
        ms = C.c_float()
        _lib.call("ge_event_elapsed_ms", self.handles[i], self.handles[j], C.byref(ms))
        return float(ms.value)

    def close(self):
        for h in self.handles:
            _lib.load().ge_event_destroy(h)
        self.handles = []


class BernoulliSampler:
    """Filtered Bernoulli negative sampler: the GPU form of the reference's native init.so
    (init.cpp:159-246) for tables that share relation and entity rows (holE.py layout: entities are
    rows [relation_count, entity_count)).  Known triples -> two sorted device indexes + per-relation
    tail-corruption thresholds from tails-per-head / heads-per-tail (init.cpp:107-127, defects fixed)."""

    def __init__(self, known_triples: np.ndarray, relation_count: int, entity_count: int, device="cuda"):
        tri = np.unique(np.asarray(known_triples, dtype=np.int64), axis=0)
        R = int(relation_count)
        self.n_rel, self.ent_lo, self.n_ent = R, R, int(entity_count) - R
        bh = tri[np.lexsort((tri[:, 1], tri[:, 2], tri[:, 0]))]
        bt = tri[np.lexsort((tri[:, 0], tri[:, 2], tri[:, 1]))]
        freq = np.bincount(tri[:, 2], minlength=R).astype(np.float64)
        n_hr = np.bincount(np.unique(tri[:, [0, 2]], axis=0)[:, 1], minlength=R).astype(np.float64)
        n_tr = np.bincount(np.unique(tri[:, [1, 2]], axis=0)[:, 1], minlength=R).astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            tph = np.where(n_hr > 0, freq / n_hr, 1.0)
            hpt = np.where(n_tr > 0, freq / n_tr, 1.0)
        thr = np.minimum(np.floor(hpt / (hpt + tph) * 4294967296.0), 4294967295.0).astype(np.uint32)
        to = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a.astype(dt))).to(device)
        self.bh_key, self.bh_ent = to(bh[:, 0] * R + bh[:, 2], np.int64), to(bh[:, 1], np.int32)
        self.bt_key, self.bt_ent = to(bt[:, 1] * R + bt[:, 2], np.int64), to(bt[:, 0], np.int32)
        self.tail_threshold = torch.as_tensor(thr.view(np.int32)).to(device)   # bit pattern of the uint32s
        self.n_known = int(len(tri))

    def corrupt(self, triples: torch.Tensor, *, seed: int = 0, step: int = 0) -> torch.Tensor:
        tb = _triples(triples, "triples")
        neg = torch.empty_like(tb)
        _lib.call("ge_bernoulli_corrupt_batch", tb.data_ptr(), tb.shape[0], self.bh_key.data_ptr(),
                  self.bh_ent.data_ptr(), self.bt_key.data_ptr(), self.bt_ent.data_ptr(), self.n_known,
                  self.tail_threshold.data_ptr(), self.n_rel, self.ent_lo, self.n_ent,
                  int(seed) & (2**64 - 1), int(step) & (2**64 - 1), neg.data_ptr(), _stream())
        return neg
