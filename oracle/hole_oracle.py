"""CPU oracle (NumPy) for the holE.py hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module.  The product path (`graphembeddings_amd/`) never does: it
fails loudly when the HIP library is missing.

What this is
------------
A restatement, function by function, of the arithmetic of
`/root/reference/holE.py` for the path gather -> max-norm clip -> ComplEx score
-> sigmoid -> pairwise hinge -> gradient of the SUM -> ScatterSub, plus the
type-safe corruption sampler, the LR schedule and the rank/MRR evaluator.
Each function cites the reference lines it follows.

Pinning status: **PINNED against the reference's own recorded training graph for everything around
the score; the ComplEx / README-HolE score lines themselves are restatements.**
The reference has no tests and cannot be executed here (TensorFlow 1.2 is not installed; SURVEY.md
section 8c), but it ships what TensorFlow built from holE.py: holE-20170724/graph.pbtxt, the complete
training GraphDef (forward, TensorFlow's autodiff sub-graph, IndexedSlices concat, ScatterSub).
oracle/graphdef.py reads that file and EXECUTES its nodes with NumPy; tests/golden/graphdef_v1.npz
holds the outputs for seeded inputs (generator: tests/golden/make_graphdef_golden.py) and
tests/test_graphdef_pins.py requires this module's model="graph20170724" -- the graph's historical
score (complex FFT correlation, Sum(Re+Im), tanh) on top of the SAME clip_scale / _clip_backward /
hinge mask / slot order / sgd_step helpers the ComplEx and HolE models use -- to reproduce the loss
vector, the IndexedSlices and the table after the ScatterSub to 1e-12, including the |x| == max_norm
tie, pre-activation == 0 ties, inactive pairs and heavy index duplication.  What that does NOT
cover: the four-op ComplEx score of today's holE.py:191-192 (Mul, Conj, Real, Sum) and the sigmoid
(holE.py:198), which no recorded graph contains; they are checked against torch autograd only.
One decided deviation found this way: an ALL-ZERO row makes the recorded graph emit NaN (RsqrtGrad
multiplies MinimumGrad's zero by rsqrt(0)^3 = inf); here the inactive branch contributes exactly 0.
Other reference-held known answers pinned in tests/test_oracle.py:
  * the Xavier-normal stddev constants stored in the two committed graph
    dumps (holE-20170714/graph.pbtxt, holE-20170724/graph.pbtxt),
  * the inverse-time-decay constants of holE-20170714/graph.pbtxt,
  * the FB15k id files (triples-valid.txt == the raw Freebase valid split
    mapped through the id files, order (head, tail, relation)).

Third-party arithmetic: TensorFlow (un-vendored, unpinned; graph producer 22
/ "1.2.1" per the committed meta-graphs).  Op semantics restated here:
  tf.nn.embedding_lookup(max_norm=1) -> tf.clip_by_norm(axes=[1..]):
      l2norm_inv = rsqrt(sum(t*t));  y = (t*clip) * minimum(l2norm_inv, 1/clip)
      (op chain visible at holE-20170724/graph.pbtxt:3108-3596)
  MinimumGrad routes the gradient to l2norm_inv when l2norm_inv <= 1/clip,
  MaximumGrad routes to x when x >= y,
  GradientDescentOptimizer on a gathered variable -> ScatterSub of the
  concatenated IndexedSlices, duplicates all applied
      (holE-20170724/graph.pbtxt:47850-48001).
"""
from __future__ import annotations

import heapq
from collections import defaultdict

import numpy as np

# --------------------------------------------------------------------------
# a1. table init  (holE.py:263-264)
# --------------------------------------------------------------------------

def xavier_normal_stddev(n_rows: int, dim: int) -> float:
    """tf.contrib.layers.xavier_initializer(uniform=False) (holE.py:264).

    = variance_scaling_initializer(factor=1, mode='FAN_AVG', uniform=False):
    truncated normal with stddev sqrt(1.3 * factor / ((fan_in+fan_out)/2)).
    Known answers held by the reference: 0.00850143656135 for [35910, 64]
    (holE-20170714/graph.pbtxt) and 0.00151367869694 for [1134637, 128]
    (holE-20170724/graph.pbtxt).
    """
    return float(np.sqrt(2.6 / (n_rows + dim)))


def init_table(n_rows: int, dim: int, seed: int = 0, dtype=np.float32) -> np.ndarray:
    """Truncated normal (re-draw beyond 2 sigma), stddev as above (holE.py:263-264).

    Distribution-level restatement only: TF's RNG stream is not reproduced.
    """
    rng = np.random.default_rng(seed)
    std = xavier_normal_stddev(n_rows, dim)
    x = rng.standard_normal((n_rows, dim))
    bad = np.abs(x) > 2.0
    while bad.any():
        x[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(x) > 2.0
    return (x * std).astype(dtype)


# --------------------------------------------------------------------------
# a2. get_embedding  (holE.py:161-168)
# --------------------------------------------------------------------------

def clip_scale(rows: np.ndarray, max_norm: float = 1.0) -> np.ndarray:
    """min(rsqrt(sum x^2), 1/max_norm) * max_norm per row (clip_by_norm chain).

    A zero row gives rsqrt(0)=inf -> min=1/max_norm -> scale 1 (y = 0).
    """
    ss = np.sum(rows * rows, axis=-1, keepdims=True)
    with np.errstate(divide="ignore"):
        inv = 1.0 / np.sqrt(ss)
    return np.minimum(inv, 1.0 / max_norm) * max_norm


def get_embedding(ids: np.ndarray, table: np.ndarray, max_norm: float = 1.0):
    """holE.py:161-168: gather, clip, split first half = Re, second half = Im."""
    rows = table[np.asarray(ids).reshape(-1)]
    y = rows * clip_scale(rows, max_norm)
    k = table.shape[1] // 2
    return y[:, :k] + 1j * y[:, k:]


# --------------------------------------------------------------------------
# a3/a4. evaluate_triples  (holE.py:179-202)
# --------------------------------------------------------------------------

def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def complex_score(triples: np.ndarray, table: np.ndarray, max_norm: float = 1.0) -> np.ndarray:
    """Raw score  sum_k Re(h_k * r_k * conj(t_k))  (holE.py:191-192).

    triples columns are (head, tail, relation) (holE.py:181-185).
    """
    triples = np.asarray(triples)
    h = get_embedding(triples[:, 0], table, max_norm)
    t = get_embedding(triples[:, 1], table, max_norm)
    r = get_embedding(triples[:, 2], table, max_norm)
    return np.sum((h * (r * np.conj(t))).real, axis=1)


def evaluate_triples(triples, table, max_norm: float = 1.0) -> np.ndarray:
    """sigma(score), shape [B, 1] (holE.py:198, hinge / inference mode)."""
    return sigmoid(complex_score(triples, table, max_norm))[:, None]


# --------------------------------------------------------------------------
# HolE (config 3): README.md:42  E = sigmoid(r . ifft(conj(fft(h)) * fft(t)))
# --------------------------------------------------------------------------

def hole_score(triples, table, max_norm: float = 1.0) -> np.ndarray:
    """HolE raw score  sum_k r_k [h (star) t]_k  with circular correlation
    [h star t]_k = sum_i h_i t_{(i+k) mod d} = ifft(conj(fft(h)) fft(t))
    (README.md:42).  Rows are the full d real values (clipped as in a2)."""
    triples = np.asarray(triples)
    def rows(col):
        x = table[triples[:, col]]
        return x * clip_scale(x, max_norm)
    h, t, r = rows(0), rows(1), rows(2)
    corr = np.fft.ifft(np.conj(np.fft.fft(h, axis=1)) * np.fft.fft(t, axis=1), axis=1).real
    return np.sum(r * corr, axis=1)


def hole_score_direct(triples, table, max_norm: float = 1.0) -> np.ndarray:
    """O(d^2) definition of the same score (cross-check of the FFT form)."""
    triples = np.asarray(triples)
    d = table.shape[1]
    out = np.zeros(len(triples), dtype=np.float64)
    idx = (np.arange(d)[:, None] + np.arange(d)[None, :]) % d  # [i, k] -> (i+k)%d
    for n, (hi, ti, ri) in enumerate(triples):
        h = table[hi] * clip_scale(table[hi][None], max_norm)[0]
        t = table[ti] * clip_scale(table[ti][None], max_norm)[0]
        r = table[ri] * clip_scale(table[ri][None], max_norm)[0]
        corr = (h[:, None] * t[idx]).sum(axis=0)
        out[n] = np.dot(r, corr)
    return out


def hole_graph20170724_evaluate(triples, table, max_norm: float = 1.0) -> np.ndarray:
    """Compatibility restatement of the HISTORICAL HolE variant recorded in the reference's
    holE-20170724/graph.pbtxt:6221-6521 (not what holE.py computes today; oracle-only, no kernel):
    rows as k = d/2 complex numbers (get_embedding, holE.py:161-168), then the op chain
    FFT(h), Conj, FFT(t), Mul, IFFT  ->  c = complex circular correlation, c_m = sum_i conj(h_i) t_{(i+m) mod k};
    Mul(r, c); Real + Imag; Sum over m; Tanh.  Returns tanh(score) [B,1]."""
    triples = np.asarray(triples)
    h = get_embedding(triples[:, 0], table, max_norm)
    t = get_embedding(triples[:, 1], table, max_norm)
    r = get_embedding(triples[:, 2], table, max_norm)
    c = np.fft.ifft(np.conj(np.fft.fft(h, axis=1)) * np.fft.fft(t, axis=1), axis=1)
    p = r * c
    return np.tanh(np.sum(p.real + p.imag, axis=1))[:, None]


def hole_graph20170724_evaluate_direct(triples, table, max_norm: float = 1.0) -> np.ndarray:
    """O(k^2) definition of the same quantity (cross-check of the FFT form)."""
    triples = np.asarray(triples)
    h = get_embedding(triples[:, 0], table, max_norm)
    t = get_embedding(triples[:, 1], table, max_norm)
    r = get_embedding(triples[:, 2], table, max_norm)
    k = h.shape[1]
    idx = (np.arange(k)[:, None] + np.arange(k)[None, :]) % k           # [i, m] -> (i+m) % k
    c = np.einsum("bi,bim->bm", np.conj(h), t[:, idx])
    p = r * c
    return np.tanh(np.sum(p.real + p.imag, axis=1))[:, None]


def hole_evaluate_triples(triples, table, max_norm: float = 1.0):
    return sigmoid(hole_score(triples, table, max_norm))[:, None]


# --------------------------------------------------------------------------
# a5. evaluate_batch hinge  (holE.py:222-234)
# --------------------------------------------------------------------------

def evaluate_batch(pos, neg, table, margin: float = 0.2, max_norm: float = 1.0,
                   model: str = "complex") -> np.ndarray:
    """max(E(pos) - E(neg) + margin, 0), shape [B,1] (holE.py:231)."""
    ev = evaluate_triples if model == "complex" else hole_evaluate_triples
    return np.maximum(ev(pos, table, max_norm) - ev(neg, table, max_norm) + margin, 0.0)


# --------------------------------------------------------------------------
# a6. closed-form gradient of sum_i L_i  (holE.py:296 minimize(non-scalar) => sum)
# --------------------------------------------------------------------------

def _clip_backward(x: np.ndarray, gy: np.ndarray, max_norm: float = 1.0) -> np.ndarray:
    """Gradient through y = x * min(rsqrt(ss), 1/c) * c.

    MinimumGrad: flows to rsqrt branch iff rsqrt(ss) <= 1/c  (ss >= c^2).
    Active branch: y = c x / |x|  ->  gx = c (gy/|x| - x (gy.x)/|x|^3).
    """
    ss = np.sum(x * x, axis=-1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / np.sqrt(ss)
        active = inv <= (1.0 / max_norm)
        g_active = max_norm * (gy * inv - x * np.sum(gy * x, axis=-1, keepdims=True) * inv ** 3)
    return np.where(active, g_active, gy)


def _side_grads(triples, table, coef, max_norm, model):
    """d(sum coef_i * s_i)/d(raw rows) for one side. Returns (gh, gt, gr) [B,d]."""
    triples = np.asarray(triples)
    xh, xt, xr = table[triples[:, 0]], table[triples[:, 1]], table[triples[:, 2]]
    yh = xh * clip_scale(xh, max_norm)
    yt = xt * clip_scale(xt, max_norm)
    yr = xr * clip_scale(xr, max_norm)
    d = table.shape[1]
    if model == "complex":
        k = d // 2
        a, b = yh[:, :k], yh[:, k:]
        e, f = yt[:, :k], yt[:, k:]
        c, dd = yr[:, :k], yr[:, k:]
        gh = np.concatenate([c * e + dd * f, c * f - dd * e], axis=1)
        gt = np.concatenate([a * c - b * dd, a * dd + b * c], axis=1)
        gr = np.concatenate([a * e + b * f, a * f - b * e], axis=1)
    elif model == "graph20170724":
        # the HISTORICAL variant recorded in holE-20170724/graph.pbtxt:6221-6521 (see
        # hole_graph20170724_evaluate): s = sum_m (Re + Im)(r_m c_m) = Re((1-i) sum_m r_m c_m),
        # c_m = sum_i conj(h_i) t_{i+m}.  With g = ds/dRe + i ds/dIm per complex component:
        #   g_r = conj((1-i) c),  g_t = (1+i) (conj(r) (*) h)  [circular convolution],
        #   g_h = (1-i) sum_m r_m t_{i+m}.
        k = d // 2
        ch, ct, cr = (v[:, :k] + 1j * v[:, k:] for v in (yh, yt, yr))
        al = 1.0 - 1.0j
        fh, ft, fcr = np.fft.fft(ch, axis=1), np.fft.fft(ct, axis=1), np.fft.fft(np.conj(cr), axis=1)
        g_r = np.conj(al * np.fft.ifft(np.conj(fh) * ft, axis=1))
        g_t = np.conj(al) * np.fft.ifft(fcr * fh, axis=1)
        g_h = al * np.fft.ifft(np.conj(fcr) * ft, axis=1)
        gh, gt, gr = (np.concatenate([g.real, g.imag], axis=1) for g in (g_h, g_t, g_r))
    else:  # hole
        fh, ft, fr = (np.fft.fft(v, axis=1) for v in (yh, yt, yr))
        # s = sum_k r_k sum_i h_i t_{i+k}
        gr = np.fft.ifft(np.conj(fh) * ft, axis=1).real            # (h star t)
        gh = np.fft.ifft(np.conj(fr) * ft, axis=1).real            # (r star t)
        gt = np.fft.ifft(fr * fh, axis=1).real                     # (r conv h)
    cc = coef[:, None]
    return (_clip_backward(xh, cc * gh, max_norm),
            _clip_backward(xt, cc * gt, max_norm),
            _clip_backward(xr, cc * gr, max_norm))


def hinge_grads(pos, neg, table, margin: float = 0.2, max_norm: float = 1.0,
                model: str = "complex"):
    """IndexedSlices of d(sum_i L_i)/d(table): (indices [6B], values [6B,d], loss [B]).

    Order of the concatenation follows the committed graph
    (holE-20170724/graph.pbtxt:47850-48001): r+, r-, t+, t-, h+, h-.
    m_i = 1[sigma(s+) - sigma(s-) + margin >= 0]   (MaximumGrad, x >= y)
    c+_i = m_i sigma'(s+),  c-_i = -m_i sigma'(s-).
    """
    pos, neg = np.asarray(pos), np.asarray(neg)
    if model == "graph20170724":     # tanh activation (holE-20170724/graph.pbtxt:6520), same hinge around it
        sp, sn = hole_graph20170724_evaluate(pos, table, max_norm)[:, 0], hole_graph20170724_evaluate(neg, table, max_norm)[:, 0]
        dsp, dsn = 1 - sp * sp, 1 - sn * sn
    else:
        sfun = complex_score if model == "complex" else hole_score
        sp, sn = sigmoid(sfun(pos, table, max_norm)), sigmoid(sfun(neg, table, max_norm))
        dsp, dsn = sp * (1 - sp), sn * (1 - sn)
    pre = sp - sn + margin
    m = (pre >= 0).astype(table.dtype)
    loss = np.maximum(pre, 0.0)
    ghp, gtp, grp = _side_grads(pos, table, m * dsp, max_norm, model)
    ghn, gtn, grn = _side_grads(neg, table, -m * dsn, max_norm, model)
    idx = np.concatenate([pos[:, 2], neg[:, 2], pos[:, 1], neg[:, 1], pos[:, 0], neg[:, 0]])
    val = np.concatenate([grp, grn, gtp, gtn, ghp, ghn], axis=0)
    return idx, val, loss


# --------------------------------------------------------------------------
# a7. sparse SGD apply  (holE.py:296 -> ScatterSub, duplicates accumulate)
# --------------------------------------------------------------------------

def sgd_step(table, pos, neg, lr: float, margin: float = 0.2, max_norm: float = 1.0,
             model: str = "complex"):
    """One reference training step. Returns (new_table, loss[B])."""
    idx, val, loss = hinge_grads(pos, neg, table, margin, max_norm, model)
    new = table.copy()
    np.subtract.at(new, idx, (lr * val).astype(table.dtype))
    return new, loss


# --------------------------------------------------------------------------
# a8. LR schedule  (holE.py:291-295)
# --------------------------------------------------------------------------

def inverse_time_decay(lr0: float, step: int, decay_steps: float, decay_rate: float) -> float:
    """lr0 / (1 + decay_rate * step / decay_steps), no staircase
    (op chain holE-20170714/graph.pbtxt:15847-16110: Cast, RealDiv, Mul, Add, RealDiv)."""
    return lr0 / (1.0 + decay_rate * (float(step) / float(decay_steps)))


# --------------------------------------------------------------------------
# a9/a10. type-safe corruption  (holE.py:97-140 graph side, 343-347 host refill)
# --------------------------------------------------------------------------
# The reference draws, per batch, a with-replacement `padded_size` subsample of
# every type's id list (host, holE.py:343-344), then per row one uniform slot in
# [0, padded_size) of its type's subsample (holE.py:108-112), and ONE coin per
# batch to pick heads or tails (holE.py:137-140).  TF's and CPython's RNG
# streams are not part of the contract; the oracle and the HIP sampler share a
# counter-based Philox4x32-10 stream instead, defined here:
#     key   = (seed_lo ^ TAG, seed_hi)
#     coin  : ctr = (step_lo, step_hi, 0, 0),  TAG_COIN ; heads iff top bit == 0
#     slot_i: ctr = (step_lo, step_hi, i, 0),  TAG_SLOT ; slot = w0 % padded_size
#     pick  : ctr = (step_lo, step_hi, type, slot), TAG_PICK ;
#             index = (w0 * len_type) >> 32 ;  id = type_ids[type_off + index]
# which reproduces the reference's JOINT distribution (rows of one batch that
# share a type and draw the same slot get the same id) without materialising
# the [n_types, padded_size] table.  padded_size == 0 selects a plain uniform
# pick over the whole type list (index from the slot stream).

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = 0x9E3779B9
PHILOX_W1 = 0xBB67AE85
TAG_COIN = 0x636F696E  # 'coin'
TAG_SLOT = 0x736C6F74  # 'slot'
TAG_PICK = 0x7069636B  # 'pick'
TAG_SIDE = 0x73696465  # 'side' (per-row coin, mode 1)

MODE_BATCH_COIN = 0   # reference behaviour (holE.py:137-140)
MODE_ROW_COIN = 1     # per-row side choice (extension)
MODE_HEADS = 2        # always corrupt heads (holE.py:97-114)
MODE_TAILS = 3        # always corrupt tails (holE.py:117-133)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 (Salmon et al. 2011). All args uint32 arrays/scalars."""
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint64) & np.uint64(0xFFFFFFFF) for v in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = PHILOX_M0 * c0
        p1 = PHILOX_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)) & mask, lo1, (hi0 ^ c3 ^ np.uint64(k1)) & mask, lo0
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return tuple(v.astype(np.uint32) for v in (c0, c1, c2, c3))


def batch_coin_is_heads(seed: int, step: int) -> bool:
    s_lo, s_hi = step & 0xFFFFFFFF, (step >> 32) & 0xFFFFFFFF
    w = philox4x32_10(s_lo, s_hi, 0, 0, (seed & 0xFFFFFFFF) ^ TAG_COIN, (seed >> 32) & 0xFFFFFFFF)[0]
    return (int(w) >> 31) == 0


def corrupt_batch(pos, id_to_type, type_offsets, type_ids, seed: int, step: int,
                  padded_size: int = 1024, mode: int = MODE_BATCH_COIN) -> np.ndarray:
    """holE.py:152-153 -> corrupt_entities (136-140) -> corrupt_heads/tails (97-133).

    id_to_type[int32 N] (-1 = unknown id => corrupted id is -1, the reference's
    default row of -1s, holE.py:39); CSR type_offsets[int64 n_types+1], type_ids[int32].
    """
    pos = np.asarray(pos, dtype=np.int32)
    B = pos.shape[0]
    s_lo, s_hi = step & 0xFFFFFFFF, (step >> 32) & 0xFFFFFFFF
    k_lo, k_hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    rows = np.arange(B, dtype=np.uint64)
    if mode == MODE_BATCH_COIN:
        heads = np.full(B, batch_coin_is_heads(seed, step))
    elif mode == MODE_ROW_COIN:
        w = philox4x32_10(s_lo, s_hi, rows & 0xFFFFFFFF, rows >> 32, k_lo ^ TAG_SIDE, k_hi)[0]
        heads = (w >> 31) == 0
    else:
        heads = np.full(B, mode == MODE_HEADS)
    col = np.where(heads, 0, 1)
    x = pos[rows.astype(np.int64), col].astype(np.int64)
    n = len(id_to_type)
    known = (x >= 0) & (x < n)
    typ = np.where(known, np.asarray(id_to_type)[np.clip(x, 0, n - 1)], -1).astype(np.int64)
    known &= typ >= 0
    typ_c = np.clip(typ, 0, len(type_offsets) - 2)
    off = np.asarray(type_offsets)[typ_c].astype(np.int64)
    ln = (np.asarray(type_offsets)[typ_c + 1] - off).astype(np.uint64)
    known &= ln > 0
    w_slot = philox4x32_10(s_lo, s_hi, rows & 0xFFFFFFFF, rows >> 32, k_lo ^ TAG_SLOT, k_hi)[0]
    if padded_size > 0:
        slot = w_slot.astype(np.uint64) % np.uint64(padded_size)
        w_pick = philox4x32_10(s_lo, s_hi, typ_c.astype(np.uint64), slot, k_lo ^ TAG_PICK, k_hi)[0]
    else:
        w_pick = w_slot
    index = ((w_pick.astype(np.uint64) * ln) >> np.uint64(32)).astype(np.int64)
    new_id = np.where(known, np.asarray(type_ids)[np.clip(off + index, 0, len(type_ids) - 1)], -1)
    neg = pos.copy()
    neg[rows.astype(np.int64), col] = new_id.astype(np.int32)
    return neg


# --------------------------------------------------------------------------
# a12. rank + MRR  (holE.py:427-490)
# --------------------------------------------------------------------------

def eval_link_prediction(scores, triples, true_triples, test_triples,
                         raw_positions, filtered_positions,
                         infer_threshold=None):
    """Ranking logic of holE.py:427-472 without printing / file output.

    scores: [C] sigma(score) per candidate; triples: [C,3] (h,t,r).
    Ascending pop order by (loss, triple tuple) (holE.py:434); raw_rank counts
    every pop; a tail in true_triples[h][r] is skipped for the filtered rank
    (holE.py:454-461); ranks are recorded when the tail is in
    test_triples[h][r] (holE.py:464-466).  `is_confident` gating
    (holE.py:438) applies only when infer_threshold is given.
    """
    heap = []
    min_loss = 100
    for loss, tr in zip(scores, triples):
        loss = float(loss)
        min_loss = min(min_loss, loss)
        heapq.heappush(heap, (loss, tuple(int(v) for v in tr)))
    is_confident = True if infer_threshold is None else (min_loss < infer_threshold)
    raw_rank = filtered_rank = 0
    while heap:
        loss, (h, t, r) = heapq.heappop(heap)
        raw_rank += 1
        in_sample = t in true_triples[h][r]
        if is_confident and in_sample:
            continue
        filtered_rank += 1
        if is_confident and t in test_triples[h][r]:
            raw_positions.append(raw_rank)
            filtered_positions.append(filtered_rank)


def score_mrr(raw_positions, filtered_positions):
    """holE.py:475-490. Returns dict(raw_mrr, filtered_mrr, mean positions, hits@1/3/10 in %)."""
    raw = np.array(raw_positions, dtype=np.float64)
    fil = np.array(filtered_positions, dtype=np.float64)
    return {
        "raw_mrr": float(np.mean(1.0 / raw)), "mean_raw_pos": float(np.mean(raw)),
        "filtered_mrr": float(np.mean(1.0 / fil)), "mean_filtered_pos": float(np.mean(fil)),
        "hits1": float(np.mean(fil <= 1) * 100), "hits3": float(np.mean(fil <= 3) * 100),
        "hits10": float(np.mean(fil <= 10) * 100),
    }


def triple_dict(triples):
    d = defaultdict(lambda: defaultdict(set))
    for h, t, r in np.asarray(triples):
        d[int(h)][int(r)].add(int(t))
    return d


# --------------------------------------------------------------------------
# reference host loop pieces used by the CPU baseline (holE.py:343-347)
# --------------------------------------------------------------------------

def reference_padded_resample(type_lists, padded_size, rng):
    """The per-batch host loop of holE.py:343-344 verbatim in behaviour:
    for every type, padded_size random.choice() draws with replacement."""
    return np.array([[rng.choice(v) for _ in range(padded_size)] for v in type_lists])


# --------------------------------------------------------------------------
# f2. logistic-loss mode  (holE.py:194-196, 206-220; flags --log_loss --negative_ratio --l2_regularization)
# --------------------------------------------------------------------------

def logloss_values(triples, labels, table, l2: float, max_norm: float = 1.0) -> np.ndarray:
    """holE.py:195-196 per triple:  log(1 + exp(-label * score)) + l2 * tf.nn.l2_loss(embeddings),
    l2_loss = sum(table**2)/2 over the WHOLE table (added to every row of the loss vector)."""
    s = complex_score(triples, table, max_norm)
    y = np.asarray(labels, dtype=table.dtype)
    return np.log1p(np.exp(-y * s)) + l2 * 0.5 * np.sum(table * table)


def logloss_step(table, pos, negs, lr: float, l2: float, max_norm: float = 1.0):
    """One step of the --log_loss branch (holE.py:206-220 then minimize, holE.py:296): the loss
    vector is concat(positives with label +1, negative_ratio corrupted batches with label -1);
    minimize() differentiates its SUM, so the dense L2 term is counted once per loss row:
        grad = sum_i (-y_i sigma(-y_i s_i)) ds_i/dtable  +  M * l2 * table,   M = (1+K) * B
        new  = table - lr * grad.
    negs: [K,B,3].  Returns (new_table, loss [M])."""
    pos = np.asarray(pos)
    negs = np.asarray(negs).reshape(-1, pos.shape[0], 3)
    triples = np.concatenate([pos] + [n for n in negs], 0)
    labels = np.concatenate([np.ones(len(pos)), -np.ones(len(triples) - len(pos))]).astype(table.dtype)
    loss = logloss_values(triples, labels, table, l2, max_norm)
    s = complex_score(triples, table, max_norm)
    coef = -labels * sigmoid(-labels * s)
    gh, gt, gr = _side_grads(triples, table, coef, max_norm, "complex")
    new = table * (1.0 - lr * len(triples) * l2)
    np.subtract.at(new, triples[:, 0], lr * gh)
    np.subtract.at(new, triples[:, 1], lr * gt)
    np.subtract.at(new, triples[:, 2], lr * gr)
    return new, loss
