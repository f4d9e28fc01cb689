"""Pins and cross-checks of the CPU oracle (oracle/): runs without a GPU.

What pins exist (the reference has no tests):
  * constants the reference itself committed in its graph dumps (init stddev, LR decay),
  * the published Random123 known-answer vectors for Philox4x32-10 (the sampler's stream),
  * the FB15k id files (see test_data.py),
  * an independent torch-autograd implementation of the same TF op chain (second opinion).
"""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as CO
from oracle import hole_oracle as O


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "golden_v1.npz"))


# ---------------------------------------------------------------- reference-held known answers
def test_xavier_stddev_matches_reference_graph_dumps():
    # holE-20170714/graph.pbtxt  embeddings [35910, 64]   stddev const 0.00850143656135
    # holE-20170724/graph.pbtxt  embeddings [1134637,128] stddev const 0.00151367869694
    assert O.xavier_normal_stddev(35910, 64) == pytest.approx(0.00850143656135, rel=1e-7)
    assert O.xavier_normal_stddev(1134637, 128) == pytest.approx(0.00151367869694, rel=1e-7)


def test_init_table_distribution():
    t = O.init_table(4000, 64, seed=1)
    std = O.xavier_normal_stddev(4000, 64)
    assert np.abs(t).max() <= 2 * std * (1 + 1e-6)          # truncated at 2 sigma
    assert t.std() == pytest.approx(std * 0.8796, rel=0.02)  # std of a 2-sigma truncated normal


def test_inverse_time_decay_reference_constants(G):
    # recorded run A (holE-20170714/graph.pbtxt:15847-16110): lr 0.01, decay_steps 6192, rate 0.5
    assert O.inverse_time_decay(0.01, 0, 6192, 0.5) == pytest.approx(0.01)
    assert O.inverse_time_decay(0.01, 6192, 6192, 0.5) == pytest.approx(0.01 / 1.5)
    assert O.inverse_time_decay(0.01, 3 * 6192, 6192, 0.5) == pytest.approx(0.01 / 2.5)
    for s, v in zip(G["lr_steps"], G["lr_values"]):
        assert O.inverse_time_decay(0.1, int(s), 32 * 943, 0.5) == pytest.approx(float(v), rel=1e-12)


def test_philox_random123_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    kat = [((0, 0, 0, 0, 0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
           ((0xFFFFFFFF,) * 6, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
           ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0),
            (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1))]
    for inp, exp in kat:
        got = tuple(int(v) for v in O.philox4x32_10(*inp))
        assert got == exp


# ---------------------------------------------------------------- torch autograd second opinion
def _torch_step(table, pos, neg, margin, lr, model):
    """The TF op chain of holE.py:161-234,296 written with torch ops in fp64 + autograd."""
    emb = torch.tensor(table, dtype=torch.float64, requires_grad=True)

    def lookup(ids):
        t = emb[torch.as_tensor(ids, dtype=torch.long)]
        l2inv = torch.rsqrt((t * t).sum(dim=1, keepdim=True))
        return t * torch.minimum(l2inv, torch.ones_like(l2inv))   # clip_by_norm, max_norm = 1

    def evaluate(tr):
        h, t, r = lookup(tr[:, 0]), lookup(tr[:, 1]), lookup(tr[:, 2])
        if model == "complex":
            k = table.shape[1] // 2
            hc, tc, rc = (torch.complex(v[:, :k], v[:, k:]) for v in (h, t, r))
            s = (hc * (rc * torch.conj(tc))).real.sum(dim=1)
        else:
            corr = torch.fft.ifft(torch.conj(torch.fft.fft(h, dim=1)) * torch.fft.fft(t, dim=1), dim=1).real
            s = (r * corr).sum(dim=1)
        return torch.sigmoid(s)

    loss = torch.clamp(evaluate(pos) - evaluate(neg) + margin, min=0.0)
    loss.sum().backward()
    return (emb.detach() - lr * emb.grad).numpy(), loss.detach().numpy(), emb.grad.numpy()


@pytest.mark.parametrize("model", ["complex", "hole"])
@pytest.mark.parametrize("d", [50, 128, 200])
def test_closed_form_step_matches_torch_autograd(G, d, model):
    table = G[f"d{d}_table"].astype(np.float64)
    pos, neg = G[f"d{d}_pos"], G[f"d{d}_neg"]
    # rows 0 (zero norm) and 1 (norm exactly 1) sit on non-differentiable points of the clip where
    # torch's minimum() splits the gradient but TF's MinimumGrad routes it one way; compare on the
    # pairs that avoid them, the tie rule itself is tested separately below.
    keep = ~np.isin(pos, [0, 1]).any(axis=1) & ~np.isin(neg, [0, 1]).any(axis=1)
    pos, neg = pos[keep], neg[keep]
    new, loss = O.sgd_step(table, pos, neg, lr=0.05, margin=0.2, model=model)
    tnew, tloss, _ = _torch_step(table, pos, neg, 0.2, 0.05, model)
    assert np.abs(loss - tloss).max() < 1e-12
    assert np.abs(new - tnew).max() < 1e-12


def test_clip_tie_routes_to_rsqrt_branch():
    # ||x|| == 1: MinimumGrad (x <= y) sends the gradient through rsqrt -> radial part projected out
    x = np.zeros((1, 8)); x[0, 3] = 1.0
    gy = np.arange(8, dtype=np.float64)[None]
    gx = O._clip_backward(x, gy)
    exp = gy.copy(); exp[0, 3] = 0.0
    assert np.allclose(gx, exp)
    # ||x|| < 1: identity ; zero row: identity, no NaN
    assert np.allclose(O._clip_backward(0.5 * x, gy), gy)
    assert np.allclose(O._clip_backward(0.0 * x, gy), gy)


def test_hole_fft_equals_direct_definition(G):
    table = G["d200_table"].astype(np.float64)
    pos = G["d200_pos"][:16]
    assert np.abs(O.hole_score(pos, table) - O.hole_score_direct(pos, table)).max() < 1e-12
    # frequency-domain identity: s = (1/d) Re sum_f conj(r^) conj(h^) t^
    h, t, r = (table[pos[:, c]] * O.clip_scale(table[pos[:, c]]) for c in (0, 1, 2))
    fh, ft, fr = (np.fft.fft(v, axis=1) for v in (h, t, r))
    s = (np.conj(fr) * np.conj(fh) * ft).sum(axis=1).real / table.shape[1]
    assert np.abs(s - O.hole_score(pos, table)).max() < 1e-12


# ---------------------------------------------------------------- golden regression + C port
@pytest.mark.parametrize("d", [50, 128, 200])
def test_golden_vectors_reproduce(G, d):
    t64 = G[f"d{d}_table"].astype(np.float64)
    pos, neg = G[f"d{d}_pos"], G[f"d{d}_neg"]
    assert np.array_equal(O.evaluate_triples(pos, t64)[:, 0], G[f"d{d}_sigma"])
    for margin in (0.2, 0.0, -0.5):
        new, loss = O.sgd_step(t64, pos, neg, lr=0.05, margin=margin)
        assert np.array_equal(loss, G[f"d{d}_m{margin}_loss"])
        if margin != -0.5:
            assert np.array_equal(new.astype(np.float32), G[f"d{d}_m{margin}_table_after"])
    # all-inactive hinge leaves the table untouched
    new, loss = O.sgd_step(t64, pos, neg, lr=0.05, margin=-0.5)
    assert (loss == 0).all() and np.array_equal(new, t64)


@pytest.mark.parametrize("d", [50, 128, 200])
def test_c_port_matches_numpy_oracle(G, d):
    table = G[f"d{d}_table"]
    pos, neg = G[f"d{d}_pos"], G[f"d{d}_neg"]
    assert np.abs(CO.complex_score(table, pos) - G[f"d{d}_sigma"]).max() < 2e-6
    assert np.abs(CO.complex_score(table, pos, apply_sigmoid=False) - G[f"d{d}_score_raw"]).max() < 2e-6
    assert np.abs(CO.complex_score(table, pos, apply_sigmoid=False, hole=True) - G[f"d{d}_hole_raw"]).max() < 5e-6
    for hole, tag in ((False, ""), (True, "_hole")):
        for margin in (0.2, 0.0):
            t = table.copy()
            loss = CO.hinge_step(t, pos, neg, margin, 0.05, hole=hole, threads=2)
            exp_loss = G[f"d{d}_m{margin}{tag}_loss"]
            # at margin 0 a pair sits exactly on the hinge kink when pos == neg: value 0 either way
            assert np.abs(loss - exp_loss).max() < 2e-6
            exp = G[f"d{d}_m{margin}{tag}_table_after"]
            if margin == 0.0:
                # pairs whose pre-activation is within fp32 rounding of 0 may flip the mask
                pre = np.abs(exp_loss) < 1e-6
                if pre.any():
                    continue
            assert np.abs(t - exp).max() < 5e-6


def test_c_port_bad_ids_give_nan(G):
    table = G["d50_table"]
    tr = np.array([[1, 2, 3], [1, 2, 64], [-1, 2, 3]], dtype=np.int32)
    s = CO.complex_score(table, tr)
    assert np.isfinite(s[0]) and np.isnan(s[1]) and np.isnan(s[2])


# ---------------------------------------------------------------- sampler
def test_sampler_c_port_bit_exact_and_golden(G):
    a = (G["smp_id_to_type"], G["smp_offsets"], G["smp_ids"])
    pos = G["smp_pos"]
    for mode in range(4):
        for (seed, step, padded) in ((0, 0, 1024), (0x1234567890ABCDEF, 77, 16), (5, 2**40 + 3, 0)):
            exp = G[f"smp_neg_m{mode}_s{seed}_t{step}_p{padded}"]
            assert np.array_equal(O.corrupt_batch(pos, *a, seed, step, padded, mode), exp)
            assert np.array_equal(CO.corrupt_batch(pos, *a, seed, step, padded, mode), exp)


def test_sampler_semantics(G):
    id_to_type, offsets, ids = G["smp_id_to_type"], G["smp_offsets"], G["smp_ids"]
    pos = G["smp_pos"]
    sides = set()
    for step in range(40):
        neg = O.corrupt_batch(pos, id_to_type, offsets, ids, seed=3, step=step)
        ch = (neg != pos)
        assert not ch[:, 2].any()                       # relations never corrupted (holE.py:152-158)
        assert not (ch[:, 0].any() and ch[:, 1].any())  # one side per batch (holE.py:137-140)
        col = 0 if batch_heads(3, step) else 1
        sides.add(col)
        known = id_to_type[np.clip(pos[:, col], 0, len(id_to_type) - 1)] >= 0
        assert (neg[~known, col] == -1).all()           # '?' default row of -1s (holE.py:39)
        assert (id_to_type[neg[known, col]] == id_to_type[pos[known, col]]).all()  # type-safe
    assert sides == {0, 1}


def batch_heads(seed, step):
    return O.batch_coin_is_heads(seed, step)


def test_sampler_uniform_within_type():
    # one type of 50 ids: marginal of the two-stage draw (holE.py:343-344 then 108-112) is uniform
    n_ids = 50
    ids = np.arange(3, 3 + n_ids, dtype=np.int32)
    id_to_type = np.concatenate([[-1, -1, -1], np.zeros(n_ids, np.int32)]).astype(np.int32)
    offsets = np.array([0, n_ids], dtype=np.int64)
    pos = np.tile(np.array([[3, 4, 0]], dtype=np.int32), (4000, 1))
    counts = np.zeros(n_ids)
    for step in range(25):
        neg = O.corrupt_batch(pos, id_to_type, offsets, ids, seed=11, step=step, padded_size=1024,
                              mode=O.MODE_HEADS)
        counts += np.bincount(neg[:, 0] - 3, minlength=n_ids)
    exp = counts.sum() / n_ids
    chi2 = ((counts - exp) ** 2 / exp).sum()
    # the with-replacement subsample inflates the variance (rows of one batch share 1024 draws):
    # var ~ exp * (1 + B/padded) => scale the 49-dof chi2 bound accordingly
    assert chi2 < 90 * (1 + 4000 / 1024)
    # within one batch at most padded_size distinct (type, slot) pairs exist
    neg = O.corrupt_batch(pos, id_to_type, offsets, ids, seed=11, step=0, padded_size=8, mode=O.MODE_HEADS)
    assert len(np.unique(neg[:, 0])) <= 8


# ---------------------------------------------------------------- rank / MRR (holE.py:427-490)
def test_eval_link_prediction_toy():
    # head 7, relation 0; candidates tails 10..15 with losses; ties broken by the triple tuple
    triples = np.array([[7, t, 0] for t in range(10, 16)])
    scores = np.array([0.30, 0.10, 0.10, 0.50, 0.05, 0.40])
    true = O.triple_dict([[7, 14, 0]])                  # tail 14 is a known train triple -> filtered
    test = O.triple_dict([[7, 12, 0], [7, 13, 0]])
    raw, fil = [], []
    O.eval_link_prediction(scores, triples, true, test, raw, fil)
    # pop order: 14 (0.05), 11 (0.10), 12 (0.10), 10, 15, 13
    assert raw == [3, 6]
    assert fil == [2, 5]
    m = O.score_mrr(raw, fil)
    assert m["raw_mrr"] == pytest.approx((1 / 3 + 1 / 6) / 2)
    assert m["filtered_mrr"] == pytest.approx((1 / 2 + 1 / 5) / 2)
    assert m["hits1"] == 0 and m["hits3"] == 50 and m["hits10"] == 100


# ---------------------------------------------------------------- f2: logistic-loss mode
def test_logloss_step_matches_torch_autograd(G):
    d = 128
    table = G[f"d{d}_table"].astype(np.float64)
    pos = G[f"d{d}_pos"]
    keep = ~np.isin(pos, [0, 1]).any(axis=1)
    pos = pos[keep]
    rng = np.random.default_rng(0)
    negs = np.stack([pos.copy() for _ in range(3)])
    for kk in range(3):
        negs[kk, :, kk % 2] = rng.integers(8, 64, len(pos))
    negs[(negs == 0) | (negs == 1)] = 9
    lr, l2 = 0.01, 0.003
    new, loss = O.logloss_step(table, pos, negs, lr, l2)
    emb = torch.tensor(table, dtype=torch.float64, requires_grad=True)

    def lookup(ids):
        t = emb[torch.as_tensor(ids, dtype=torch.long)]
        return t * torch.minimum(torch.rsqrt((t * t).sum(1, keepdim=True)), torch.ones(1, dtype=torch.float64))
    tri = np.concatenate([pos] + list(negs), 0)
    y = torch.tensor(np.concatenate([np.ones(len(pos)), -np.ones(3 * len(pos))]))
    h, t, r = lookup(tri[:, 0]), lookup(tri[:, 1]), lookup(tri[:, 2])
    k = d // 2
    hc, tc, rc = (torch.complex(v[:, :k], v[:, k:]) for v in (h, t, r))
    s = (hc * (rc * torch.conj(tc))).real.sum(1)
    lvec = torch.log(1.0 + torch.exp(-y * s)) + l2 * 0.5 * (emb * emb).sum()     # holE.py:195-196
    lvec.sum().backward()
    assert np.abs(loss - lvec.detach().numpy()).max() < 1e-10
    assert np.abs(new - (emb.detach() - lr * emb.grad).numpy()).max() < 1e-12


def test_historical_hole_graph_variant_fft_equals_direct():
    """The 2017-07-24 graph's HolE (complex FFT correlation, Re+Im, tanh; graph.pbtxt:6221-6521) is kept in
    the oracle as a documented compatibility restatement: its FFT form equals the O(k^2) definition."""
    rng = np.random.default_rng(3)
    for d in (50, 128, 200):
        table = rng.standard_normal((40, d)) * rng.uniform(0.02, 0.3, (40, 1))
        tr = rng.integers(0, 40, (64, 3))
        a = O.hole_graph20170724_evaluate(tr, table)
        b = O.hole_graph20170724_evaluate_direct(tr, table)
        assert a.shape == (64, 1) and np.abs(a - b).max() < 1e-12
        assert np.all(np.abs(a) < 1.0)
