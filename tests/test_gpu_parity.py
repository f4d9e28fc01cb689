"""Parity of the HIP path (through the C ABI, via the Python host) against the CPU oracle.

Bars: integer / index work (sampler, gather, IndexedSlices indices) bit-exact; floating point within
the tolerance BASELINE.json's north_star states -- scores within 1e-5 of the fp32 reference
arithmetic (we hold sigma and raw scores to 1e-5 absolute against the fp64 restatement), and the
table after one step within 5e-6 absolute (fp32 atomics change the summation order).
"""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as CO
from oracle import hole_oracle as O

pytestmark = pytest.mark.gpu

SCORE_TOL = 1e-5   # north_star: "scores matching the reference within 1e-5 fp32"
TABLE_TOL = 5e-6


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU: the HIP path cannot be checked")
    from graphembeddings_amd import hole
    return hole


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "golden_v1.npz"))


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


# ---------------------------------------------------------------- evaluate_triples
@pytest.mark.parametrize("d", [50, 128, 200])
def test_score_matches_golden(H, G, d):
    table, pos = dev(G[f"d{d}_table"]), dev(G[f"d{d}_pos"])
    sig = H.evaluate_triples(pos, table).cpu().numpy()
    assert sig.shape == (len(G[f"d{d}_pos"]), 1)
    assert np.abs(sig[:, 0] - G[f"d{d}_sigma"]).max() < SCORE_TOL
    raw = H.evaluate_triples(pos, table, apply_sigmoid=False).cpu().numpy()[:, 0]
    assert np.abs(raw - G[f"d{d}_score_raw"]).max() < SCORE_TOL
    hole_raw = H.evaluate_triples(pos, table, model="hole", apply_sigmoid=False).cpu().numpy()[:, 0]
    assert np.abs(hole_raw - G[f"d{d}_hole_raw"]).max() < SCORE_TOL
    # int64 triples are accepted (inference placeholder, holE.py:547)
    sig64 = H.evaluate_triples(pos.to(torch.int64), table).cpu().numpy()
    assert np.array_equal(sig, sig64)


@pytest.mark.parametrize("B", [0, 1, 3, 65, 1000])
def test_score_ragged_batch_sizes(H, G, B):
    table = G["d200_table"]
    rng = np.random.default_rng(B)
    tr = rng.integers(0, table.shape[0], size=(B, 3)).astype(np.int32)
    got = H.evaluate_triples(dev(tr), dev(table)).cpu().numpy()
    assert got.shape == (B, 1)
    if B:
        assert np.abs(got[:, 0] - O.evaluate_triples(tr, table.astype(np.float64))[:, 0]).max() < SCORE_TOL


@pytest.mark.parametrize("d", [2, 8, 30, 64, 100, 256, 400, 512, 1024])
def test_score_other_dims(H, d):
    rng = np.random.default_rng(d)
    table = (rng.standard_normal((40, d)) * rng.uniform(0.1, 1.5, (40, 1)) / np.sqrt(d)).astype(np.float32)
    tr = rng.integers(0, 40, size=(77, 3)).astype(np.int32)
    got = H.evaluate_triples(dev(tr), dev(table)).cpu().numpy()[:, 0]
    assert np.abs(got - O.evaluate_triples(tr, table.astype(np.float64))[:, 0]).max() < SCORE_TOL
    if d <= 512:
        goth = H.evaluate_triples(dev(tr), dev(table), model="hole").cpu().numpy()[:, 0]
        assert np.abs(goth - O.hole_evaluate_triples(tr, table.astype(np.float64))[:, 0]).max() < SCORE_TOL


def test_unsupported_dims_fail_loudly(H):
    from graphembeddings_amd._lib import GeError
    table = torch.zeros(4, 4096, device="cuda")
    tr = torch.zeros(2, 3, dtype=torch.int32, device="cuda")
    with pytest.raises(GeError):
        H.evaluate_triples(tr, table)
    with pytest.raises(GeError):
        H.evaluate_triples(tr, torch.zeros(4, 7, device="cuda"))  # odd d has no complex split


def test_bad_ids_give_nan_and_touch_nothing(H, G):
    table = dev(G["d50_table"])
    tr = dev(np.array([[1, 2, 3], [1, 2, 64], [-1, 2, 3]], dtype=np.int32))
    s = H.evaluate_triples(tr, table).cpu().numpy()[:, 0]
    assert np.isfinite(s[0]) and np.isnan(s[1]) and np.isnan(s[2])
    before = table.clone()
    opt = H.HingeSGD(table, 3, margin=0.2)
    loss = opt.step(tr, tr.clone(), 0.1).cpu().numpy()[:, 0]
    assert np.isnan(loss[1]) and np.isnan(loss[2]) and np.isfinite(loss[0])
    torch.cuda.synchronize()
    # pair 0 has pos == neg: gradients cancel exactly; pairs 1,2 are skipped
    assert np.abs((table - before).cpu().numpy()).max() < 1e-7


# ---------------------------------------------------------------- hinge forward
@pytest.mark.parametrize("model", ["complex", "hole"])
@pytest.mark.parametrize("d", [50, 128, 200])
def test_hinge_loss_forward(H, G, d, model):
    table, pos, neg = dev(G[f"d{d}_table"]), dev(G[f"d{d}_pos"]), dev(G[f"d{d}_neg"])
    tag = "" if model == "complex" else "_hole"
    for margin in (0.2, 0.0, -0.5):
        loss, sp, sn = H.hinge_loss(pos, neg, table, margin=margin, model=model, return_sigmoids=True)
        assert np.abs(loss.cpu().numpy()[:, 0] - G[f"d{d}_m{margin}{tag}_loss"]).max() < SCORE_TOL
    if model == "complex":
        assert np.abs(sp.cpu().numpy()[:, 0] - G[f"d{d}_sigma"]).max() < SCORE_TOL


# ---------------------------------------------------------------- one SGD step
@pytest.mark.parametrize("model", ["complex", "hole"])
@pytest.mark.parametrize("d", [50, 128, 200])
@pytest.mark.parametrize("margin", [0.2, 0.0])
def test_hinge_step_matches_golden(H, G, d, model, margin):
    tag = "" if model == "complex" else "_hole"
    table = dev(G[f"d{d}_table"]).clone()
    pos, neg = dev(G[f"d{d}_pos"]), dev(G[f"d{d}_neg"])
    opt = H.HingeSGD(table, len(G[f"d{d}_pos"]), margin=margin, model=model)
    loss = opt.step(pos, neg, 0.05).cpu().numpy()[:, 0]
    exp_loss = G[f"d{d}_m{margin}{tag}_loss"]
    assert np.abs(loss - exp_loss).max() < SCORE_TOL
    got = table.cpu().numpy()
    exp = G[f"d{d}_m{margin}{tag}_table_after"]
    # a pair whose fp64 pre-activation is within fp32 rounding of the kink may flip its mask: the
    # rows it touches are excluded (none for margin 0.2)
    pre64 = _pre_activation(G, d, model, margin)
    risky = np.abs(pre64) < 1e-6
    if risky.any():
        rows = np.unique(np.concatenate([G[f"d{d}_pos"][risky].ravel(), G[f"d{d}_neg"][risky].ravel()]))
        mask = np.ones(got.shape[0], bool); mask[rows] = False
        got, exp = got[mask], exp[mask]
        assert margin == 0.0
    assert np.abs(got - exp).max() < TABLE_TOL


def _pre_activation(G, d, model, margin):
    t64 = G[f"d{d}_table"].astype(np.float64)
    f = O.complex_score if model == "complex" else O.hole_score
    return O.sigmoid(f(G[f"d{d}_pos"], t64)) - O.sigmoid(f(G[f"d{d}_neg"], t64)) + margin


@pytest.mark.parametrize("model", ["complex", "hole"])
def test_indexed_slices_match_closed_form(H, G, model):
    d = 200
    table64 = G[f"d{d}_table"].astype(np.float64)
    pos, neg = G[f"d{d}_pos"], G[f"d{d}_neg"]
    loss, gi, gv = H.hinge_grad(dev(G[f"d{d}_table"]), dev(pos), dev(neg), 0.05, margin=0.2, model=model)
    gi, gv = gi.cpu().numpy(), gv.cpu().numpy()
    B = len(pos)
    assert gi.shape == (6 * B,) and gv.shape == (6 * B, d)
    # slot order h+,t+,r+,h-,t-,r-; shared rows merged into the positive slot, the other slot is -1
    # pairs whose hinge is inactive (MaximumGrad mask 0) emit no slices at all
    gi2 = gi.reshape(B, 6)
    on = _pre_activation(G, d, model, 0.2) >= 0
    assert on.any() and (model == "hole" or not on.all())
    for X in range(3):
        same = pos[:, X] == neg[:, X]
        assert np.array_equal(gi2[:, X], np.where(on, pos[:, X], -1))
        assert np.array_equal(gi2[:, 3 + X], np.where(on & ~same, neg[:, X], -1))
    acc = np.zeros_like(table64)
    live = gi >= 0
    np.add.at(acc, gi[live], gv[live].astype(np.float64))
    idx, val, _ = O.hinge_grads(pos, neg, table64, margin=0.2, model=model)
    exp = np.zeros_like(table64)
    np.add.at(exp, idx, -0.05 * val)
    assert np.abs(acc - exp).max() < TABLE_TOL


def test_all_inactive_margin_leaves_table_bit_identical(H, G):
    table = dev(G["d200_table"]).clone()
    before = table.clone()
    opt = H.HingeSGD(table, 64, margin=-0.5)
    loss = opt.step(dev(G["d200_pos"]), dev(G["d200_neg"]), 0.05)
    torch.cuda.synchronize()
    assert (loss == 0).all() and torch.equal(table, before)


# ---------------------------------------------------------------- config-2 size against the C port
@pytest.mark.parametrize("model,d,B", [("complex", 200, 4096), ("complex", 50, 128), ("hole", 200, 1024),
                                       ("hole", 200, 4096),      # BASELINE config 3 at its headline batch
                                       # HolE beyond one 256-lag chunk: 16-byte path (300) and the generic path (258, 67)
                                       ("hole", 300, 96), ("hole", 258, 64), ("hole", 67, 64), ("hole", 512, 32),
                                       ("complex", 1024, 64), ("complex", 6, 64)])
def test_step_at_baseline_sizes_against_c_port(H, model, d, B):
    from graphembeddings_amd import data as D
    fb = D.fb15k_shape()
    names, id_to_type, offsets, ids = fb.type_arrays()
    rng = np.random.default_rng(7)
    table = O.init_table(fb.entity_count, d, seed=3)
    # scale a third of the rows past the unit ball so the clip and its Jacobian are exercised
    big = rng.random(fb.entity_count) < 0.33
    table[big] *= (rng.uniform(1.1, 2.0, big.sum()) / np.linalg.norm(table[big], axis=1))[:, None]
    pos = D.synthetic_fb15k_triples(fb, n_triples=B, seed=5)
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
    neg_t = H.corrupt_batch(tt, fb.relation_count, dev(pos), seed=9, step=4)
    neg = neg_t.cpu().numpy()
    assert np.array_equal(neg, CO.corrupt_batch(pos, id_to_type, offsets, ids, 9, 4, 1024, 0))  # bit-exact
    gtab = dev(table).clone()
    opt = H.HingeSGD(gtab, B, margin=0.2, model=model)
    loss = opt.step(dev(pos), neg_t, 0.1).cpu().numpy()[:, 0]
    ctab = table.copy()
    closs = CO.hinge_step(ctab, pos, neg, 0.2, 0.1, hole=(model == "hole"), threads=8)
    assert np.abs(loss - closs).max() < SCORE_TOL
    assert np.abs(gtab.cpu().numpy() - ctab).max() < 2e-5   # two fp32 implementations, 100s of adds on hot rows
    # scores of the updated table agree too
    s = H.evaluate_triples(dev(pos), gtab, model=model).cpu().numpy()[:, 0]
    assert np.abs(s - CO.complex_score(ctab, pos, hole=(model == "hole"), threads=8)).max() < SCORE_TOL


def test_step_matches_torch_eager_fp32_on_device(H):
    """An independent fp32 implementation on the same GPU: the TF op chain of holE.py:161-234, 296 written with
    stock PyTorch-ROCm ops and autograd (no oracle, no library code) gives the same loss and table."""
    from graphembeddings_amd import data as D
    fb = D.fb15k_shape()
    d, B = 200, 2048
    names, id_to_type, offsets, ids = fb.type_arrays()
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
    pos = dev(D.synthetic_fb15k_triples(fb, n_triples=B, seed=11))
    neg = H.corrupt_batch(tt, fb.relation_count, pos, seed=2, step=1)
    rng = np.random.default_rng(2)
    table = O.init_table(fb.entity_count, d, seed=4)
    big = rng.random(fb.entity_count) < 0.3                      # rows beyond the unit ball: the clip Jacobian matters
    table[big] *= (rng.uniform(1.1, 2.0, big.sum()) / np.linalg.norm(table[big], axis=1))[:, None]
    ours, ref = dev(table).clone(), dev(table).clone()
    loss = H.HingeSGD(ours, B, margin=0.2).step(pos, neg, 0.1)[:, 0]
    idx = torch.cat([pos, neg], 0).long()
    rows = [ref[idx[:, c]].detach().requires_grad_(True) for c in range(3)]
    k = d // 2
    y = [r * torch.clamp(torch.rsqrt((r * r).sum(1, keepdim=True)), max=1.0) for r in rows]
    h, t, r = [torch.complex(v[:, :k], v[:, k:]) for v in y]
    e = torch.sigmoid((h * r * torch.conj(t)).real.sum(1))
    ref_loss = torch.clamp(e[:B] - e[B:] + 0.2, min=0.0)
    ref_loss.sum().backward()
    with torch.no_grad():
        for c in range(3):
            ref.index_add_(0, idx[:, c], rows[c].grad, alpha=-0.1)
    assert (loss - ref_loss.detach()).abs().max().item() < SCORE_TOL
    assert (ours - ref).abs().max().item() < TABLE_TOL


def test_hole_step_matches_torch_fft_autograd_on_device(H):
    """HolE: README.md:42's FFT formula (ifft(conj(fft(h)) fft(t)), score r.(h star t)) evaluated with torch.fft
    (rocFFT) and differentiated by autograd on the device == the direct-correlation kernels, loss and table."""
    rng = np.random.default_rng(8)
    N, d, B = 3000, 200, 1024
    table = (rng.standard_normal((N, d)) * rng.uniform(0.02, 0.12, (N, 1))).astype(np.float32)   # norms 0.3 .. 1.7
    pos = np.stack([rng.integers(20, N, B), rng.integers(20, N, B), rng.integers(0, 20, B)], 1).astype(np.int32)
    neg = pos.copy()
    neg[: B // 2, 0] = rng.integers(20, N, B // 2)             # head-corrupted half, tail-corrupted half
    neg[B // 2:, 1] = rng.integers(20, N, B - B // 2)
    ours, ref = dev(table).clone(), dev(table).clone()
    loss = H.HingeSGD(ours, B, margin=0.2, model="hole").step(dev(pos), dev(neg), 0.1)[:, 0]
    idx = torch.cat([dev(pos), dev(neg)], 0).long()
    rows = [ref[idx[:, c]].detach().requires_grad_(True) for c in range(3)]
    h, t, r = [x * torch.clamp(torch.rsqrt((x * x).sum(1, keepdim=True)), max=1.0) for x in rows]
    corr = torch.fft.irfft(torch.conj(torch.fft.rfft(h, dim=1)) * torch.fft.rfft(t, dim=1), n=d, dim=1)
    e = torch.sigmoid((r * corr).sum(1))
    ref_loss = torch.clamp(e[:B] - e[B:] + 0.2, min=0.0)
    ref_loss.sum().backward()
    with torch.no_grad():
        for c in range(3):
            ref.index_add_(0, idx[:, c], rows[c].grad, alpha=-0.1)
    assert (loss - ref_loss.detach()).abs().max().item() < SCORE_TOL
    assert (ours - ref).abs().max().item() < 2e-5


def test_step_is_linear_in_duplicated_pairs(H, G):
    # ScatterSub applies every occurrence: a batch holding each pair twice moves the table twice as far
    t0 = dev(G["d200_table"])
    pos, neg = dev(G["d200_pos"]), dev(G["d200_neg"])
    a = t0.clone(); H.HingeSGD(a, 64).step(pos, neg, 0.01)
    b = t0.clone(); H.HingeSGD(b, 128).step(torch.cat([pos, pos]), torch.cat([neg, neg]), 0.01)
    torch.cuda.synchronize()
    assert torch.allclose((b - t0), 2 * (a - t0), atol=2e-6, rtol=0)


def test_scores_are_permutation_equivariant_at_full_size(H):
    N, d, B = 16296, 200, 1 << 20
    g = torch.Generator(device="cpu").manual_seed(0)
    table = (torch.randn(N, d, generator=g) * 0.1).cuda()
    tr = torch.randint(0, N, (B, 3), generator=g, dtype=torch.int32).cuda()
    s = H.evaluate_triples(tr, table)
    perm = torch.randperm(B, generator=g).cuda()
    s2 = H.evaluate_triples(tr[perm].contiguous(), table)
    assert torch.equal(s[perm], s2)   # bitwise: each triple is reduced in a fixed lane order
    assert torch.isfinite(s).all() and float(s.min()) > 0 and float(s.max()) < 1


# ---------------------------------------------------------------- sampler
def test_corrupt_batch_bit_exact(H, G):
    tt = None
    pos = dev(G["smp_pos"])
    for (seed, step, padded) in ((0, 0, 1024), (0x1234567890ABCDEF, 77, 16), (5, 2**40 + 3, 0)):
        tt = H.TypeTables.from_host(G["smp_id_to_type"], G["smp_offsets"], G["smp_ids"], padded_size=padded)
        for mode in range(4):
            neg = H.corrupt_batch(tt, 8, pos, seed=seed, step=step, mode=mode).cpu().numpy()
            assert np.array_equal(neg, G[f"smp_neg_m{mode}_s{seed}_t{step}_p{padded}"])
    # evaluate_batch = corrupt + hinge, deterministic in (seed, step)
    table = dev(np.random.default_rng(0).standard_normal((len(G["smp_id_to_type"]), 64)).astype(np.float32) * 0.1)
    l1 = H.evaluate_batch(pos, table, tt, None, 8, seed=1, step=2)
    l2 = H.evaluate_batch(pos, table, tt, None, 8, seed=1, step=2)
    assert l1.shape == (len(G["smp_pos"]), 1) and torch.allclose(l1, l2, rtol=0, atol=0, equal_nan=True)
    assert torch.isnan(l1).sum().item() == 1   # the row whose corrupted-side id has an unknown type (-1)


# ---------------------------------------------------------------- rows
def test_gather_and_scatter_rows(H):
    rng = np.random.default_rng(1)
    for d in (200, 50, 7):
        table = rng.standard_normal((100, d)).astype(np.float32)
        idx = rng.integers(-1, 100, size=333).astype(np.int32)
        idx[:5] = [3, 3, 3, -1, 99]
        got = H.gather_rows(dev(table), dev(idx)).cpu().numpy()
        exp = np.where((idx >= 0)[:, None], table[np.clip(idx, 0, 99)], 0.0)
        assert np.array_equal(got, exp)
        val = rng.standard_normal((333, d)).astype(np.float32)
        t = dev(table).clone()
        H.scatter_add_rows(t, dev(idx), dev(val))
        exp = table.astype(np.float64)
        np.add.at(exp, idx[idx >= 0], val[idx >= 0].astype(np.float64))
        assert np.abs(t.cpu().numpy() - exp).max() < 1e-5


# ---------------------------------------------------------------- 1-vs-K (MFMA GEMM)
@pytest.mark.parametrize("d,B,K", [(200, 37, 129), (200, 256, 256), (50, 5, 70), (128, 130, 64),
                                   # every instantiation of the split-precision tile kernel (2 ... 7 column groups a lane,
                                   # ragged last group) and the first embedding_dim beyond it (fp32 tile kernel)
                                   (56, 70, 100), (64, 70, 100), (88, 70, 100), (152, 70, 100), (184, 130, 64), (224, 65, 64), (232, 65, 64),
                                   # fp32 tile kernel where its 64 x 65 output tile is LARGER than its operands (d < 40:
                                   # the clip scales must sit behind the tile, not behind the operands), rows 63 / cols 63 used
                                   (8, 128, 128), (16, 70, 100), (24, 128, 64), (32, 64, 128), (40, 128, 128)])
@pytest.mark.parametrize("cand_is_head", [False, True])
def test_score_candidates_matches_per_triple_scores(H, d, B, K, cand_is_head):
    rng = np.random.default_rng(d + B)
    N = 300
    table = (rng.standard_normal((N, d)) * rng.uniform(0.3, 2.0, (N, 1)) / np.sqrt(d)).astype(np.float32)
    hr = np.stack([rng.integers(10, N, B), rng.integers(0, 10, B)], axis=1).astype(np.int32)
    cand = rng.permutation(np.arange(10, N))[:K].astype(np.int32)
    got = H.score_candidates(dev(table), dev(hr), dev(cand), cand_is_head=cand_is_head).cpu().numpy()
    assert got.shape == (B, K)
    fixed = np.repeat(hr[:, 0], K); rel = np.repeat(hr[:, 1], K); c = np.tile(cand, B)
    tr = np.stack([c, fixed, rel], 1) if cand_is_head else np.stack([fixed, c, rel], 1)
    exp = O.evaluate_triples(tr, table.astype(np.float64))[:, 0].reshape(B, K)
    assert np.abs(got - exp).max() < SCORE_TOL
    # and it equals the per-triple kernel
    per = H.evaluate_triples(dev(tr.astype(np.int32)), dev(table)).cpu().numpy().reshape(B, K)
    assert np.abs(got - per).max() < SCORE_TOL


def test_score_candidates_full_fb15k_shape(H):
    # BASELINE config 5 at full size: B=4096 positives x 256 shared negatives, d=200, through the fp32-MFMA GEMM;
    # 2,000 sampled cells against the fp64 ORACLE (not against another kernel of this library)
    N, d, B, K = 16296, 200, 4096, 256
    g = torch.Generator(device="cpu").manual_seed(1)
    table = (torch.randn(N, d, generator=g) * 0.12)
    table[::6] *= 3.0                                                   # rows outside the unit ball
    hr = torch.stack([torch.randint(1345, N, (B,), generator=g), torch.randint(0, 1345, (B,), generator=g)], 1).int()
    cand = torch.randint(1345, N, (K,), generator=g).int()
    t64 = table.numpy().astype(np.float64)
    rows = torch.randint(0, B, (2000,), generator=g)
    cols = torch.randint(0, K, (2000,), generator=g)
    for cand_is_head in (False, True):
        out = H.score_candidates(table.cuda(), hr.cuda(), cand.cuda(), cand_is_head=cand_is_head).cpu()
        fixed, c, rel = hr[rows, 0].numpy(), cand[cols].numpy(), hr[rows, 1].numpy()
        tr = np.stack([c, fixed, rel], 1) if cand_is_head else np.stack([fixed, c, rel], 1)
        ref = O.evaluate_triples(tr, t64)[:, 0]
        assert np.abs(out[rows, cols].numpy() - ref).max() < SCORE_TOL


@pytest.mark.parametrize("d,B,K", [(104, 2100, 4200), (200, 8200, 1030), (56, 2100, 4200)])
def test_score_candidates_large_sweeps_run_on_the_candidate_planes(H, d, B, K):
    """ge_complex_score_1vK above 512 tiles of 128 x 128 takes the split-precision sweep in its scores-only mode (planes
    built inside the call): ragged B and K, a k block that ends inside the row (d = 104, 56), raw scores and losses,
    tails and heads -- 3,000 sampled cells and the whole last row / last column against the fp64 oracle."""
    N = 6000
    g = torch.Generator(device="cpu").manual_seed(d)
    table = (torch.randn(N, d, generator=g) * 0.15)
    table[::5] *= 4.0                                                   # rows outside the unit ball
    hr = torch.stack([torch.randint(50, N, (B,), generator=g), torch.randint(0, 50, (B,), generator=g)], 1).int()
    cand = torch.randint(50, N, (K,), generator=g).int()
    t64 = table.numpy().astype(np.float64)
    rows = torch.cat([torch.randint(0, B, (3000,), generator=g), torch.full((K,), B - 1), torch.arange(B)])
    cols = torch.cat([torch.randint(0, K, (3000,), generator=g), torch.arange(K), torch.full((B,), K - 1)])
    for cand_is_head in (False, True):
        fixed, c, rel = hr[rows, 0].numpy(), cand[cols].numpy(), hr[rows, 1].numpy()
        tr = np.stack([c, fixed, rel], 1) if cand_is_head else np.stack([fixed, c, rel], 1)
        out = H.score_candidates(table.cuda(), hr.cuda(), cand.cuda(), cand_is_head=cand_is_head).cpu()
        assert np.abs(out[rows, cols].numpy() - O.evaluate_triples(tr, t64)[:, 0]).max() < SCORE_TOL
        raw = H.score_candidates(table.cuda(), hr.cuda(), cand.cuda(), cand_is_head=cand_is_head, apply_sigmoid=False).cpu()
        assert np.abs(raw[rows, cols].numpy() - np.asarray(O.complex_score(tr, t64)).reshape(-1)).max() < SCORE_TOL


# ---------------------------------------------------------------- row-sharded path, HIP kernels
@pytest.mark.parametrize("shape", ["fb15k", "config4"])
def test_sharded_trainer_single_rank_uses_hip_kernels(H, shape):
    """world_size 1 on the GPU: the exchange degenerates to local copies, the four kernels are the
    real HIP ones.  Result must equal the plain fused step (same negatives) and the C port.  config4 =
    BASELINE config 4's workload: 1,200,018 x 200 table (960 MB), Zipf(0.8) ids, B = 16,384."""
    from graphembeddings_amd import data as D
    from graphembeddings_amd import sharded as S
    if shape == "fb15k":
        fb = D.fb15k_shape()
        names, id_to_type, offsets, ids = fb.type_arrays()
        table_h = O.init_table(fb.entity_count, 200, seed=2) * 6.0
        B = 2048
        pos_h = D.synthetic_fb15k_triples(fb, n_triples=B, seed=3)
        n_rows, n_rel = fb.entity_count, fb.relation_count
    else:
        B = 16384
        data, pos_h = D.synthetic_large(n_entities=1_200_000, n_triples=B, seed=1234)
        names, id_to_type, offsets, ids = D.synthetic_large_type_arrays(data)
        table_h = np.random.default_rng(5).standard_normal((data.entity_count, 200), dtype=np.float32) * np.float32(0.02)
        table_h[::7] *= np.float32(5.0)
        n_rows, n_rel = data.entity_count, data.relation_count
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
    table = dev(table_h)
    pos = dev(pos_h)
    a = table.clone()
    tr = S.ShardedTrainer(a, n_rows, tt, seed=4)
    loss_s = tr.step(pos, lr=0.1)
    neg = H.corrupt_batch(tt, n_rel, pos, seed=4, step=0)
    b = table.clone()
    loss_p = H.HingeSGD(b, B).step(pos, neg, 0.1)[:, 0]
    torch.cuda.synchronize()
    assert (loss_s - loss_p).abs().max().item() < SCORE_TOL
    assert (a - b).abs().max().item() < TABLE_TOL
    assert tr.stats.unique_rows > 0 and tr.stats.remote_rows == 0
    ctab = table_h.copy()
    closs = CO.hinge_step(ctab, pos_h, neg.cpu().numpy(), 0.2, 0.1, threads=16)
    assert np.abs(loss_s.cpu().numpy() - closs).max() < SCORE_TOL
    assert np.abs(a.cpu().numpy() - ctab).max() < 2e-5


@pytest.mark.parametrize("model", ["hole", "hole_spectral"])
def test_sharded_trainer_hole_models_equal_the_plain_step(H, model):
    """The row-sharded step with the HolE score at world size 1: model="hole" (a real table; the trainer carries its
    shard in the frequency domain and gather_full_table() hands real rows back) and "hole_spectral" (a table that
    already is spectral) equal the plain fused step of the same model on the same negatives."""
    from graphembeddings_amd import data as D
    from graphembeddings_amd import sharded as S
    fb = D.fb15k_shape()
    names, id_to_type, offsets, ids = fb.type_arrays()
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
    d, B = 64, 1024
    table = dev(O.init_table(fb.entity_count, d, seed=6) * 6.0)
    if model == "hole_spectral":
        H.hole_to_spectral(table)
    pos = dev(D.synthetic_fb15k_triples(fb, n_triples=B, seed=8))
    tr = S.ShardedTrainer(table.clone(), fb.entity_count, tt, seed=4, model=model)
    loss_s = tr.step(pos, lr=0.1)
    a = tr.gather_full_table()
    neg = H.corrupt_batch(tt, fb.relation_count, pos, seed=4, step=0)
    b = table.clone()
    loss_p = H.HingeSGD(b, B, model=model).step(pos, neg, 0.1)[:, 0]
    torch.cuda.synchronize()
    assert (loss_s - loss_p).abs().max().item() < SCORE_TOL
    assert (a - b).abs().max().item() < 2e-5 * (d if model == "hole_spectral" else 1)   # spectral entries are d x larger


def test_rank_sweep_model_argument(H):
    """ge_rank_1vK: a real-valued HolE table is GE_ENOTSUP (transform it first), an unknown model GE_EINVAL."""
    from graphembeddings_amd import _lib
    emb = torch.zeros(64, 40, device="cuda")
    hr = torch.zeros(4, 2, dtype=torch.int32, device="cuda")
    tid = torch.ones(4, dtype=torch.int32, device="cuda")
    cand = torch.arange(8, dtype=torch.int32, device="cuda")
    nb = torch.zeros(4, dtype=torch.int32, device="cuda")
    nk = torch.zeros(4, dtype=torch.int32, device="cuda")
    for model, code in ((1, _lib.GE_ENOTSUP), (3, _lib.GE_ENOTSUP), (7, _lib.GE_EINVAL)):
        with pytest.raises(_lib.GeError) as e:
            _lib.call("ge_rank_1vK", emb.data_ptr(), 64, 40, hr.data_ptr(), 4, tid.data_ptr(), cand.data_ptr(), 8, 1.0,
                      model, 0, None, None, nb.data_ptr(), nk.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream)
        assert e.value.code == code
    with pytest.raises(ValueError):
        H.rank_candidates(emb, hr, tid, cand, model="hole")


@pytest.mark.parametrize("shape,world,rank", [("fb15k", 1, 0), ("fb15k", 2, 1), ("hot", 3, 0), ("config4", 8, 5), ("config4", 1, 0)])
def test_shard_plan_equals_numpy_model(H, shape, world, rank):
    """ge_shard_plan (csrc/ge_shard.hip: virtual-row keys, multi-workgroup sort, work items, staging order) against the
    NumPy model of the same plan, word for word, for every step of a chunk: the record's items and slot lists (own
    rows, then staged rows R + u), the direct tags (-2 own, -3 - u staged), where every pair reads its rows, the
    request list and the per-owner counts; with invalid ids, a pair whose negative equals its positive, rows split
    over several items ('hot': 5000 slots on one row) and runs that cross tile boundaries (config4: B = 16,384)."""
    from graphembeddings_amd import data as D
    import prep_model as PM
    if shape == "config4":
        S_, B = 2, 16384
        data, pos_h = D.synthetic_large(n_entities=1_200_000, n_triples=S_ * B, seed=77)
        names, id_to_type, offsets, ids = D.synthetic_large_type_arrays(data)
        n_rows, n_rel = data.entity_count, data.relation_count
    else:
        S_, B = 3, 2048 if shape == "fb15k" else 6000
        fb = D.fb15k_shape()
        names, id_to_type, offsets, ids = fb.type_arrays()
        pos_h = D.synthetic_fb15k_triples(fb, n_triples=S_ * B, seed=5)
        n_rows, n_rel = fb.entity_count, fb.relation_count
        if shape == "hot":
            pos_h[:5000, 0] = fb.relation_count + 17
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
    pos = dev(pos_h).view(S_, B, 3).clone()
    neg = torch.stack([H.corrupt_batch(tt, n_rel, pos[s], seed=4, step=s) for s in range(S_)], 0)
    pos[1, 5, 0] = -3                                   # invalid ids: the pair has no slots
    pos[0, 9, 1] = n_rows + 11
    neg[0, 3] = pos[0, 3]                               # negative == positive: no row of its own
    records, pos_src, neg_src, req_row, counts = H.shard_plan(pos, neg, n_rows, world, rank)
    torch.cuda.synchronize()
    lay = H.prepared_layout(B)
    for s in range(S_):
        exp = PM.expected_shard_plan(pos[s].cpu().numpy(), neg[s].cpu().numpy(), n_rows, world, rank, lay)
        got = PM.parse_record(records[s].cpu().numpy(), B, lay)
        assert np.array_equal(counts[s].cpu().numpy(), exp["counts"]), s
        U = len(exp["req_row"])
        assert np.array_equal(req_row[s, :U].cpu().numpy(), exp["req_row"]), s
        assert np.array_equal(pos_src[s].cpu().numpy(), exp["pos_src"]), s
        assert np.array_equal(neg_src[s].cpu().numpy(), exp["neg_src"]), s
        assert np.array_equal(got["slot_item"], exp["slot_item"]), s
        assert got["n_items"] == exp["n_items"], (s, got["n_items"], exp["n_items"])
        for t in range(lay[1]):
            assert np.array_equal(got["items"][t], exp["items"][t]), (s, t)
            assert np.array_equal(got["islots"][t], exp["islots"][t]), (s, t)
    assert shape != "hot" or (got["items"][0][:, 1] >> 30).any()


def test_validation_tick_matches_oracle_and_keeps_the_best_table(H):
    """ge_validation_tick (hole.ValidationPocket): the batch it draws (Philox, restated here), its negatives and mean
    hinge equal the oracle's on the same rows; the pocket takes the table exactly when the mean improves."""
    from graphembeddings_amd import data as D
    fb = D.fb15k_shape()
    names, id_to_type, offsets, ids = fb.type_arrays()
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
    table_h = O.init_table(fb.entity_count, 64, seed=3) * 8.0
    emb = dev(table_h)
    valid_h = fb.validation_triples[:5000].astype(np.int32)
    B, seed = 512, 0xABCDEF0123
    vp = H.ValidationPocket(emb, dev(valid_h), tt, B, margin=0.2, seed=seed)

    def expect(counter, tab):
        i = np.arange(B, dtype=np.uint64)
        lo, hi = i & np.uint64(0xFFFFFFFF), i >> np.uint64(32)
        k0, k1 = (seed & 0xFFFFFFFF) ^ 0x7673656C, (seed >> 32) & 0xFFFFFFFF
        w0 = O.philox4x32_10(counter & 0xFFFFFFFF, counter >> 32, lo, hi, k0, k1)[0].astype(np.uint64)
        w1 = O.philox4x32_10(counter & 0xFFFFFFFF, counter >> 32, lo, hi ^ np.uint64(0x80000000), k0, k1)[0].astype(np.uint64)
        rows = ((w1 << np.uint64(32)) | w0) % np.uint64(len(valid_h))
        pos = valid_h[rows.astype(np.int64)]
        neg = O.corrupt_batch(pos, id_to_type, offsets, ids, seed, counter, 1024, 0)
        return float(O.evaluate_batch(pos, neg, tab.astype(np.float64), 0.2).mean())

    vp.tick(7, 100)
    first = emb.clone()
    emb.mul_(0.5)                                     # another table: its mean hinge differs
    vp.tick(8, 107)
    second = emb.clone()
    emb.zero_()                                       # all scores 0: mean hinge = margin, worse than a trained-ish table? checked below
    vp.tick(9, 114)
    got = vp.read()
    exp = [expect(7, table_h), expect(8, table_h * np.float32(0.5)), expect(9, np.zeros_like(table_h))]
    assert [g[0] for g in got] == [100, 107, 114]
    assert np.abs(np.array([g[1] for g in got]) - np.array(exp)).max() < 1e-5
    best = int(np.argmin([g[1] for g in got]))
    want = [first, second, emb][best]
    assert torch.equal(vp.pocket, want) and abs(float(vp.best) - got[best][1]) == 0.0
    assert vp.read() == []


def test_logloss_validation_tick_matches_oracle(H):
    """ge_validation_tick_logloss (hole.ValidationPocket(log_loss=(K, l2))): the --log_loss objective of holE.py:194-196,
    206-220 on a validation batch -- the batch drawn like the hinge tick's, K corrupted batches with Philox step keys
    counter * K + k, every loss plus l2 * l2_loss(table) -- against the oracle's logloss_values on the same rows; the
    pocket takes the table when the mean improves."""
    from graphembeddings_amd import data as D
    fb = D.fb15k_shape()
    names, id_to_type, offsets, ids = fb.type_arrays()
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
    table_h = O.init_table(fb.entity_count, 64, seed=3) * 8.0
    emb = dev(table_h)
    valid_h = fb.validation_triples[:5000].astype(np.int32)
    B, K, l2, seed = 384, 3, 1e-4, 0x1234567
    vp = H.ValidationPocket(emb, dev(valid_h), tt, B, seed=seed, log_loss=(K, l2))

    def expect(counter, tab):
        i = np.arange(B, dtype=np.uint64)
        lo, hi = i & np.uint64(0xFFFFFFFF), i >> np.uint64(32)
        k0, k1 = (seed & 0xFFFFFFFF) ^ 0x7673656C, (seed >> 32) & 0xFFFFFFFF
        w0 = O.philox4x32_10(counter & 0xFFFFFFFF, counter >> 32, lo, hi, k0, k1)[0].astype(np.uint64)
        w1 = O.philox4x32_10(counter & 0xFFFFFFFF, counter >> 32, lo, hi ^ np.uint64(0x80000000), k0, k1)[0].astype(np.uint64)
        pos = valid_h[(((w1 << np.uint64(32)) | w0) % np.uint64(len(valid_h))).astype(np.int64)]
        negs = [O.corrupt_batch(pos, id_to_type, offsets, ids, seed, counter * K + k, 1024, 0) for k in range(K)]
        tri = np.concatenate([pos] + negs, 0)
        labels = np.concatenate([np.ones(B), -np.ones(K * B)])
        return float(O.logloss_values(tri, labels, tab.astype(np.float64), l2).mean())

    vp.tick(5, 50)
    first = emb.clone()
    emb.mul_(0.25)
    vp.tick(6, 57)
    got = vp.read()
    exp = [expect(5, table_h), expect(6, table_h * np.float32(0.25))]
    assert [g[0] for g in got] == [50, 57]
    assert np.abs(np.array([g[1] for g in got]) / np.array(exp) - 1).max() < 2e-6
    best = int(np.argmin([g[1] for g in got]))
    assert torch.equal(vp.pocket, [first, emb][best])


# ---------------------------------------------------------------- native training loop (ge_train_steps)
@pytest.mark.parametrize("model,B,d,steps", [("complex", 1024, 200, 70), ("complex", 4096, 200, 6),
                                             ("complex", 100, 50, 9), ("hole", 256, 64, 5), ("hole_direct", 256, 64, 5),
                                             ("hole", 128, 67, 4),          # odd d: no packed half spectrum, direct kernels
                                             ("hole", 4096, 200, 12)])      # BASELINE config 3 shape (spectral steps)
def test_train_steps_match_c_port_step_by_step(H, model, B, d, steps):
    """ge_train_steps (prepared path: bulk negatives + LDS-sorted slot index + one RMW per distinct
    row) against the C port replaying the same loop: same batches (incl. the wrap that never yields a
    short batch), same Philox negatives, same LR schedule."""
    from graphembeddings_amd import data as D
    fb = D.fb15k_shape()
    names, id_to_type, offsets, ids = fb.type_arrays()
    T = 5 * B + 77                                   # forces wraps inside and across prepare chunks
    tri = D.synthetic_fb15k_triples(fb, n_triples=T, seed=11)
    table = O.init_table(fb.entity_count, d, seed=5)
    table[::4] *= 7.0                                # rows outside the unit ball
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
    emb = dev(table).clone()
    tr = H.Trainer(emb, dev(tri), tt, B, margin=0.2, learning_rate=0.1, decay_steps=50.0, decay_rate=0.5,
                   model=model, seed=21)
    tr.global_step = 3
    tr.row = 2 * B
    losses = tr.run(steps, keep_losses=True).cpu().numpy()
    assert tr.global_step == 3 + steps
    ctab = table.copy()
    row = 2 * B
    for s in range(steps):
        if row + B > T:
            row = 0
        pos = tri[row:row + B]
        gs = 3 + s
        neg = CO.corrupt_batch(pos, id_to_type, offsets, ids, 21, gs, 1024, 0)
        lr = np.float32(0.1) / (np.float32(1.0) + np.float32(0.5) * (np.float32(gs) / np.float32(50.0)))
        closs = CO.hinge_step(ctab, pos, neg, 0.2, float(lr), hole=model.startswith("hole"), threads=16)
        assert np.abs(losses[s] - closs).max() < 2e-5, s
        row += B
    assert tr.row == row
    assert np.abs(emb.cpu().numpy() - ctab).max() < 1e-4     # fp32 drift over `steps` dependent updates
    # the last step's negatives are handed back in the scratch buffer
    assert np.array_equal(tr._neg.cpu().numpy(), neg)


@pytest.mark.parametrize("B", [2048, 8192, 20000])
def test_train_steps_fast_path_is_reproducible(H, B):
    """Two runs from the same state give bitwise-identical tables on rows with <= 16 occurrences
    per step (here: uniform ids, so every row) -- also when the step's keys are sorted across several
    workgroups (B = 8192: two tiles; B = 20000: five, the last one partial)."""
    rng = np.random.default_rng(0)
    N, d, T = 50000, 200, 90000
    tri = np.stack([rng.integers(10, N, T), rng.integers(10, N, T), rng.integers(0, 10, T)], 1).astype(np.int32)
    # relations would be hot rows (10 ids): give every triple its own pseudo-relation row instead
    tri[:, 2] = rng.integers(10, N, T)
    id_to_type = np.zeros(N, np.int32)
    offsets = np.array([0, N], np.int64)
    ids = np.arange(N, dtype=np.int32)
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=0)
    base = dev((rng.standard_normal((N, d)) * 0.05).astype(np.float32))
    outs = []
    for _ in range(2):
        emb = base.clone()
        tr = H.Trainer(emb, dev(tri), tt, B, seed=3)
        tr.run(12)
        torch.cuda.synchronize()
        outs.append(emb)
    assert torch.equal(outs[0], outs[1])


def test_train_steps_soak_across_prepare_chunks(H):
    """200 steps = 7 look-ahead prepare launches (32 steps each, two recycled buffers, side stream):
    two runs are bitwise identical, and splitting the run at arbitrary points (70 + 1 + 129 steps, i.e.
    re-entering ge_train_steps mid-chunk) changes nothing."""
    rng = np.random.default_rng(1)
    N, d, B, T = 30000, 200, 1024, 50000
    tri = np.stack([rng.integers(10, N, T), rng.integers(10, N, T), rng.integers(10, N, T)], 1).astype(np.int32)
    tt = H.TypeTables.from_host(np.zeros(N, np.int32), np.array([0, N], np.int64), np.arange(N, dtype=np.int32),
                                padded_size=0)
    base = dev((rng.standard_normal((N, d)) * 0.05).astype(np.float32))
    outs, losses = [], []
    for parts in ((200,), (200,), (70, 1, 129)):
        emb = base.clone()
        tr = H.Trainer(emb, dev(tri), tt, B, seed=5)
        for n in parts:
            last = tr.run(n)
        torch.cuda.synchronize()
        outs.append(emb)
        losses.append(last.clone())
    assert torch.isfinite(outs[0]).all() and not torch.equal(outs[0], base)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.equal(losses[0], losses[2])


# ---------------------------------------------------------------- Bernoulli filtered sampler (init.cpp)
def test_bernoulli_sampler_bit_exact_vs_pinned_oracle(H):
    from oracle import transx_oracle as TO
    rng = np.random.default_rng(4)
    R, E, T = 9, 500, 6000
    lo = R
    tri = np.unique(np.stack([lo + rng.integers(0, E, T), lo + rng.integers(0, E, T), rng.integers(0, R, T)], 1), axis=0)
    tri[:200, 0] = lo + 3                         # one head with hundreds of known (r, t)
    tri = np.unique(tri, axis=0)
    idx = TO.BernoulliIndex(tri, lo, E, R)
    smp = H.BernoulliSampler(tri, R, R + E)
    assert np.array_equal(smp.tail_threshold.cpu().numpy().view(np.uint32), idx.tail_threshold)
    pos = tri[rng.integers(0, len(tri), 3000)].astype(np.int32)
    pos[5] = [lo + 1, lo + 2, 3]                  # may be an unknown triple: unfiltered uniform draw
    pos[6, 2] = R + 4                             # invalid relation -> all -1
    for (seed, step) in ((0, 0), (123456789012345, 2**33 + 5)):
        neg = smp.corrupt(dev(pos), seed=seed, step=step).cpu().numpy()
        exp = TO.bernoulli_corrupt_batch(pos, idx, seed, step)
        assert np.array_equal(neg, exp)
    known = {tuple(x) for x in tri}
    ok = neg[:, 2] >= 0
    assert not any(tuple(int(v) for v in row) in known for row in neg[ok])


# ---------------------------------------------------------------- HolE in the frequency domain
def _packed_rfft(x):
    """NumPy statement of the packed half spectrum of ge_hole_to_spectral."""
    X = np.fft.rfft(x.astype(np.float64), axis=1)
    k = x.shape[1] // 2
    out = np.empty_like(x, dtype=np.float64)
    out[:, :k] = X[:, :k].real
    out[:, k] = X[:, k].real
    out[:, k + 1:] = X[:, 1:k].imag
    return out


@pytest.mark.parametrize("d", [200, 64, 50, 2, 1024])
def test_hole_spectral_transform_round_trip(H, d):
    rng = np.random.default_rng(d)
    N = 777
    x = (rng.standard_normal((N, d)) * rng.uniform(0.01, 0.3, (N, 1))).astype(np.float32)
    x[3] = 0.0
    t = dev(x).clone()
    H.hole_to_spectral(t)
    ref = _packed_rfft(x)
    assert np.abs(t.cpu().numpy() - ref).max() < 2e-6 * max(1.0, np.abs(ref).max())
    # Parseval with Hermitian weights: the clip norm is available without transforming back
    w = np.full(d, 2.0); w[0] = 1.0; w[d // 2] = 1.0
    assert np.abs((ref ** 2 * w).sum(1) / d - (x.astype(np.float64) ** 2).sum(1)).max() < 1e-9
    H.hole_from_spectral(t)
    assert np.abs(t.cpu().numpy() - x).max() < 2e-6 * max(1.0, np.abs(x).max())


@pytest.mark.parametrize("d", [200, 50, 128])
def test_hole_spectral_scores_and_step_match_the_hole_oracle(H, G, d):
    """README.md:42's HolE evaluated on the spectral table (ComplEx-shaped kernels with Hermitian weights):
    sigma(score), hinge and one SGD step equal the fp64 numpy.fft oracle on the real table."""
    table, pos, neg = G[f"d{d}_table"], G[f"d{d}_pos"], G[f"d{d}_neg"]
    t64 = table.astype(np.float64)
    spec = H.hole_to_spectral(dev(table).clone())
    sig = H.evaluate_triples(dev(pos), spec, model="hole_spectral").cpu().numpy()[:, 0]
    assert np.abs(sig - O.hole_evaluate_triples(pos, t64)[:, 0]).max() < SCORE_TOL
    for margin in (0.2, 0.0):
        loss = H.hinge_loss(dev(pos), dev(neg), spec, margin=margin, model="hole_spectral").cpu().numpy()[:, 0]
        assert np.abs(loss - O.evaluate_batch(pos, neg, t64, margin=margin, model="hole")[:, 0]).max() < SCORE_TOL
    work = spec.clone()
    loss = H.HingeSGD(work, len(pos), margin=0.2, model="hole_spectral").step(dev(pos), dev(neg), 0.1).cpu().numpy()[:, 0]
    new, oloss = O.sgd_step(t64, pos, neg, lr=0.1, margin=0.2, model="hole")
    assert np.abs(loss - oloss).max() < SCORE_TOL
    assert np.abs(H.hole_from_spectral(work).cpu().numpy() - new).max() < TABLE_TOL


def _config23_problem(H):
    from graphembeddings_amd import data as D
    fb = D.fb15k_shape()
    names, id_to_type, offsets, ids = fb.type_arrays()
    B, d = 4096, 200
    T = 7 * B + 13
    tri = D.synthetic_fb15k_triples(fb, n_triples=T, seed=17)
    table = O.init_table(fb.entity_count, d, seed=8)
    table[::4] *= 7.0                                # rows outside the unit ball
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
    return fb, id_to_type, offsets, ids, B, d, T, tri, table, tt


def _lr32(s):
    return float(np.float32(0.1) / (np.float32(1.0) + np.float32(0.5) * (np.float32(s) / np.float32(200.0))))


def test_config2_fifty_dependent_steps_vs_fp64_oracle(H):
    """BASELINE config 2 (FB15k-shaped, ComplEx d=200, B=4096) through ONE ge_train_steps call of 50 dependent
    steps (two prepare chunks and the look-ahead third) against the fp64 oracle replaying the same loop (same
    Philox negatives, fp32 LR schedule): every step's loss vector within 1e-5, the final table within 2e-5."""
    fb, id_to_type, offsets, ids, B, d, T, tri, table, tt = _config23_problem(H)
    steps = 50
    emb = dev(table).clone()
    tr = H.Trainer(emb, dev(tri), tt, B, margin=0.2, learning_rate=0.1, decay_steps=200.0, decay_rate=0.5, seed=33)
    losses = tr.run(steps, keep_losses=True).cpu().numpy()
    t64 = table.astype(np.float64)
    row, worst = 0, 0.0
    for s in range(steps):
        if row + B > T:
            row = 0
        pos = tri[row:row + B]
        neg = CO.corrupt_batch(pos, id_to_type, offsets, ids, 33, s, 1024, 0)
        t64, oloss = O.sgd_step(t64, pos, neg, lr=_lr32(s), margin=0.2)
        worst = max(worst, float(np.abs(losses[s] - oloss).max()))
        row += B
    assert worst < 1e-5, worst
    assert np.abs(emb.cpu().numpy() - t64).max() < 2e-5
    tr.close()


def test_config3_fifty_dependent_spectral_steps_vs_fp64_oracle(H):
    """BASELINE config 3 (FB15k-shaped, HolE d=200, B=4096): 50 dependent steps with the table held in the
    frequency domain.  HolE under the reference's hyper-parameters (lr 0.1 on the SUM gradient of 4,096 pairs) is
    a chaotic map: two runs of ANY fp32 implementation -- these kernels or the direct-correlation ones -- that
    differ only in the order of their float atomics drift apart by ~1.15x per step (tools/probes/race_probe.py:
    6e-8 after one step, 1e-5 after fifty; ComplEx stays at 6e-8).  So each step is checked against the fp64
    numpy.fft oracle started from THE DEVICE'S OWN table before that step: loss vector within 1e-5, table after
    the step within 5e-6; the free-running trajectories are compared over the first 10 steps."""
    fb, id_to_type, offsets, ids, B, d, T, tri, table, tt = _config23_problem(H)
    steps = 50
    emb = dev(table).clone()
    tr = H.Trainer(emb, dev(tri), tt, B, margin=0.2, learning_rate=0.1, decay_steps=200.0, decay_rate=0.5,
                   model="hole", seed=33, spectral_resident=True)
    free64 = table.astype(np.float64)
    before = tr.real_embeddings().cpu().numpy().astype(np.float64)
    assert np.abs(before - free64).max() < 1e-6          # the transform pair itself
    row = 0
    for s in range(steps):
        if row + B > T:
            row = 0
        pos = tri[row:row + B]
        neg = CO.corrupt_batch(pos, id_to_type, offsets, ids, 33, s, 1024, 0)
        loss = tr.run(1).cpu().numpy()
        after = tr.real_embeddings().cpu().numpy().astype(np.float64)
        exp, oloss = O.sgd_step(before, pos, neg, lr=_lr32(s), margin=0.2, model="hole")
        assert np.abs(loss - oloss).max() < 1e-5, s
        assert np.abs(after - exp).max() < 5e-6, s
        if s < 10:
            free64, floss = O.sgd_step(free64, pos, neg, lr=_lr32(s), margin=0.2, model="hole")
            assert np.abs(loss - floss).max() < 1e-5 and np.abs(after - free64).max() < 5e-6, s
        before = after
        row += B
    assert tr.global_step == steps
    tr.to_real()
    assert np.abs(emb.cpu().numpy() - before).max() < 1e-6
    tr.close()


def test_hole_resident_spectral_trainer_equals_per_call_transform(H):
    """Trainer(spectral_resident=True) transforms once and keeps the table spectral across run() calls;
    the default transforms in and out on every call.  Same losses, same final real table."""
    from graphembeddings_amd import data as D
    fb = D.fb15k_shape()
    names, id_to_type, offsets, ids = fb.type_arrays()
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
    tri = dev(D.synthetic_fb15k_triples(fb, n_triples=40000, seed=3))
    base = H.init_embeddings(fb.entity_count, 200, seed=5) * 5.0
    outs, losses = [], []
    for resident in (False, True):
        emb = base.clone()
        tr = H.Trainer(emb, tri, tt, 2048, model="hole", seed=6, spectral_resident=resident)
        ls = torch.stack([tr.run(7).clone() for _ in range(4)])
        if resident:
            v = H.evaluate_triples(tri[:100], emb, model="hole_spectral")
            assert (v - H.evaluate_triples(tri[:100], tr.real_embeddings(), model="hole")).abs().max().item() < SCORE_TOL
            tr.to_real()
        torch.cuda.synchronize()
        outs.append(emb); losses.append(ls)
        tr.close()
    # 28 HolE steps apart: the two runs differ by the roundings of the per-call transforms (1e-7 each way) and by the order
    # of the float atomics on rows with more than 16 slots, and HolE's SGD map amplifies a difference about 1.15 x per
    # step (profiles/r02_hole_chaos_probe.txt): 1e-6 ... 3e-6 is what that gives; 1e-5 is north_star's score tolerance
    assert (losses[0] - losses[1]).abs().max().item() < 1e-5
    assert (outs[0] - outs[1]).abs().max().item() < 1e-5


# ---------------------------------------------------------------- the prepare launch on its own
@pytest.mark.parametrize("B,N_extra,direct,mode", [(4096, 0, True, 0), (1000, 0, True, 1), (37, 0, False, 2),
                                                   (4096, 1_200_000, True, 0), (8192, 0, False, 0), (300, 70_000, True, 3),
                                                   (8192, 0, True, 0), (13000, 1_200_000, True, 1), (16384, 0, True, 0)])
def test_prepared_records_equal_numpy_model(H, B, N_extra, direct, mode):
    """ge_train_prepare_steps (sampler + stable radix sort by row + work-item cut) against a NumPy
    model of the same record, word for word: negatives (C oracle), the slots tagged for direct update,
    every item's (row, count, multi) and slot list in (row, slot) order.  Covers 2 radix passes (FB15k
    rows) and 3 (1.2 M rows), partial tiles, the multi-workgroup sort (B = 8192: 2 tiles, 13000: 4 with a
    partial one, 16384: 4; FB15k's hot relation rows give runs that cross tile boundaries), unknown-type rows (-1)."""
    from graphembeddings_amd import data as D
    import prep_model as PM
    fb = D.fb15k_shape()
    names, id_to_type, offsets, ids = fb.type_arrays()
    N = fb.entity_count + N_extra
    if N_extra:   # a table far larger than the typed id space: rows beyond it have no type (-1)
        id_to_type = np.concatenate([id_to_type, np.full(N_extra, -1, np.int32)])
    tri = D.synthetic_fb15k_triples(fb, n_triples=3 * B + 50, seed=5)
    if N_extra:
        idx = np.arange(0, len(tri), 2)                       # half the tails live in the high rows
        tri[idx, 1] = np.random.default_rng(3).integers(fb.entity_count, N, len(idx))
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
    lay = H.prepared_layout(B)
    steps = 3
    rec = H.prepare_steps(dev(tri), tt, B, steps, first_row=B + 7, seed=99, global_step=41, mode=mode, direct=direct).cpu().numpy()
    T = len(tri)
    row = B + 7
    for s in range(steps):
        if row + B > T:
            row = 0
        pos = tri[row:row + B]
        neg = CO.corrupt_batch(pos, id_to_type, offsets, ids, 99, 41 + s, 1024, mode)
        got = PM.parse_record(rec[s], B, lay)
        exp = PM.expected_record(pos, neg, N, direct, lay)
        assert np.array_equal(got["neg"], neg), s
        assert got["n_items"] == exp["n_items"], (s, got["n_items"], exp["n_items"])
        for sub in range(lay[1]):
            assert np.array_equal(got["items"][sub], exp["items"][sub]), (s, sub)
            assert np.array_equal(got["islots"][sub], exp["islots"][sub]), (s, sub)
        if direct:
            assert np.array_equal(got["slot_item"], exp["slot_item"]), s
        row += B


def test_train_steps_large_batch_and_fallback_branch(H):
    """B = 8192 (the step's keys sorted across two workgroups) and the fallback branch
    of ge_train_steps (workspace without room for prepared records: per-step sampler + float-atomic
    scatter) both reproduce the C port's loop."""
    from graphembeddings_amd import data as D
    fb = D.fb15k_shape()
    names, id_to_type, offsets, ids = fb.type_arrays()
    B, d, steps = 8192, 64, 4
    tri = D.synthetic_fb15k_triples(fb, n_triples=3 * B + 11, seed=13)
    table = O.init_table(fb.entity_count, d, seed=6)
    table[::5] *= 6.0
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
    ctab = table.copy()
    closs = []
    row = 0
    for s in range(steps):
        if row + B > len(tri):
            row = 0
        pos = tri[row:row + B]
        neg = CO.corrupt_batch(pos, id_to_type, offsets, ids, 4, s, 1024, 0)
        closs.append(CO.hinge_step(ctab, pos, neg, 0.2, 0.1, threads=8))
        row += B
    for prepared in (True, False):
        emb = dev(table).clone()
        tr = H.Trainer(emb, dev(tri), tt, B, margin=0.2, learning_rate=0.1, seed=4, prepared=prepared)
        losses = tr.run(steps, keep_losses=True).cpu().numpy()
        for s in range(steps):
            assert np.abs(losses[s] - closs[s]).max() < 2e-5, (prepared, s)
        assert np.abs(emb.cpu().numpy() - ctab).max() < 1e-4, prepared
        assert np.array_equal(tr._neg.cpu().numpy(), neg)
        tr.close()


@pytest.mark.parametrize("B,n_ent", [(16384, 300_000), (65536, 1_200_000)])
def test_train_steps_multi_tile_batches_vs_c_port(H, B, n_ent):
    """The native loop above B = 4096 (BASELINE config 4's per-GPU batch: 65,536 pairs on the 1.2 M-row table, and
    16,384 on a smaller one): the step's 4B keys are sorted across workgroups (ge_prep_big.hip), sole-slot rows are
    updated by the producing pair and every other row by ONE read-modify-write.  Three dependent steps against
    the C port replaying the same batches and Philox negatives."""
    from graphembeddings_amd import data as D
    data, tri = D.synthetic_large(n_entities=n_ent, n_triples=3 * B + 5, seed=77)
    names, id_to_type, offsets, ids = D.synthetic_large_type_arrays(data)
    N, d, steps = data.entity_count, 200, 3
    rng = np.random.default_rng(3)
    table = (rng.standard_normal((N, d)) * 0.05).astype(np.float32)
    table[::4] *= 7.0
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
    emb = dev(table).clone()
    tr = H.Trainer(emb, dev(tri), tt, B, margin=0.2, learning_rate=0.1, decay_steps=40.0, decay_rate=0.5, seed=9)
    losses = tr.run(steps, keep_losses=True).cpu().numpy()
    ctab, row = table, 0
    for s in range(steps):
        if row + B > len(tri):
            row = 0
        pos = tri[row:row + B]
        neg = CO.corrupt_batch(pos, id_to_type, offsets, ids, 9, s, 1024, 0)
        lr = np.float32(0.1) / (np.float32(1.0) + np.float32(0.5) * (np.float32(s) / np.float32(40.0)))
        closs = CO.hinge_step(ctab, pos, neg, 0.2, float(lr), threads=16)
        assert np.abs(losses[s] - closs).max() < 2e-5, s
        row += B
    assert np.abs(emb.cpu().numpy() - ctab).max() < 1e-4
    assert np.array_equal(tr._neg.cpu().numpy(), neg)
    tr.close()


def test_train_steps_lookahead_survives_between_calls(H):
    """The records prepared ahead on the pipeline's side stream are reused by the call that continues the
    sequence: many short calls (7 steps, train.py's tick at FB15k / B=4096), a reset in the middle, a
    call without a pipeline handle and one long call give bitwise-identical tables and losses."""
    rng = np.random.default_rng(2)
    N, d, B, T = 30000, 200, 1024, 50000
    tri = np.stack([rng.integers(10, N, T), rng.integers(10, N, T), rng.integers(10, N, T)], 1).astype(np.int32)
    tt = H.TypeTables.from_host(np.zeros(N, np.int32), np.array([0, N], np.int64), np.arange(N, dtype=np.int32),
                                padded_size=0)
    base = dev((rng.standard_normal((N, d)) * 0.05).astype(np.float32))
    outs, losses = [], []
    for variant in ("one", "sevens", "sevens_reset", "no_handle"):
        emb = base.clone()
        tr = H.Trainer(emb, dev(tri), tt, B, seed=8, lookahead=(variant != "no_handle"))
        if variant == "one":
            last = tr.run(105)
        else:
            for k in range(15):
                if variant == "sevens_reset" and k == 6:
                    tr.invalidate()
                last = tr.run(7)
        torch.cuda.synchronize()
        outs.append(emb)
        losses.append(last.clone())
        tr.close()
    for o, l in zip(outs[1:], losses[1:]):
        assert torch.equal(outs[0], o) and torch.equal(losses[0], l)


def test_score_candidates_large_tiles_path(H):
    # >= 512 tiles of 128x128: the chunked 2x2-tiles-per-wave kernel (FB15k evaluation shape family)
    N, d, B, K = 16296, 200, 2048, 8192
    g = torch.Generator(device="cpu").manual_seed(2)
    table = (torch.randn(N, d, generator=g) * 0.12).cuda()
    table[::5] *= 9.0
    hr = torch.stack([torch.randint(1345, N, (B,), generator=g), torch.randint(0, 1345, (B,), generator=g)], 1).int().cuda()
    cand = torch.randperm(N - 1345, generator=g)[:K].add(1345).int().cuda()
    for cand_is_head in (False, True):
        out = H.score_candidates(table, hr, cand, cand_is_head=cand_is_head)
        rows = torch.randint(0, B, (3000,), generator=g).cuda()
        cols = torch.randint(0, K, (3000,), generator=g).cuda()
        if cand_is_head:
            tr = torch.stack([cand[cols], hr[rows, 0], hr[rows, 1]], 1).contiguous()
        else:
            tr = torch.stack([hr[rows, 0], cand[cols], hr[rows, 1]], 1).contiguous()
        per = H.evaluate_triples(tr, table)[:, 0]
        assert (out[rows, cols] - per).abs().max().item() < SCORE_TOL


# ---------------------------------------------------------------- f2: --log_loss branch
@pytest.mark.parametrize("d,K", [(200, 1), (128, 3), (50, 2)])
def test_logloss_step_matches_oracle(H, G, d, K):
    table = G[f"d{d}_table"]
    pos = G[f"d{d}_pos"]
    rng = np.random.default_rng(K)
    negs = np.stack([pos.copy() for _ in range(K)])
    for kk in range(K):
        negs[kk, :, kk % 2] = rng.integers(8, table.shape[0], len(pos))
    lr, l2 = 0.01, 0.003
    emb = dev(table).clone()
    opt = H.LogLossSGD(emb, l2_regularization=l2)
    loss = opt.step(dev(pos), dev(negs), lr).cpu().numpy()[:, 0]
    new, oloss = O.logloss_step(table.astype(np.float64), pos, negs, lr, l2)
    assert loss.shape == ((1 + K) * len(pos),)
    assert np.abs(loss - oloss).max() < 2e-5 * max(1.0, np.abs(oloss).max())
    assert np.abs(emb.cpu().numpy() - new).max() < TABLE_TOL
    # lr = 0 leaves the table bit-identical (used for the validation loss)
    before = emb.clone()
    opt.step(dev(pos), dev(negs), 0.0)
    torch.cuda.synchronize()
    assert torch.equal(emb, before)


def test_evaluate_triples_with_label_is_the_logloss_branch(H, G):
    """evaluate_triples(triple_batch, embeddings, label) (holE.py:179, 194-196): label = +1 / -1 returns
    log(1 + exp(-label * score)) + l2 * l2_loss(embeddings); label = None the sigmoid."""
    table, pos = G["d200_table"], G["d200_pos"]
    emb = dev(table)
    for label in (1, -1):
        got = H.evaluate_triples(dev(pos), emb, label, l2_regularization=0.003).cpu().numpy()[:, 0]
        exp = O.logloss_values(pos, np.full(len(pos), float(label)), table.astype(np.float64), 0.003)
        assert np.abs(got - exp).max() < 2e-5 * max(1.0, np.abs(exp).max())
    with pytest.raises(NotImplementedError):
        H.evaluate_triples(dev(pos), emb, 1, model="hole")


def test_init_embeddings_is_the_reference_initializer(H):
    """holE.py:263-264: xavier_initializer(uniform=False) = truncated normal, sigma = sqrt(2.6 / (rows + dim)),
    re-drawn beyond 2 sigma; drawn on the device.  The two stddev constants are the ones the reference's
    graph dumps hold."""
    for (n, d, sigma_ref) in ((35910, 64, 0.00850143656135), (1134637, 128, 0.00151367869694)):
        assert abs(np.sqrt(2.6 / (n + d)) - sigma_ref) < 1e-9          # the dumps print the fp32 constant
    n, d = 35910, 64
    sigma = np.sqrt(2.6 / (n + d))
    t = H.init_embeddings(n, d, seed=5)
    assert t.is_cuda and t.dtype == torch.float32 and tuple(t.shape) == (n, d)
    assert t.abs().max().item() <= 2 * sigma * (1 + 1e-6)                  # truncation at 2 sigma
    assert abs(t.mean().item()) < 5e-5
    # std of a normal truncated at +-2 sigma: sigma * sqrt(1 - 4 phi(2) / (2 Phi(2) - 1)) = 0.87963 sigma
    assert abs(t.std().item() / sigma - 0.87963) < 3e-3
    assert (t.abs() > 1.9 * sigma).float().mean().item() > 1e-3             # the tail up to the cut is populated
    assert torch.equal(t, H.init_embeddings(n, d, seed=5)) and not torch.equal(t, H.init_embeddings(n, d, seed=6))


@pytest.mark.parametrize("B,K,d,l2", [(512, 1, 128, 1e-4), (300, 3, 50, 3e-5), (2048, 2, 200, 2e-5), (512, 1, 64, 0.0)])
def test_native_logloss_loop_matches_oracle_over_twenty_dependent_steps(H, B, K, d, l2):
    """ge_train_steps_logloss (negatives of all K corrupted batches from the prepare launch, row-sorted update, the
    dense L2 decay carried as one scalar and materialised at the end) against the fp64 oracle replaying
    holE.py:206-220 + 296 step by step: every step's loss vector and the final table.  (2048, 2): M = 6,144
    triples = two sort sub-batches, update through atomics.)"""
    from graphembeddings_amd import data as D
    fb = D.fb15k_shape()
    names, id_to_type, offsets, ids = fb.type_arrays()
    steps, T = 20, 6 * B + 31
    tri = D.synthetic_fb15k_triples(fb, n_triples=T, seed=19)
    table = O.init_table(fb.entity_count, d, seed=9)
    table[::5] *= 9.0
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
    emb = dev(table).clone()
    tr = H.Trainer(emb, dev(tri), tt, B, learning_rate=0.05, decay_steps=40.0, decay_rate=0.5, seed=77)
    tr.enable_log_loss(K, l2)
    tr.global_step = 5
    losses = torch.cat([tr.run(7, keep_losses=True), tr.run(13, keep_losses=True)], 0).cpu().numpy()   # two calls: the scalar restarts at 1
    assert losses.shape == (steps, (1 + K) * B)
    t64 = table.astype(np.float64)
    row = 0
    for s in range(steps):
        if row + B > T:
            row = 0
        pos = tri[row:row + B]
        gs = 5 + s
        negs = np.stack([CO.corrupt_batch(pos, id_to_type, offsets, ids, 77, gs * K + k, 1024, 0) for k in range(K)])
        lr = np.float32(0.05) / (np.float32(1.0) + np.float32(0.5) * (np.float32(gs) / np.float32(40.0)))
        t64, oloss = O.logloss_step(t64, pos, negs, float(lr), l2)
        assert np.abs(losses[s] - oloss).max() < 3e-5 * max(1.0, np.abs(oloss).max()), s
        row += B
    assert np.abs(emb.cpu().numpy() - t64).max() < 2e-5
    assert np.array_equal(tr._neg.cpu().numpy(), negs)
    tr.close()


def test_native_logloss_loop_survives_the_reference_default_l2(H):
    """holE.py's own defaults (lr 0.1, batch 512, l2 0.1) make the dense factor 1 - lr*M*l2 = -9.24: the table
    changes sign and grows ~9x per step.  The carried scalar is re-materialised when it leaves its range; the
    loop must follow the oracle as long as fp32 holds (6 steps: ~6e5 growth)."""
    from graphembeddings_amd import data as D
    fb = D.fb15k_shape()
    names, id_to_type, offsets, ids = fb.type_arrays()
    B, K, d, steps = 512, 1, 64, 6
    tri = D.synthetic_fb15k_triples(fb, n_triples=8 * B, seed=23)
    table = O.init_table(fb.entity_count, d, seed=10)
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
    emb = dev(table).clone()
    tr = H.Trainer(emb, dev(tri), tt, B, learning_rate=0.1, seed=3).enable_log_loss(K, 0.1)
    tr.run(steps)
    t64 = table.astype(np.float64)
    for s in range(steps):
        pos = tri[s * B:(s + 1) * B]
        negs = np.stack([CO.corrupt_batch(pos, id_to_type, offsets, ids, 3, s * K + k, 1024, 0) for k in range(K)])
        t64, _ = O.logloss_step(t64, pos, negs, 0.1, 0.1)
    scale = np.abs(t64).max()
    assert scale > 1e3                                                   # it did blow up, as the reference would
    assert np.abs(emb.cpu().numpy() - t64).max() < 1e-4 * scale
    tr.close()


@pytest.mark.parametrize("B,model", [(4096, "complex"), (16384, "complex"), (4096, "hole")])
def test_deterministic_hot_rows_are_bitwise_reproducible(H, B, model):
    """Trainer(deterministic=True) = GE_STEP_DETERMINISTIC: rows with more than 16 gradient slots in a step (FB15k's hot
    relation rows and Zipf heads: hundreds of slots) are split over several work items whose partial sums otherwise meet by
    float atomics in scheduler order.  With the flag every item parks its sum and one workgroup per hot row adds them in
    item order: two runs give the SAME BITS in every row and loss (one-tile and multi-tile steps, ComplEx and HolE on the
    half spectrum), and the table is the atomics path's to rounding."""
    from graphembeddings_amd import data as D
    fb = D.fb15k_shape()
    names, id_to_type, offsets, ids = fb.type_arrays()
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
    tri = dev(D.synthetic_fb15k_triples(fb, n_triples=6 * B, seed=3))
    base = H.init_embeddings(fb.entity_count, 200, seed=5) * 4.0
    outs = []
    for det in (True, True, False):
        emb = base.clone()
        tr = H.Trainer(emb, tri, tt, B, model=model, seed=6, deterministic=det, spectral_resident=(model == "hole"))
        ls = tr.run(5, keep_losses=True).clone()
        if model == "hole":
            tr.to_real()
        torch.cuda.synchronize()
        outs.append((emb, ls))
        tr.close()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])       # run to run: the same bits
    assert (outs[0][0] - outs[2][0]).abs().max().item() < 2e-6                                  # and the atomics path's table
    assert (outs[0][1] - outs[2][1]).abs().max().item() < 2e-6
