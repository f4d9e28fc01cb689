"""GPU: link-prediction ranks vs the oracle's heap, and the training driver end to end."""
import os

import numpy as np
import pytest
import torch

from oracle import hole_oracle as O

pytestmark = pytest.mark.gpu


def _toy_kg(tmp_path, n_ent=120, gsz=6, seed=0):
    """A learnable toy KG (calibrated on CPU with the oracle: filtered MRR ~0.3 vs chance ~0.04):
    relation 0 links every ordered pair inside groups of 6 entities (held-out pairs are predictable
    from the others), relation 1 links entity i of group g to entity i of group g+1."""
    rng = np.random.default_rng(seed)
    R, N = 2, 2 + n_ent
    rows = [(i, f"r{i}", f"r{i}", "RELATION") for i in range(R)]
    rows += [(R + e, f"e{e}", f"e{e}", "A" if e % 3 else "B") for e in range(n_ent)]
    with open(tmp_path / "entity_metadata.tsv", "w") as f:
        f.write("Index\tId\tName\tType\n")
        for r in rows:
            f.write("\t".join(str(x) for x in r) + "\n")
    (tmp_path / "relation_ids.txt").write_text("".join(f"r{i}\t{i}\n" for i in range(R)))
    r0, r1 = [], []
    ng = n_ent // gsz
    for g in range(ng):
        mem = [R + g * gsz + i for i in range(gsz)]
        r0 += [[a, b, 0] for a in mem for b in mem if a != b]
        r1 += [[R + g * gsz + i, R + ((g + 1) % ng) * gsz + i, 1] for i in range(gsz)]
    r0 = np.array(r0, dtype=np.int64)
    rng.shuffle(r0)
    n_test, n_valid = 40, 64
    train = np.concatenate([r0[n_test + n_valid:], np.array(r1, dtype=np.int64)])
    rng.shuffle(train)
    np.savetxt(tmp_path / "test_positive_triples.txt", r0[:n_test], fmt="%d", delimiter="\t")
    np.savetxt(tmp_path / "triples-valid.txt", r0[n_test:n_test + n_valid], fmt="%d", delimiter="\t")
    np.savetxt(tmp_path / "triples.txt", train, fmt="%d", delimiter="\t")
    return str(tmp_path)


def _all_scores(emb, test, cand, side, fused):
    """[B,K] losses from the kernel the ranking path under test uses (exact ties must be that kernel's ties)."""
    from graphembeddings_amd import hole as H
    fixed_col, true_col = (0, 1) if side == "tail" else (1, 0)
    hr = torch.as_tensor(np.stack([test[:, fixed_col], test[:, 2]], 1).astype(np.int32)).cuda()
    c = torch.as_tensor(cand.astype(np.int32)).cuda()
    if fused:
        tid = torch.as_tensor(test[:, true_col].astype(np.int32)).cuda()
        return H.rank_candidates(emb, hr, tid, c, cand_is_head=(side == "head"), return_scores=True)[2].cpu().numpy()
    return H.score_candidates(emb, hr, c, cand_is_head=(side == "head")).cpu().numpy()


@pytest.mark.parametrize("fused,d", [(True, 64), (False, 64), (True, 96), (True, 128), (True, 160), (True, 192), (True, 200), (True, 56),
                                     (True, 72), (True, 104), (True, 120), (True, 136), (True, 176), (True, 208), (True, 216), (True, 232), (True, 256), (True, 288), (True, 48), (True, 40)])
def test_gpu_ranks_equal_reference_heap_semantics(fused, d):
    """Raw and filtered ranks (counted in the GEMM epilogue when fused, with tensor ops on the stored scores
    otherwise) equal the reference's heap (holE.py:427-472, oracle restatement) fed with the same losses,
    exact ties included; the losses themselves are within 1e-5 of the fp64 oracle.  embedding_dim 56 ... 288 (every number
    of 16-column k blocks from 4 to 15, and 16 and 18, full and ragged last block) run the split-precision sweep, 48 / 40 the fp32
    pipeline with chunk widths 24 / 40."""
    from graphembeddings_amd import evaluate as E
    rng = np.random.default_rng(1)
    R, N = 5, 405
    table = (rng.standard_normal((N, d)) * 0.2).astype(np.float32)
    table[50] = table[51]                                   # exact score ties between candidates
    table[60] = table[61]
    for j in range(70, 90):                                 # near-duplicates of row 52: scores a few ulps apart, inside the
        table[j] = table[52] * np.float32(1.0 + 2e-7 * (j - 79.5))   # raw-score bracket of the pipelined kernel's epilogue
    emb = torch.as_tensor(table).cuda()
    B = 150                                                 # two row blocks of the fused kernel
    test = np.stack([rng.integers(R, N, B), rng.integers(R, N, B), rng.integers(0, R, B)], 1)
    test[:4, 1] = [50, 51, 60, 61]                          # targets that tie with another candidate
    test[4:12, 1] = [52, 70, 75, 79, 80, 84, 89, 52]        # targets among the near-duplicates
    known = np.stack([np.repeat(test[:, 0], 6), rng.integers(R, N, 6 * B), np.repeat(test[:, 2], 6)], 1)
    known = known[~(known[:, None, :] == test[None, :, :]).all(-1).any(1)]      # test tails are not "known"
    cand = np.arange(R, N)
    t64 = table.astype(np.float64)
    for side in ("tail", "head"):
        kn = known if side == "tail" else known[:, [1, 0, 2]]
        kn = kn[~(kn[:, None, :] == test[None, :, :]).all(-1).any(1)]              # a test triple is never "known"
        raw, fil = E.link_prediction_ranks(emb, test, cand, kn, side=side, batch=97, fused=fused)
        all_scores = []
        for s0 in range(0, B, 97):                          # same chunking: the fused kernel's blocks restart per call
            all_scores.append(_all_scores(emb, test[s0:s0 + 97], cand, side, fused))
        all_scores = np.concatenate(all_scores, 0)
        for i, (h, t, r) in enumerate(test):
            scores = all_scores[i]
            if side == "tail":
                triples = np.stack([np.full(len(cand), h), cand, np.full(len(cand), r)], 1)
                assert np.abs(scores - O.evaluate_triples(triples, t64)[:, 0]).max() < 1e-5
                true = O.triple_dict(kn[(kn[:, 0] == h) & (kn[:, 2] == r)])
                tst = O.triple_dict([[h, t, r]])
                rp, fp = [], []
                O.eval_link_prediction(scores, triples, true, tst, rp, fp)
            else:
                # mirror problem: rank heads; reuse the tail-ranking heap on swapped columns
                triples = np.stack([cand, np.full(len(cand), t), np.full(len(cand), r)], 1)
                assert np.abs(scores - O.evaluate_triples(triples, t64)[:, 0]).max() < 1e-5
                sw = np.stack([np.full(len(cand), t), cand, np.full(len(cand), r)], 1)
                true = O.triple_dict(kn[(kn[:, 1] == t) & (kn[:, 2] == r)][:, [1, 0, 2]])
                tst = O.triple_dict([[t, h, r]])
                rp, fp = [], []
                O.eval_link_prediction(scores, sw, true, tst, rp, fp)
            assert [raw[i]] == rp and [fil[i]] == fp, (side, i, raw[i], rp, fil[i], fp)


@pytest.mark.parametrize("d,model", [(200, "complex"), (104, "hole_spectral"), (40, "complex")])
def test_rank_planes_built_once_equal_planes_built_per_call(d, model):
    """ge_rank_planes / ge_rank_1vK_planes: the candidates' fp16 planes built once (H.RankPlanes) give the counts and losses
    of the call that builds them itself, tails and heads, with a candidate list that is not a multiple of 128, bad ids in
    it, and true ids that are not candidates (no rank: counts 0, loss NaN).  d = 40 has no split-precision sweep: the
    planes object is empty and ignored."""
    from graphembeddings_amd import hole as H
    rng = np.random.default_rng(11)
    N, B = 700, 300
    table = (rng.standard_normal((N, d)) * 0.3).astype(np.float32)
    emb = torch.as_tensor(table).cuda()
    if model == "hole_spectral":
        emb = H.hole_to_spectral(emb)
    cand_np = rng.permutation(np.arange(10, N - 40))[:517].astype(np.int32)     # 517 candidates: five 128-tiles, the last ragged
    cand_np[5] = -3; cand_np[400] = N + 7                                        # bad ids: never rank before anything
    cand = torch.as_tensor(cand_np).cuda()
    hr = torch.as_tensor(np.stack([rng.integers(0, N, B), rng.integers(0, 10, B)], 1).astype(np.int32)).cuda()
    tid_np = rng.choice(cand_np[(cand_np >= 0) & (cand_np < N)][:500], B).astype(np.int32)
    tid_np[7] = N - 3; tid_np[100] = N - 5                                      # not in the candidate list
    tid = torch.as_tensor(tid_np).cuda()
    planes = H.RankPlanes(emb, cand, model=model)
    assert (planes.buffer is None) == (d == 40)
    for head in (False, True):
        a = H.rank_candidates(emb, hr, tid, cand, cand_is_head=head, return_true_loss=True, model=model)
        b = H.rank_candidates(emb, hr, tid, cand, cand_is_head=head, return_true_loss=True, model=model, planes=planes)
        for x, y in zip(a, b):
            assert torch.equal(torch.nan_to_num(x.float(), nan=-1.0), torch.nan_to_num(y.float(), nan=-1.0))
        nb, nk, tl = b
        if d != 40:                                                             # the split-precision sweep's contract
            assert int(nb[7]) == 0 and int(nb[100]) == 0 and bool(torch.isnan(tl[7])) and bool(torch.isnan(tl[100]))
        # against the stored scores
        sc = H.rank_candidates(emb, hr, tid, cand, cand_is_head=head, return_scores=True, model=model, planes=planes)[2]
        col = (cand.view(1, -1) == tid.view(-1, 1)).float().argmax(1)
        st = sc.gather(1, col.view(-1, 1))
        ok = torch.ones(B, dtype=torch.bool, device="cuda"); ok[7] = ok[100] = False
        ref = ((sc < st) | ((sc == st) & (cand.view(1, -1) < tid.view(-1, 1)))).sum(1).int()
        assert torch.equal(nb[ok], ref[ok])
    with pytest.raises(ValueError):
        H.rank_candidates(emb, hr, tid, cand[:-1].contiguous(), planes=planes) if d != 40 else (_ for _ in ()).throw(ValueError())


def test_known_cells_kernel_matches_brute_force():
    from graphembeddings_amd import evaluate as E_
    """ge_known_cells (evaluate.KnownIndex.cells): the per-(128 x 128)-tile lists of known-true cells the fused ranking
    kernel takes, against a brute-force enumeration; duplicates in the known triples count once; a test row whose fixed
    entity has no known triple and an empty index give empty lists."""
    rng = np.random.default_rng(7)
    N, R, B, K = 900, 6, 300, 700
    known = np.stack([rng.integers(R, N, 4000), rng.integers(R, N, 4000), rng.integers(0, R, 4000)], 1)
    known = np.concatenate([known, known[:500]])                    # duplicates
    cand = np.sort(rng.permutation(np.arange(R, N))[:K])
    pos_of = torch.full((N,), -1, dtype=torch.int64)
    pos_of[torch.as_tensor(cand)] = torch.arange(K)
    pos_dev = pos_of.cuda()
    test = known[rng.integers(0, len(known), B)]
    for side in ("tail", "head"):
        fc, oc = (0, 1) if side == "tail" else (1, 0)
        idx = E_.KnownIndex(known, N, side, torch.device("cuda"))
        off, rc = idx.cells(torch.as_tensor(test[:, fc]).cuda(), torch.as_tensor(test[:, 2]).cuda(), pos_dev, K)
        off, rc = off.cpu(), rc.cpu()
        n_ct = (K + 127) // 128
        got = set()
        off = off.numpy()
        for tile in range(len(off) - 1):
            for v in rc.numpy()[off[tile]:off[tile + 1]].astype(np.int64):
                got.add(((tile // n_ct) * 128 + v // 128, (tile % n_ct) * 128 + v % 128))
        exp = set()
        kn = {(int(a[fc]), int(a[2]), int(a[oc])) for a in known}
        by = {}
        for f, r, o in kn:
            by.setdefault((f, r), []).append(o)
        for i, t in enumerate(test):
            for o in by.get((int(t[fc]), int(t[2])), []):
                if pos_of[o] >= 0:
                    exp.add((i, int(pos_of[o])))
        assert got == exp and int(off[-1]) == len(exp)
    empty = E_.KnownIndex(None, N, "tail", torch.device("cuda"))
    off, rc = empty.cells(torch.as_tensor(test[:, 0]).cuda(), torch.as_tensor(test[:, 2]).cuda(), pos_dev, K)
    assert int(off.abs().sum()) == 0 and off.numel() == ((B + 127) // 128) * ((K + 127) // 128) + 1


@pytest.mark.parametrize("B,K", [(1, 1), (1, 33), (129, 31), (5, 128), (300, 129)])
def test_rank_sweep_small_and_ragged_shapes(B, K):
    """The split-precision sweep at its smallest and most ragged shapes (one row, one candidate; a candidate list shorter
    than a wave's 32 columns; one row more than a row block; exactly one tile; one candidate more than a tile): counts
    equal those implied by the stored scores, nothing is counted for the padding."""
    from graphembeddings_amd import hole as H
    rng = np.random.default_rng(100 * B + K)
    N, d = 400, 200
    emb = torch.as_tensor((rng.standard_normal((N, d)) * 0.25).astype(np.float32)).cuda()
    cand = torch.as_tensor(rng.permutation(np.arange(20, N))[:K].astype(np.int32)).cuda()
    hr = torch.as_tensor(np.stack([rng.integers(20, N, B), rng.integers(0, 20, B)], 1).astype(np.int32)).cuda()
    tid = cand[torch.as_tensor(rng.integers(0, K, B)).cuda()].contiguous()
    for head in (False, True):
        nb, nk = H.rank_candidates(emb, hr, tid, cand, cand_is_head=head)[:2]
        nb1, _, sc = H.rank_candidates(emb, hr, tid, cand, cand_is_head=head, return_scores=True)
        col = (cand.view(1, -1) == tid.view(-1, 1)).float().argmax(1)
        st = sc.gather(1, col.view(-1, 1))
        ref = ((sc < st) | ((sc == st) & (cand.view(1, -1) < tid.view(-1, 1)))).sum(1).int()
        assert torch.equal(nb, ref) and torch.equal(nb1, ref) and int(nk.abs().sum()) == 0


def test_rank_sweep_many_candidates():
    """150,001 candidates (1,172 tiles of 128, the last with one column; 125 MB of planes), 130 test rows, ids up to
    200,000: the counts of the rank-only sweep equal those implied by the stored scores, and with a known-true list the
    filtered counts equal a direct evaluation on those scores."""
    from graphembeddings_amd import evaluate as E_
    from graphembeddings_amd import hole as H
    rng = np.random.default_rng(77)
    N, d, B, K = 200_000, 200, 130, 150_001
    emb = (torch.randn(N, d, generator=torch.Generator().manual_seed(5)) * 0.2).cuda()
    cand_np = rng.permutation(np.arange(100, N))[:K].astype(np.int32)
    cand = torch.as_tensor(cand_np).cuda()
    hr = torch.as_tensor(np.stack([rng.integers(100, N, B), rng.integers(0, 100, B)], 1).astype(np.int32)).cuda()
    tid = cand[torch.as_tensor(rng.integers(0, K, B)).cuda()].contiguous()
    known = np.stack([np.repeat(hr[:, 0].cpu().numpy(), 40), rng.choice(cand_np, 40 * B), np.repeat(hr[:, 1].cpu().numpy(), 40)], 1)
    pos_of = torch.full((N,), -1, dtype=torch.int64, device="cuda")
    pos_of[cand.long()] = torch.arange(K, device="cuda")
    off, rc = E_.KnownIndex(known, N, "tail", torch.device("cuda")).cells(hr[:, 0].long(), hr[:, 1].long(), pos_of, K)
    planes = H.RankPlanes(emb, cand)
    nb, nk = H.rank_candidates(emb, hr, tid, cand, known_off=off, known_rc=rc, planes=planes)[:2]
    sc = H.rank_candidates(emb, hr, tid, cand, return_scores=True, planes=planes)[2]
    col = pos_of[tid.long()]
    st = sc.gather(1, col.view(-1, 1))
    before = (sc < st) | ((sc == st) & (cand.view(1, -1) < tid.view(-1, 1)))
    assert torch.equal(nb, before.sum(1).int())
    kn = torch.zeros(B, dtype=torch.int64, device="cuda")
    cells = {(int(r), int(pos_of[int(e)])) for r, e in zip(np.repeat(np.arange(B), 40), known[:, 1])}
    rows = torch.as_tensor([c[0] for c in cells]).cuda(); cols = torch.as_tensor([c[1] for c in cells]).cuda()
    kn.index_add_(0, rows, before[rows, cols].long())
    assert torch.equal(nk.long(), kn)


@pytest.mark.parametrize("d", [64, 200, 40, 56])
def test_hole_ranks_from_the_spectral_sweep(d):
    """HolE link prediction (README.md:42 score): the sweep on the table held in the frequency domain gives losses
    within 1e-5 of the oracle's FFT-based HolE score on the REAL table, and ranks equal to the reference heap fed with
    the kernel's own losses -- tails and heads; split-precision sweep (d = 56, 64, 200) and fp32 pipeline (d = 40)."""
    from graphembeddings_amd import evaluate as E
    from graphembeddings_amd import hole as H
    rng = np.random.default_rng(5)
    R, N = 4, 300
    table = (rng.standard_normal((N, d)) * 0.25).astype(np.float32)     # some rows above the clip norm, some below
    table[::7] *= 0.2
    emb = torch.as_tensor(table).cuda()
    spec = H.hole_to_spectral(emb.clone())
    B = 140
    test = np.stack([rng.integers(R, N, B), rng.integers(R, N, B), rng.integers(0, R, B)], 1)
    known = np.stack([np.repeat(test[:, 0], 3), rng.integers(R, N, 3 * B), np.repeat(test[:, 2], 3)], 1)
    known = known[~(known[:, None, :] == test[None, :, :]).all(-1).any(1)]
    cand = np.arange(R, N)
    c = torch.as_tensor(cand.astype(np.int32)).cuda()
    t64 = table.astype(np.float64)
    for side in ("tail", "head"):
        kn = known if side == "tail" else known[:, [1, 0, 2]]
        kn = kn[~(kn[:, None, :] == test[None, :, :]).all(-1).any(1)]              # a test triple is never "known"
        raw, fil = E.link_prediction_ranks(emb, test, cand, kn, side=side, model="hole")
        fixed_col, true_col = (0, 1) if side == "tail" else (1, 0)
        hr = torch.as_tensor(np.stack([test[:, fixed_col], test[:, 2]], 1).astype(np.int32)).cuda()
        tid = torch.as_tensor(test[:, true_col].astype(np.int32)).cuda()
        scores = H.rank_candidates(spec, hr, tid, c, cand_is_head=(side == "head"), return_scores=True,
                                   model="hole_spectral")[2].cpu().numpy()
        for i, (h, t, r) in enumerate(test):
            if side == "tail":
                triples = np.stack([np.full(len(cand), h), cand, np.full(len(cand), r)], 1)
                heap_triples = triples
                true = O.triple_dict(kn[(kn[:, 0] == h) & (kn[:, 2] == r)])
                tst = O.triple_dict([[h, t, r]])
            else:
                triples = np.stack([cand, np.full(len(cand), t), np.full(len(cand), r)], 1)
                heap_triples = np.stack([np.full(len(cand), t), cand, np.full(len(cand), r)], 1)
                true = O.triple_dict(kn[(kn[:, 1] == t) & (kn[:, 2] == r)][:, [1, 0, 2]])
                tst = O.triple_dict([[t, h, r]])
            if i < 25:
                assert np.abs(scores[i] - O.hole_evaluate_triples(triples, t64)[:, 0]).max() < 1e-5
            rp, fp = [], []
            O.eval_link_prediction(scores[i], heap_triples, true, tst, rp, fp)
            assert [raw[i]] == rp and [fil[i]] == fp, (side, i, raw[i], rp, fil[i], fp)
    with pytest.raises(ValueError):
        E.link_prediction_ranks(emb, test, cand, known, model="hole", fused=False)


@pytest.mark.parametrize("d,max_norm", [(200, 2.0), (64, 8.0), (200, 8.0)])
def test_spectral_sweep_single_bin_rows_above_unit_max_norm(d, max_norm):
    """The split-precision sweep on a SPECTRAL table with max_norm > 1 and rows whose energy sits in one frequency bin
    (constant rows: the DC bin; one cosine: one Hermitian-weighted bin).  The clip bounds a row's Parseval norm, so a
    single bin reaches max_norm sqrt(d/2) and a weighted product d max_norm^2: the 1/d of the correlation theorem
    must be folded in before the fp16 split or the planes overflow (max_norm > ~1.1 at d = 200).  Losses within 1e-5
    of the FFT oracle on the real table, ranks equal to the reference heap fed with the kernel's own losses."""
    from graphembeddings_amd import hole as H
    rng = np.random.default_rng(d)
    R, N = 4, 260
    j = np.arange(d)
    table = (rng.standard_normal((N, d)) * 0.3).astype(np.float32)
    table[0:R] = 3.0 * max_norm / np.sqrt(d)                                    # relations: constant rows, norm 3 max_norm
    table[10:40] = (2.5 * max_norm / np.sqrt(d)) * np.ones((30, d), np.float32)  # constant entities
    table[40:70] = (4.0 * max_norm * np.sqrt(2.0 / d) * np.cos(2 * np.pi * 3 * j / d)).astype(np.float32)   # one cosine bin
    table[40:70] *= rng.uniform(0.5, 1.5, (30, 1)).astype(np.float32)
    table[10:40] *= rng.uniform(0.5, 1.5, (30, 1)).astype(np.float32)
    spec = H.hole_to_spectral(torch.as_tensor(table).cuda())
    B = 130
    test = np.stack([rng.integers(R, 80, B), rng.integers(R, N, B), rng.integers(0, R, B)], 1)
    cand = np.arange(R, N)
    c = torch.as_tensor(cand.astype(np.int32)).cuda()
    t64 = table.astype(np.float64)
    for side in ("tail", "head"):
        fixed_col, true_col = (0, 1) if side == "tail" else (1, 0)
        hr = torch.as_tensor(np.stack([test[:, fixed_col], test[:, 2]], 1).astype(np.int32)).cuda()
        tid = torch.as_tensor(test[:, true_col].astype(np.int32)).cuda()
        nb, nk, scores = H.rank_candidates(spec, hr, tid, c, cand_is_head=(side == "head"), return_scores=True,
                                           model="hole_spectral", max_norm=max_norm)
        nb2, _ = H.rank_candidates(spec, hr, tid, c, cand_is_head=(side == "head"), model="hole_spectral", max_norm=max_norm)
        assert torch.equal(nb, nb2)                                             # bracket epilogue == exact epilogue
        scores, nb = scores.cpu().numpy(), nb.cpu().numpy()
        assert np.isfinite(scores).all()
        for i, (h, t, r) in enumerate(test):
            triples = (np.stack([np.full(len(cand), h), cand, np.full(len(cand), r)], 1) if side == "tail"
                       else np.stack([cand, np.full(len(cand), t), np.full(len(cand), r)], 1))
            if i < 30:
                assert np.abs(scores[i] - O.hole_evaluate_triples(triples, t64, max_norm=max_norm)[:, 0]).max() < 1e-5
            true = test[i, true_col]
            st = scores[i][true - R]
            before = int(((scores[i] < st) | ((scores[i] == st) & (cand < true))).sum())
            assert nb[i] == before, (side, i)


@pytest.mark.parametrize("d", [40, 56])
def test_rank_sweep_edge_shapes_and_bad_ids(d):
    """One row / one candidate, sizes one past the 128-wide tiles, a max_norm below every row norm, ids outside the
    table: a bad (fixed, relation) row or true id ranks nothing before it, a bad candidate is never counted, nothing
    faults (pipelined kernel at d = 40, generic at 56)."""
    from graphembeddings_amd import hole as H
    rng = np.random.default_rng(11)
    N = 300
    table = (rng.standard_normal((N, d)) * 0.3).astype(np.float32)
    emb = torch.as_tensor(table).cuda()

    def heap_counts(scores, cand_ids, tid):
        out = []
        for i in range(scores.shape[0]):
            st = scores[i, list(cand_ids).index(tid[i])]
            out.append(int(((scores[i] < st) | ((scores[i] == st) & (np.asarray(cand_ids) < tid[i]))).sum()))
        return out

    for B, K, mn in ((1, 1, 1.0), (129, 257, 1.0), (5, 130, 0.5)):
        cand = rng.permutation(np.arange(4, N))[:K].astype(np.int32)
        hr = np.stack([rng.integers(4, N, B), rng.integers(0, 4, B)], 1).astype(np.int32)
        tid = cand[rng.integers(0, K, B)].astype(np.int32)
        for head in (False, True):
            nb, nk, sc = H.rank_candidates(emb, torch.as_tensor(hr).cuda(), torch.as_tensor(tid).cuda(),
                                           torch.as_tensor(cand).cuda(), cand_is_head=head, max_norm=mn, return_scores=True)
            nb2, _ = H.rank_candidates(emb, torch.as_tensor(hr).cuda(), torch.as_tensor(tid).cuda(),
                                       torch.as_tensor(cand).cuda(), cand_is_head=head, max_norm=mn)
            exp = heap_counts(sc.cpu().numpy(), cand, tid)
            assert nb.cpu().tolist() == exp and nb2.cpu().tolist() == exp and int(nk.abs().sum()) == 0
    # ids outside the table
    B, K = 130, 200
    cand = np.arange(4, 4 + K).astype(np.int32)
    hr = np.stack([rng.integers(4, N, B), rng.integers(0, 4, B)], 1).astype(np.int32)
    tid = cand[rng.integers(0, K, B)].astype(np.int32)
    good = H.rank_candidates(emb, torch.as_tensor(hr).cuda(), torch.as_tensor(tid).cuda(), torch.as_tensor(cand).cuda())[0].cpu().numpy()
    hr_b, tid_b, cand_b = hr.copy(), tid.copy(), cand.copy()
    hr_b[3, 0] = N + 5
    hr_b[64, 1] = -1
    tid_b[7] = -2
    cand_b[10] = N + 100                                 # never counted; rows whose true id it was are excluded below
    got = H.rank_candidates(emb, torch.as_tensor(hr_b).cuda(), torch.as_tensor(tid_b).cuda(), torch.as_tensor(cand_b).cuda())[0].cpu().numpy()
    assert got[3] == 0 and got[64] == 0 and got[7] == 0
    sc = H.rank_candidates(emb, torch.as_tensor(hr).cuda(), torch.as_tensor(tid).cuda(), torch.as_tensor(cand).cuda(),
                           return_scores=True)[2].cpu().numpy()
    for i in range(B):
        if i in (3, 64, 7) or tid[i] == cand[10]:
            continue
        st = sc[i, tid[i] - 4]
        before_bad = (sc[i, 10] < st) or (sc[i, 10] == st and cand[10] < tid[i])
        assert got[i] == good[i] - int(before_bad)


def test_fused_and_unfused_rankers_agree_at_fb15k_scale():
    """Full-width sweep (14,951 candidates, 117 column tiles, d = 200): the epilogue-counted ranks and the
    ranks from stored scores differ only where two candidates' losses are within a few ulps of each other in one
    kernel and not the other (different k order, 4-instruction sigmoid in the epilogue); on random tables that
    is a rank or two on a few percent of the rows."""
    from graphembeddings_amd import data as D
    from graphembeddings_amd import evaluate as E
    from graphembeddings_amd import hole as H
    inf = D.init_inference_data(D.PACKAGE_FB15K_DIR)
    emb = H.init_embeddings(inf.entity_count, 200, seed=3) * 4.0
    R, N = inf.relation_count, inf.entity_count
    cand = np.arange(R, N, dtype=np.int32)
    test = inf.test_array[:700]
    known = inf.validation_triples
    for side in ("tail", "head"):
        r1, f1 = E.link_prediction_ranks(emb, test, cand, known, side=side, fused=True)
        r2, f2 = E.link_prediction_ranks(emb, test, cand, known, side=side, fused=False)
        assert np.abs(r1 - r2).max() <= 3 and np.mean(r1 != r2) < 0.06
        assert np.abs(f1 - f2).max() <= 3 and np.mean(f1 != f2) < 0.06
        assert (f1 <= r1).all() and (f1 >= 1).all()


def test_training_driver_end_to_end(tmp_path):
    from graphembeddings_amd import data as D
    from graphembeddings_amd import train as T
    dd = tmp_path / "data"
    dd.mkdir()
    data_dir = _toy_kg(dd)
    out = str(tmp_path / "run")
    argv = ["--data_dir", data_dir, "--output_dir", out, "--batch_size", "64", "--embedding_dim", "32",
            "--num_epochs", "150", "--learning_rate", "0.5", "--margin", "0.5", "--padded_size", "64", "--seed", "1"]
    FLAGS = T.build_parser().parse_args(argv)
    data = D.init_data(data_dir)
    logs = []
    res = T.run_training(data, FLAGS, log=lambda *a: logs.append(" ".join(str(x) for x in a)))
    batch_count = data.triple_count // 64
    assert res["steps"] == 150 * (batch_count - 1)           # an epoch is batch_count-1 steps (holE.py:340)
    assert os.path.exists(T.checkpoint_path(out))
    assert res["pocket_loss"] < 0.4                          # started at margin 0.5; learning happened
    assert any("Validation Loss" in l for l in logs)
    # the output-dir guard of holE.py:254-255 and --resume_checkpoint
    with pytest.raises(Exception, match="already exists"):
        T.run_training(data, FLAGS, log=lambda *a: None)
    FLAGS2 = T.build_parser().parse_args(argv + ["--resume_checkpoint", "--num_epochs", "1"])
    res2 = T.run_training(data, FLAGS2, log=lambda *a: None)
    assert res2["global_step"] > res2["steps"] > 0            # global_step restored (LR decay resumes)
    # --infer: filtered MRR far above chance (random ranking over 120 candidates: MRR ~ 0.04)
    # (--infer_threshold 1.0: every sweep's lowest loss is below it, so every position is recorded; the reference's default
    # 0.05 gates the positions of holE.py:464-466 on is_confident, holE.py:436-438)
    m = T.infer_triples(T.build_parser().parse_args(argv + ["--infer", "--infer_threshold", "1.0"]), log=lambda *a: None)
    assert m["recorded"] == m["sweeps"] == 80
    assert m["filtered_mrr"] > 0.15 and m["filtered_mrr"] >= m["raw_mrr"]
    assert m["hits10"] > 50 and m["mean_filtered_pos"] < 20
    gated = T.infer_triples(T.build_parser().parse_args(argv + ["--infer"]), log=lambda *a: None)     # default threshold 0.05
    assert gated["recorded"] <= gated["sweeps"] and (gated["recorded"] > 0 or np.isnan(gated["filtered_mrr"]))


def test_fb15k_scale_ranks_match_oracle_on_real_id_files():
    """FB15k id space (real entity_metadata / valid / test files shipped with the package): raw and
    filtered tail ranks over all 14,951 entities from the MFMA candidate sweep equal the reference's
    heap semantics (holE.py:427-472) evaluated by the oracle on the same scores."""
    from graphembeddings_amd import data as D
    from graphembeddings_amd import evaluate as E
    from graphembeddings_amd import hole as H
    inf = D.init_inference_data(D.PACKAGE_FB15K_DIR)
    assert inf.entity_count == 16296 and len(inf.test_array) == 59071
    emb = H.init_embeddings(inf.entity_count, 200, seed=3) * 4.0
    R, N = inf.relation_count, inf.entity_count
    cand = np.arange(R, N, dtype=np.int32)
    test = inf.test_array[:40]
    known = inf.validation_triples
    raw, fil = E.link_prediction_ranks(emb, test, cand, known, side="tail")          # fused epilogue ranks
    all_scores = _all_scores(emb, test, cand, "tail", True)
    true = O.triple_dict(known)
    for i, (h, t, r) in enumerate(test):
        s = all_scores[i]
        triples = np.stack([np.full(len(cand), h), cand, np.full(len(cand), r)], 1)
        rp, fp = [], []
        O.eval_link_prediction(s, triples, true, O.triple_dict([[h, t, r]]), rp, fp)
        assert [raw[i]] == rp and [fil[i]] == fp
    m = E.mrr_and_hits(raw, fil)
    assert 0 < m["filtered_mrr"] <= 1 and m["filtered_mrr"] >= m["raw_mrr"]


def test_log_loss_driver_runs(tmp_path):
    """--log_loss --negative_ratio 2 with a small l2: the loop of holE.py:206-220 end to end."""
    from graphembeddings_amd import data as D
    from graphembeddings_amd import train as T
    dd = tmp_path / "data"
    dd.mkdir()
    data_dir = _toy_kg(dd)
    out = str(tmp_path / "run_ll")
    argv = ["--data_dir", data_dir, "--output_dir", out, "--batch_size", "64", "--embedding_dim", "32",
            "--num_epochs", "40", "--learning_rate", "0.05", "--log_loss", "--negative_ratio", "2",
            "--l2_regularization", "1e-5", "--padded_size", "64", "--seed", "1"]
    FLAGS = T.build_parser().parse_args(argv)
    data = D.init_data(data_dir)
    res = T.run_training(data, FLAGS, log=lambda *a: None)
    assert res["steps"] == 40 * (data.triple_count // 64 - 1)
    assert np.isfinite(res["final_mean_hinge"]) and res["final_mean_hinge"] < np.log(2.0) + 0.05
    assert res["pocket_loss"] < 0.69      # below log(2): the model separates positives from negatives


def test_fb15k_mrr_parity_gpu_path_vs_cpu_port():
    """BASELINE metric, second clause ("FB15k MRR parity"), as far as the reference's data allows (tests/mrr_parity.py):
    ComplEx d=200 B=4096 trained on the real FB15k validation split by the native loop and by the C port replaying the
    same batches and Philox negatives, then ranked both ways.  The GPU sweep and the reference heap fed with the sweep's
    own losses agree on every rank; the two training paths end within fp32 rounding of each other, so the heap on the
    CPU port's table (fp64 losses) gives the same ranks except where other candidates' losses sit within 5e-7 of the
    true one (each difference is bounded by the number of such candidates), and the filtered MRRs agree to 1e-3.  No reference MRR exists to compare with (none published, no train split): unpinned."""
    import mrr_parity
    res = mrr_parity.run(n_steps=400, n_test=30)
    n = res["ranks"]["n"]
    assert n == 60
    assert res["train"]["max_abs_loss_diff_any_step"] < 2e-5 and res["train"]["max_abs_table_diff"] < 1e-4
    same = res["ranks"]["sweep_vs_reference_heap_on_the_sweeps_own_losses"]
    assert same["raw_equal"] == n and same["filtered_equal"] == n
    cross = res["ranks"]["gpu_path_vs_cpu_port_with_fp64_losses"]
    assert cross["differences_explained_by_candidates_within_5e-7_of_the_true_loss"] == n
    assert res["filtered_mrr_abs_diff"] < 1e-3
    assert res["metrics_gpu_path"]["filtered_mrr"] > 1.5 * res["metrics_untrained"]["filtered_mrr"]     # it learned


@pytest.mark.parametrize("d,model", [(200, "complex"), (64, "complex"), (104, "hole_spectral"), (40, "complex"), (48, "complex"), (16, "complex")])
def test_ranks_against_given_losses_add_over_candidate_shards(d, model):
    """ge_rank_1vK_vs_loss: (1) ranking every row against ITS OWN true loss and id gives ge_rank_1vK_planes' counts;
    (2) the candidate list cut into 3 ragged shards (the true candidate is in one of them, or -- rows 0..9 -- in none):
    the shards' counts against the true loss ADD to the one-list counts, raw and known, bit for bit: what a row-sharded
    evaluation all-reduces (holE.py:427-472 positions are sums over disjoint candidate sets)."""
    from graphembeddings_amd import evaluate as E
    from graphembeddings_amd import hole as H
    rng = np.random.default_rng(d)
    R, N, B = 4, 700, 300
    table = (rng.standard_normal((N, d)) * 0.2).astype(np.float32)
    table[50] = table[51]; table[60] = table[61]            # exact ties: the id decides
    emb = torch.as_tensor(table).cuda()
    if model == "hole_spectral":
        emb = H.hole_to_spectral(emb)
    cand = np.arange(R, N).astype(np.int32)
    test = np.stack([rng.integers(R, N, B), rng.integers(R, N, B), rng.integers(0, R, B)], 1)
    test[:4, 1] = [50, 51, 60, 61]
    known = np.stack([np.repeat(test[:, 0], 5), rng.integers(R, N, 5 * B), np.repeat(test[:, 2], 5)], 1)
    known = known[~(known[:, None, :] == test[None, :, :]).all(-1).any(1)]
    for side in ("tail", "head"):
        fixed_col, true_col = (0, 1) if side == "tail" else (1, 0)
        kn = known if side == "tail" else known[:, [1, 0, 2]]
        kn = kn[~(kn[:, None, :] == test[None, :, :]).all(-1).any(1)]
        hr = torch.as_tensor(np.stack([test[:, fixed_col], test[:, 2]], 1).astype(np.int32)).cuda()
        tid = torch.as_tensor(test[:, true_col].astype(np.int32)).cuda()
        index = E.KnownIndex(kn, N, side, emb.device)

        def cells(c):
            pos_of = torch.full((N,), -1, dtype=torch.int64, device="cuda")
            pos_of[torch.as_tensor(c.astype(np.int64)).cuda()] = torch.arange(len(c), device="cuda")
            return index.cells(hr[:, 0].long(), hr[:, 1].long(), pos_of, len(c))
        dc = torch.as_tensor(cand).cuda()
        off, rc = cells(cand)
        nb, nk, tl = H.rank_candidates(emb, hr, tid, dc, known_off=off, known_rc=rc, cand_is_head=(side == "head"),
                                       return_true_loss=True, model=model)
        nb2, nk2 = H.rank_candidates_vs_loss(emb, hr, tid, tl, dc, known_off=off, known_rc=rc, cand_is_head=(side == "head"), model=model)
        assert torch.equal(nb, nb2) and torch.equal(nk, nk2), side
        perm = rng.permutation(cand)
        cuts = [0, 231, 500, len(cand)]
        sb = torch.zeros_like(nb); sk = torch.zeros_like(nk)
        for a, b in zip(cuts, cuts[1:]):
            c = np.sort(perm[a:b])
            o2, r2 = cells(c)
            x, y = H.rank_candidates_vs_loss(emb, hr, tid, tl, torch.as_tensor(c).cuda(), known_off=o2, known_rc=r2,
                                             cand_is_head=(side == "head"), model=model)
            sb += x; sk += y
        assert torch.equal(sb, nb) and torch.equal(sk, nk), side


@pytest.mark.parametrize("d,fused", [(200, True), (64, True), (40, True), (16, True), (50, False)])     # f16 sweep, fp32 pipeline, generic kernel, stored scores
def test_is_confident_gate_equals_the_reference_heap(d, fused):
    """--infer_threshold (holE.py:436-438, 464-466): only sweeps whose lowest loss is below the threshold record their
    positions.  link_prediction_ranks(infer_threshold=...) against the oracle's restatement fed with the sweep's own losses;
    thresholds chosen between the rows' minima (so some rows are confident and some are not) and one ON a row's minimum
    (strict <)."""
    from graphembeddings_amd import evaluate as E
    rng = np.random.default_rng(7)
    R, N, B = 3, 260, 90
    table = (rng.standard_normal((N, d)) * 0.45).astype(np.float32)
    emb = torch.as_tensor(table).cuda()
    test = np.stack([rng.integers(R, N, B), rng.integers(R, N, B), rng.integers(0, R, B)], 1)
    cand = np.arange(R, N)
    scores = _all_scores(emb, test, cand, "tail", fused)
    mins = np.sort(scores.min(1))
    for thr in (float((mins[B // 3] + mins[B // 3 + 1]) / 2), float(mins[B // 2]), 0.0, 1.5):
        raw, fil, conf = E.link_prediction_ranks(emb, test, cand, None, side="tail", fused=fused, infer_threshold=thr, return_confident=True)
        rp, fp, exp_conf = [], [], []
        for i, (h, t, r) in enumerate(test):
            triples = np.stack([np.full(len(cand), h), cand, np.full(len(cand), r)], 1)
            n0 = len(rp)
            O.eval_link_prediction(scores[i], triples, O.triple_dict([]), O.triple_dict([[h, t, r]]), rp, fp, infer_threshold=thr)
            exp_conf.append(len(rp) > n0)
        assert list(conf) == exp_conf, thr
        assert list(raw) == rp and list(fil) == fp, thr
        assert 0 < sum(exp_conf) < B or thr in (0.0, 1.5)
