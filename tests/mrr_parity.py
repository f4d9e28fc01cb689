"""FB15k MRR parity, as far as the data in the reference allows (BASELINE metric's second clause; holE.py:427-490).

The reference ships FB15k's id files, its validation split (50,000 triples) and its test split, but NOT the train
split (.MISSING_LARGE_BLOBS:1-3) and publishes no MRR: parity against a REFERENCE MRR is therefore unpinned.  What
can be pinned is that the HIP path and the CPU port of the reference's arithmetic, fed the same data, end at the same
ranks:
  * train ComplEx d=200, B=4096 on the real validation split (real ids / types) for a fixed number of steps with
    the native loop (hole.Trainer, what train.py drives) and with the C port (oracle/ge_oracle.c) replaying the same
    batches and the same Philox negatives;
  * rank test triples (both sides, all 14,951 entities as candidates, train triples filtered) with the GPU sweep
    (evaluate.link_prediction_ranks) on the GPU-trained table and with the reference's heap (O.eval_link_prediction)
    on the CPU-trained table, and also with the heap on the GPU-trained table (evaluator alone).
Test infrastructure (uses the oracle); `python tests/mrr_parity.py` writes profiles/r04_mrr_parity.json."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _heap_ranks(losses_of, test, cand, known, O):
    """Raw / filtered ranks of the reference heap (holE.py:427-472) for both sides.  losses_of(side, i, triples) ->
    the [C] losses the heap is fed for test triple i.  Also returns, per evaluation, the true entity's loss and how
    many candidates lie within 5e-7 of it (the candidates whose order fp32 rounding can change)."""
    raw, fil, near = [], [], []
    kt = O.triple_dict(known)
    kh = O.triple_dict(known[:, [1, 0, 2]])
    for side in ("tail", "head"):
        for i, (h, t, r) in enumerate(test):
            if side == "tail":
                tri = np.stack([np.full(len(cand), h), cand, np.full(len(cand), r)], 1)
                heap_tri, true, tst, true_id = tri, kt, O.triple_dict([[h, t, r]]), t
            else:
                tri = np.stack([cand, np.full(len(cand), t), np.full(len(cand), r)], 1)
                heap_tri = np.stack([np.full(len(cand), t), cand, np.full(len(cand), r)], 1)   # the heap's (fixed, candidate, rel)
                true, tst, true_id = kh, O.triple_dict([[t, h, r]]), h
            loss = np.asarray(losses_of(side, i, tri), dtype=np.float64)
            rp, fp = [], []
            O.eval_link_prediction(loss, heap_tri, true, tst, rp, fp)
            raw += rp; fil += fp
            lt = loss[int(true_id - cand[0])]
            near.append(int((np.abs(loss - lt) < 5e-7).sum()) - 1)
    return np.array(raw, np.int64), np.array(fil, np.int64), np.array(near, np.int64)


def run(n_steps=1200, n_test=500, B=4096, d=200, seed=7, threads=16, init_scale=5.0):
    import torch
    from graphembeddings_amd import data as D
    from graphembeddings_amd import evaluate as E
    from graphembeddings_amd import hole as H
    from oracle import c_oracle as CO
    from oracle import hole_oracle as O

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
    known_set = {tuple(x) for x in train.tolist()}
    test = np.array([x for x in test.tolist() if tuple(x) not in known_set][:n_test], np.int64)
    cand = np.arange(fb.relation_count, fb.entity_count)
    decay_steps = 32.0 * (T // B)

    # ---- the HIP path: the native loop on the device-resident triple array (no reshuffle: the C port replays it)
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
    # The reference's Xavier rows have norm ~0.18 and a trilinear score's gradient is quadratic in that scale: on 50,000
    # triples the model needs thousands of steps to leave the origin, and until then every candidate's loss is 0.5 +-
    # 1e-5 (ranks decided by fp32 rounding).  The experiment starts from the same rows times `init_scale`.
    table0 = (O.init_table(fb.entity_count, d, seed=seed) * np.float32(init_scale)).astype(np.float32)
    emb = torch.as_tensor(table0).cuda()
    tr = H.Trainer(emb, torch.as_tensor(train).cuda(), tt, B, margin=0.2, learning_rate=0.1, decay_steps=decay_steps,
                   decay_rate=0.5, seed=seed)
    t0 = time.time()
    gl = tr.run(n_steps, keep_losses=True).cpu().numpy()
    torch.cuda.synchronize()
    gpu_train_s = time.time() - t0
    tr.close()

    # ---- the C port: same batches (incl. the wrap that never yields a short batch), same negatives, fp32 LR schedule
    ctab = table0.copy()
    row, worst_loss = 0, 0.0
    t0 = time.time()
    for s in range(n_steps):
        if row + B > T:
            row = 0
        pos = train[row:row + B]
        neg = CO.corrupt_batch(pos, id_to_type, offsets, ids, seed, s, 1024, 0)
        lr = np.float32(0.1) / (np.float32(1.0) + np.float32(0.5) * (np.float32(s) / np.float32(decay_steps)))
        cl = CO.hinge_step(ctab, pos, neg, 0.2, float(lr), threads=threads)
        worst_loss = max(worst_loss, float(np.abs(gl[s] - cl).max()))
        row += B
    cpu_train_s = time.time() - t0
    gtab = emb.cpu().numpy()
    table_diff = float(np.abs(gtab - ctab).max())

    # ---- ranks
    t0 = time.time()
    raw_g, fil_g, own = [], [], {}
    for side in ("tail", "head"):
        r_, f_ = E.link_prediction_ranks(emb, test, cand, train, side=side)
        raw_g.append(np.asarray(r_)); fil_g.append(np.asarray(f_))
    raw_g, fil_g = np.concatenate(raw_g), np.concatenate(fil_g)
    gpu_eval_s = time.time() - t0
    c32 = torch.as_tensor(cand.astype(np.int32)).cuda()
    for side, (fc, tc) in (("tail", (0, 1)), ("head", (1, 0))):       # the sweep's own fp32 losses, for the heap
        hr = torch.as_tensor(np.stack([test[:, fc], test[:, 2]], 1).astype(np.int32)).cuda()
        tid = torch.as_tensor(test[:, tc].astype(np.int32)).cuda()
        own[side] = H.rank_candidates(emb, hr, tid, c32, cand_is_head=(side == "head"), return_scores=True)[2].cpu().numpy()
    t0 = time.time()
    c64 = ctab.astype(np.float64)
    # (X) the reference heap fed with the sweep's own losses: the ranking logic alone -- must agree everywhere
    raw_x, fil_x, _ = _heap_ranks(lambda side, i, tri: own[side][i], test, cand, train, O)
    # (C) the reference heap on the CPU port's table with the fp64 oracle's losses
    raw_c, fil_c, near = _heap_ranks(lambda side, i, tri: O.evaluate_triples(tri, c64)[:, 0], test, cand, train, O)
    cpu_eval_s = time.time() - t0
    m_g, m_c, m_x = E.mrr_and_hits(raw_g, fil_g), E.mrr_and_hits(raw_c, fil_c), E.mrr_and_hits(raw_x, fil_x)
    t0t = torch.as_tensor(table0).cuda()
    m_0 = E.mrr_and_hits(*[np.concatenate([np.asarray(a) for a in z]) for z in zip(
        *[E.link_prediction_ranks(t0t, test, cand, train, side=s_) for s_ in ("tail", "head")])])
    diff = np.abs(fil_g - fil_c)
    return {
        "dataset": f"FB15k id files shipped with the reference: {T} triples of the validation split as training set, "
                   f"{len(test)} test triples (entities / relation seen in training), 14,951 candidates per side, train triples filtered",
        "model": f"complex d={d} B={B} hinge margin 0.2 lr 0.1 (inverse-time decay), type-safe negatives, {n_steps} steps, seed {seed}, Xavier init x {init_scale}",
        "train": {"max_abs_loss_diff_any_step": worst_loss, "max_abs_table_diff": table_diff,
                  "gpu_seconds": round(gpu_train_s, 3), "cpu_port_seconds": round(cpu_train_s, 3), "cpu_threads": threads},
        "ranks": {"n": int(len(raw_g)),
                  "sweep_vs_reference_heap_on_the_sweeps_own_losses": {"raw_equal": int((raw_g == raw_x).sum()),
                                                                       "filtered_equal": int((fil_g == fil_x).sum())},
                  "gpu_path_vs_cpu_port_with_fp64_losses": {
                      "raw_equal": int((raw_g == raw_c).sum()), "filtered_equal": int((fil_g == fil_c).sum()),
                      "max_abs_rank_diff": int(diff.max()),
                      "differences_explained_by_candidates_within_5e-7_of_the_true_loss": int((diff <= near).sum()),
                      "note": ("every rank difference sits where other candidates' losses are within fp32 rounding of the true one"
                               if int((diff <= near).sum()) == int(len(diff)) else
                               f"{int(len(diff)) - int((diff <= near).sum())} rank differences are NOT near-ties: the two fp32 training "
                               f"trajectories have diverged (tables differ by {table_diff:.1e}, losses by {worst_loss:.1e}), so the two "
                               "sides rank different tables; both sit near the untrained MRR, so this record says little about a "
                               "trained table")}},
        "metrics_gpu_path": m_g, "metrics_cpu_port": m_c, "metrics_reference_heap_on_sweep_losses": m_x, "metrics_untrained": m_0,
        "filtered_mrr_abs_diff": abs(m_g["filtered_mrr"] - m_c["filtered_mrr"]),
        "eval_seconds": {"gpu_sweep_both_sides": round(gpu_eval_s, 3), "reference_heap_two_tables": round(cpu_eval_s, 1)},
        "parity_against_a_reference_mrr": "unpinned: the reference publishes no MRR and ships no train split (SURVEY.md 6, .MISSING_LARGE_BLOBS:1-3)",
    }


if __name__ == "__main__":
    # two lengths: 400 steps (the two fp32 trajectories are still within 1e-4 of each other: every rank difference is a
    # near-tie) and 1200 steps (more learning; the trajectories have drifted apart by the summation order of the same
    # row updates, so a few ranks differ by more than the near-tie count -- the MRRs still agree to 1e-6)
    n_test = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    res = {"runs": [run(n_steps=400, n_test=n_test), run(n_steps=1200, n_test=n_test)]}
    out = os.path.join(ROOT, "gpurun_out", "r04_mrr_parity.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))
