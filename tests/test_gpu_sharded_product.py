"""GPU: the row-sharded PRODUCT path at world size 2 -- `train.py --gpus 2` (training driver, checkpoints, resume, --infer)
and the sharded evaluator -- processes sharing the test box's one GPU, collectives through gloo (RCCL needs two GPUs: it has
not run this path).  Expected results: the one-GPU kernels / evaluator on the same inputs."""
import json
import os
import subprocess
import sys
import time
import traceback
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_world(target, world, args=(), timeout=420):
    """Start `world` spawn-processes of target(rank, world, port, q, *args); rank 0 puts ("ok", ...) or anyone ("error", ...)."""
    import socket
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(args)) for r in range(world)]
    for p in procs:
        p.start()
    deadline, msg = time.time() + timeout, None
    while time.time() < deadline:
        if not q.empty():
            msg = q.get()
            break
        if not any(p.is_alive() for p in procs):
            break
        time.sleep(0.2)
    if msg is None and not q.empty():
        msg = q.get()
    for p in procs:
        p.join(60)
        if p.is_alive():
            p.kill()
    assert msg is not None, f"no result from the workers (exit codes {[p.exitcode for p in procs]})"
    assert msg[0] == "ok", f"rank {msg[1]} failed:\n{msg[2]}"
    assert all(p.exitcode == 0 for p in procs)
    return msg[1:]


def _env(rank, world, port):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "LOCAL_RANK": str(rank),
                       "WORLD_SIZE": str(world), "GE_DIST_BACKEND": "gloo", "GE_SINGLE_DEVICE": "1"})


# ------------------------------------------------------------------------------------------ the sharded evaluator
def _eval_problem():
    rng = np.random.default_rng(5)
    R, N, d, n_test = 7, 2500, 200, 500
    table = (rng.standard_normal((N, d)) * 0.25).astype(np.float32)
    table[1000] = table[1001]; table[1500] = table[1503]; table[77] = table[1078]     # exact ties across and inside shards
    test = np.stack([rng.integers(R, N, n_test), rng.integers(R, N, n_test), rng.integers(0, R, n_test)], 1)
    test[:6, 1] = [1000, 1001, 1500, 1503, 77, 1078]
    test[6:9, 0] = [1000, 1503, 77]
    known = np.stack([np.repeat(test[:, 0], 8), rng.integers(R, N, 8 * n_test), np.repeat(test[:, 2], 8)], 1)
    known = np.concatenate([known, np.stack([rng.integers(R, N, 8 * n_test), np.repeat(test[:, 1], 8), np.repeat(test[:, 2], 8)], 1)])
    known = known[~(known[:, None, :] == test[None, :, :]).all(-1).any(1)]
    return R, N, d, table, test, known


def _eval_worker(rank, world, port, q, thr, model="complex"):
    import torch.distributed as dist
    try:
        _env(rank, world, port)
        from graphembeddings_amd import sharded_train as ST
        rk, wd, dev = ST.dist_setup()
        try:
            R, N, d, table, test, known = _eval_problem()
            shard = torch.as_tensor(np.ascontiguousarray(table[rank::world])).to(dev)
            if model == "hole_spectral":                  # the HolE score on the half spectrum: the shard's rows transformed in place
                from graphembeddings_amd import hole as H
                shard = H.hole_to_spectral(shard)
            res = ST.evaluate_sharded(shard, N, R, test, known, both_sides=True, infer_threshold=thr, batch=300, rank=rk, world=wd,
                                      model=model)
            if rank == 0:
                q.put(("ok",) + tuple(np.asarray(a) for a in res))
            dist.barrier()
        finally:
            dist.destroy_process_group()
    except Exception:
        q.put(("error", rank, traceback.format_exc()))
        raise


@pytest.mark.parametrize("world,thr,model", [(2, None, "complex"), (3, None, "complex"), (2, "median", "complex"), (2, None, "hole_spectral")])
def test_sharded_evaluator_ranks_are_bit_equal_to_the_one_gpu_evaluator(world, thr, model):
    """500 test triples, both sides, train/valid-style filter, exact ties between rows of different and of the same shard:
    raw and filtered ranks from evaluate_sharded (candidates sharded like the table, counts all-reduced) EQUAL
    evaluate.link_prediction_ranks on the whole table -- and with --infer_threshold the same sweeps are recorded."""
    from graphembeddings_amd import evaluate as E
    from graphembeddings_amd import hole as H
    R, N, d, table, test, known = _eval_problem()
    emb = torch.as_tensor(table).cuda()
    if model == "hole_spectral":
        emb = H.hole_to_spectral(emb)
    cand = np.arange(R, N)
    if thr == "median":          # a threshold that cuts: the median over the test rows of the tail sweeps' lowest loss
        hr = torch.as_tensor(np.stack([test[:, 0], test[:, 2]], 1).astype(np.int32)).cuda()
        thr = float(H.score_candidates(emb, hr, torch.as_tensor(cand.astype(np.int32)).cuda()).min(1).values.median()) + 1e-4
    got = _run_world(_eval_worker, world, (thr, model))
    exp_raw, exp_fil, exp_conf = [], [], []
    for side in ("tail", "head"):
        r = E.link_prediction_ranks(emb, test, cand, known, side=side, infer_threshold=thr, return_confident=True, model=model)
        exp_raw.append(r[0]); exp_fil.append(r[1]); exp_conf.append(r[2])
    assert np.array_equal(got[0], np.concatenate(exp_raw))
    assert np.array_equal(got[1], np.concatenate(exp_fil))
    if thr is not None:
        conf = np.concatenate(exp_conf)
        assert np.array_equal(got[2], conf) and 0 < conf.sum() < conf.size     # the gate really cut


# ------------------------------------------------------------------------------------------ the training driver
def _train_flags(data_dir, out_dir, **kw):
    from graphembeddings_amd import train as T
    argv = ["--data_dir", data_dir, "--output_dir", out_dir, "--gpus", "2"]
    for k, v in kw.items():
        argv += [f"--{k}"] + ([] if v is True else [str(v)])
    return T.build_parser().parse_args(argv)


def _write_fb15k_shaped(tmp, n_triples=60000, n_valid=4000, n_test=300):
    """The FB15k-shaped problem as files the driver reads: the real id / type files' shape, synthetic triples."""
    import shutil
    from graphembeddings_amd import data as D
    fb = D.fb15k_shape()
    tri = D.synthetic_fb15k_triples(fb, n_triples=n_triples + n_valid + n_test, seed=3)
    for f in os.listdir(D.PACKAGE_FB15K_DIR):                     # the reference's own id / type files (4-column metadata)
        if f.startswith("entity_metadata") or f.startswith("relation_ids"):
            shutil.copy(os.path.join(D.PACKAGE_FB15K_DIR, f), os.path.join(tmp, f))
    D.write_triples(os.path.join(tmp, "triples.txt"), tri[:n_triples])
    D.write_triples(os.path.join(tmp, "triples-valid.txt"), tri[n_triples:n_triples + n_valid])
    D.write_triples(os.path.join(tmp, "test_positive_triples.txt"), tri[n_triples + n_valid:])
    return tmp


def _train_worker(rank, world, port, q, data_dir, out_dir, steps):
    import torch.distributed as dist
    try:
        _env(rank, world, port)
        from graphembeddings_amd import data as D
        from graphembeddings_amd import sharded_train as ST
        try:
            FLAGS = _train_flags(data_dir, out_dir, batch_size=2048, embedding_dim=200, max_steps=steps, seed=11, num_epochs=1)
            data = D.init_data(data_dir, cache=False)
            data.validation_triples = None                       # (no ticks: this test replays the training steps alone)
            res = ST.run_training_sharded(data, FLAGS, log=lambda *a, **k: None)
            if rank == 0:
                q.put(("ok", res["steps"], res["final_mean_hinge"]))
            dist.barrier()
        finally:
            if dist.is_initialized():
                dist.destroy_process_group()
    except Exception:
        q.put(("error", rank, traceback.format_exc()))
        raise


def test_sharded_training_driver_equals_the_one_gpu_step_on_the_same_global_batches(tmp_path):
    """run_training_sharded (world 2, FB15k-shaped data from files, global batch 2048, 20 steps, no validation ticks):
    the table it leaves in model.ckpt.pt equals the ONE-GPU fused step (hole.HingeSGD: ge_complex_hinge_step) fed with
    the same global batches -- each rank's head-owned, per-rank-shuffled triples, its Philox negatives -- and the same
    learning rates, within the table tolerance."""
    from graphembeddings_amd import data as D
    from graphembeddings_amd import hole as H
    from graphembeddings_amd import sharded_train as ST
    from graphembeddings_amd import train as T
    data_dir = _write_fb15k_shaped(str(tmp_path))
    out_dir = str(tmp_path / "out")
    steps, world, Bg, seed = 20, 2, 2048, 11
    n, mean_hinge = _run_world(_train_worker, world, (data_dir, out_dir, steps))
    assert n == steps and np.isfinite(mean_hinge)
    got, gs = T.load_checkpoint(out_dir)
    assert gs == steps
    assert all(os.path.exists(ST.shard_checkpoint_path(out_dir, r, world)) for r in range(world))
    # ---- replay on one GPU
    data = D.init_data(data_dir, cache=False)
    names, id_to_type, offsets, ids = data.type_arrays()
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
    emb = H.init_embeddings(data.entity_count, 200, seed=seed)
    B = Bg // world
    local = []
    for r in range(world):
        mine = torch.as_tensor(ST.partition_by_head(data.triples, r, world).astype(np.int32)).cuda()
        gen = torch.Generator(device="cuda").manual_seed(seed * 1000003 + r)
        local.append(mine[torch.randperm(mine.shape[0], device="cuda", generator=gen)])
    opt = H.HingeSGD(emb, Bg, margin=0.2)
    batch_count = data.triple_count // Bg
    for s in range(steps):
        b = 1 + s                                                         # the epoch's batches start at 1 (holE.py:340)
        pos = [local[r][((b % (local[r].shape[0] // B)) * B):((b % (local[r].shape[0] // B)) + 1) * B] for r in range(world)]
        neg = [H.corrupt_batch(tt, 0, pos[r].contiguous(), seed=seed, step=s * world + r) for r in range(world)]
        lr = H.inverse_time_decay(0.1, s, 32 * batch_count, 0.5)
        opt.step(torch.cat(pos, 0).contiguous(), torch.cat(neg, 0).contiguous(), lr)
    assert float((emb - got).abs().max()) < 2e-5


@pytest.mark.parametrize("model", ["complex", "hole"])
def test_train_cli_gpus_2_trains_checkpoints_resumes_and_infers(tmp_path, model):
    """`python -m graphembeddings_amd.train --gpus 2` with no launcher (the program starts its own two ranks): validation
    ticks with a scalar all-reduce, the pocket per shard, --resume_checkpoint from the shard files, and --infer --gpus 2
    printing the SAME numbers as --infer on one GPU from the gathered checkpoint."""
    data_dir = _write_fb15k_shaped(str(tmp_path), n_triples=40000, n_valid=3000, n_test=200)
    out_dir = str(tmp_path / "run")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update({"GE_DIST_BACKEND": "gloo", "GE_SINGLE_DEVICE": "1", "PYTHONPATH": ROOT + os.pathsep + env.get("PYTHONPATH", "")})
    base = [sys.executable, "-m", "graphembeddings_amd.train", "--data_dir", data_dir, "--output_dir", out_dir,
            "--batch_size", "1024", "--embedding_dim", "64", "--num_epochs", "1", "--seed", "3", "--model", model]
    # (--model hole: the shards train in the frequency domain, the files hold real-valued rows, both --infer paths transform
    # their copy and rank with the HolE score)

    def run(extra, timeout=600):
        p = subprocess.run(base + extra, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
        assert p.returncode == 0, p.stdout[-1500:] + "\n" + p.stderr[-3000:]
        return p.stdout
    out = run(["--gpus", "2"])
    assert "Validation Loss" in out and "Done training" in out and out.count("Training epoch 1") == 1     # rank 0 alone prints
    for f in ("model.ckpt.pt", "model.ckpt.shard0-of-2.pt", "model.ckpt.shard1-of-2.pt", "model.ckpt.shards.json"):
        assert os.path.exists(os.path.join(out_dir, f)), f
    meta = json.load(open(os.path.join(out_dir, "model.ckpt.shards.json")))
    assert meta["world"] == 2 and meta["validation_loss"] < 2.0
    # an existing output directory is refused without --resume_checkpoint (holE.py:254-255), on every rank
    p = subprocess.run(base + ["--gpus", "2"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode != 0 and "already exists" in (p.stdout + p.stderr)
    out2 = run(["--gpus", "2", "--resume_checkpoint", "--max_steps", "5"])
    assert "Done training" in out2
    inf2 = run(["--gpus", "2", "--infer", "--infer_threshold", "0.6"])
    inf1 = run(["--infer", "--infer_threshold", "0.6"])
    line2 = [l for l in inf2.splitlines() if l.startswith("raw MRR")]
    line1 = [l for l in inf1.splitlines() if l.startswith("raw MRR")]
    assert line1 and line2 == line1, (line1, line2)
    rec2 = [l for l in inf2.splitlines() if l.startswith("is_confident")]
    rec1 = [l for l in inf1.splitlines() if l.startswith("is_confident")]
    assert rec1 and rec2 == rec1
