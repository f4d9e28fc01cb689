"""Row-sharded multi-GPU step, exercised with world_size-2 gloo processes on CPU.

The exchange logic (request counts and id lists per chunk, all_to_all of rows / gradient sums per step,
apply at the owner, look-ahead plans) is the product code of graphembeddings_amd/sharded.py; the kernel backend
it calls is replaced here by an oracle-backed double (tests may use the oracle; the product default is the HIP path and
raises without a GPU).  Correctness oracle = the single-process result on the same global batch.
"""
import os
import time
import socket
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import hole_oracle as O


class OracleKernels:
    """NumPy stand-ins with the same contracts as HipKernels (graphembeddings_amd/sharded.py): the requester plan
    (own rows in place, the other owners' distinct rows in (owner, row) order = the staging order), the step's
    gradient on the two row stores, the update of own rows + reduction of staged rows, and the owner-side add."""

    def corrupt_batch(self, tt, pos, seed, step, mode):
        neg = O.corrupt_batch(pos.numpy(), tt.id_to_type, tt.type_offsets, tt.type_ids, seed, step,
                              tt.padded_size, mode)
        return torch.as_tensor(neg)

    def gather_rows(self, table, idx):
        i = idx.numpy().astype(np.int64)
        out = np.where((i >= 0)[:, None], table.numpy()[np.clip(i, 0, None)], 0.0)
        return torch.as_tensor(out.astype(table.numpy().dtype))

    def plan_requester(self, pos, neg, n_rows, world, rank):
        from graphembeddings_amd.sharded import RequesterPlan
        pos, neg = pos.numpy(), neg.numpy()
        S, B = pos.shape[:2]
        R = (n_rows + world - 1) // world
        counts = np.zeros((S, world), np.int32)
        req_row = np.zeros((S, 4 * B), np.int32)
        steps = []
        for s in range(S):
            ids = np.unique(np.concatenate([pos[s].reshape(-1), neg[s].reshape(-1)]))
            assert ids.min() >= 0 and ids.max() < n_rows
            own = ids[ids % world == rank]
            away = ids[ids % world != rank]
            away = away[np.lexsort((away // world, away % world))]        # (owner, row): the staging order
            counts[s] = np.bincount(away % world, minlength=world)
            counts[s, rank] = len(own)
            req_row[s, :len(away)] = away // world
            src = {int(i): int(i) // world for i in own}
            src.update({int(i): R + u for u, i in enumerate(away)})
            remap = lambda t: np.vectorize(src.get)(t).astype(np.int32)
            steps.append((remap(pos[s]), remap(neg[s]), R))
        return RequesterPlan(S=S, B=B, counts=torch.as_tensor(counts), req_row=torch.as_tensor(req_row), data=steps)

    def plan_owner(self, req_all, req_start, rows_local):
        return (req_all.numpy().astype(np.int64), list(req_start))

    def grad(self, shard, staged, plan, s, lr, margin, model, max_norm, gsum):
        pos, neg, R = plan.data[s]
        rows = np.zeros((R + (0 if staged is None else staged.shape[0]), shard.shape[1]))
        rows[:shard.shape[0]] = shard.numpy()
        if staged is not None:
            rows[R:] = staged.numpy()
        idx, val, loss = O.hinge_grads(pos, neg, rows, margin, max_norm, model)
        self._g = (idx, -lr * val, R)
        return torch.as_tensor(loss.astype(np.float32))

    def apply(self, shard, plan, s, gsum):
        idx, val, R = self._g
        t = shard.numpy()
        own = (idx >= 0) & (idx < R)
        np.add.at(t, idx[own], val[own].astype(t.dtype))
        if gsum is not None:
            away = idx >= R
            np.add.at(gsum.numpy(), idx[away] - R, val[away].astype(t.dtype))

    def owner_apply(self, shard, oplan, s, recv):
        req_all, req_start = oplan
        ids = req_all[req_start[s]:req_start[s + 1]]
        keep = ids >= 0                                     # (equal-split schedule: unused slots carry id -1 and zero rows)
        np.add.at(shard.numpy(), ids[keep], recv.numpy()[keep])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem(dtype=np.float64):
    rng = np.random.default_rng(5)
    N, d, R = 301, 16, 7                      # N not divisible by the world size on purpose
    table = rng.standard_normal((N, d)) * rng.uniform(0.1, 0.6, (N, 1))
    n_types = 4
    type_of = rng.integers(0, n_types, N - R)
    ids_by_type = [np.arange(R, N)[type_of == t] for t in range(n_types)]
    offsets = np.concatenate([[0], np.cumsum([len(x) for x in ids_by_type])]).astype(np.int64)
    type_ids = np.concatenate(ids_by_type).astype(np.int32)
    id_to_type = np.concatenate([np.full(R, -1), type_of]).astype(np.int32)
    B = 64
    pos = np.stack([rng.integers(R, N, B), rng.integers(R, N, B), rng.integers(0, R, B)], 1).astype(np.int32)
    pos[:8, 0] = R + 1                        # a hot head shared by both ranks' halves
    pos[B // 2:B // 2 + 8, 0] = R + 1
    return table.astype(dtype), id_to_type, offsets, type_ids, pos


def _worker_overlap(rank, world, port, q):
    """Three steps as ONE planned chunk with the overlapped schedule: the early rows of steps 1 and 2 are fetched before
    the step in front of them has run (inline here: no streams on the CPU), the late ones behind it."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from graphembeddings_amd import sharded as S
        table, id_to_type, offsets, type_ids, pos = _problem()
        tt = SimpleNamespace(id_to_type=id_to_type, type_offsets=offsets, type_ids=type_ids, padded_size=32)
        full = torch.as_tensor(table)
        tr = S.ShardedTrainer(S.shard_rows(full, rank, world), full.shape[0], tt, margin=0.2, seed=9,
                              kernels=OracleKernels(), overlap=True)
        B = len(pos)
        mine = torch.as_tensor(pos[rank * B // world:(rank + 1) * B // world])
        chunk = torch.stack([mine, mine, mine], 0).to(torch.int32)
        plan = tr.plan_chunk(chunk, tr.sample_negatives(chunk).to(torch.int32))
        early = sum(sum(r) for r in plan.pre.sc_e)
        losses = [tr.step_planned(plan, s, 0.05) for s in range(3)]
        out = tr.gather_full_table()
        stats = torch.tensor([early, sum(sum(r) for r in plan.sc)])
        dist.all_reduce(stats)
        if rank == 0:
            q.put((out.numpy(), [l.numpy() for l in losses], float(stats[0]), float(stats[1])))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from graphembeddings_amd import sharded as S
        table, id_to_type, offsets, type_ids, pos = _problem()
        tt = SimpleNamespace(id_to_type=id_to_type, type_offsets=offsets, type_ids=type_ids, padded_size=32)
        full = torch.as_tensor(table)
        tr = S.ShardedTrainer(S.shard_rows(full, rank, world), full.shape[0], tt, margin=0.2, seed=9,
                              kernels=OracleKernels(), plan_group=dist.new_group())
        B = len(pos)
        mine = torch.as_tensor(pos[rank * B // world:(rank + 1) * B // world])
        # two pipelined chunks of one step each (the second chunk's plan is built right after the first
        # chunk's steps are issued, on its own process group), then a single step: updates of earlier
        # steps must be visible to later fetches
        # ... the first through a plan that an (empty) earlier call built as its look-ahead, the second call's
        # own look-ahead is never adopted (the single step that follows passes other positives) and must be harmless
        first = mine[None].to(torch.int32).contiguous()
        assert tr.run_pipelined([], lambda gs: 0.05, lookahead=first) is None and tr._pending is not None
        l01 = tr.run_pipelined([first, mine[None]], lambda gs: 0.05, lookahead=first)
        assert tr._pending and tr._pending[0][1] == 2                 # [(positives, first global step, plan)]
        losses = [l01[0], l01[1], tr.step(mine, lr=0.05)]
        out = tr.gather_full_table()
        mean = tr.mean_loss(losses[-1])
        if rank == 0:
            q.put((out.numpy(), [l.numpy() for l in losses], mean, tr.stats.unique_rows))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_step_equals_single_process_result(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    deadline, msg = time.time() + 240, None
    while time.time() < deadline:                    # never block on a worker that died before q.put
        if not q.empty():
            msg = q.get()
            break
        if not any(p.is_alive() for p in procs):
            break
        time.sleep(0.1)
    if msg is None and not q.empty():
        msg = q.get()
    for p in procs:
        p.join(60)
        if p.is_alive():
            p.kill()
    assert msg is not None, f"no result from the workers (exit codes {[p.exitcode for p in procs]})"
    assert all(p.exitcode == 0 for p in procs)
    got_table, got_losses, mean, uniq = msg
    # single-process reference on the same global batches, same sampler streams
    table, id_to_type, offsets, type_ids, pos = _problem()
    B = len(pos)
    ref = table.copy()
    for step in range(3):
        negs = [O.corrupt_batch(pos[r * B // world:(r + 1) * B // world], id_to_type, offsets, type_ids, 9,
                                step * world + r, 32, 0) for r in range(world)]
        neg = np.concatenate(negs, 0)
        ref, loss = O.sgd_step(ref, pos, neg, lr=0.05, margin=0.2)
        assert np.abs(got_losses[step] - loss[:B // world]).max() < 1e-6   # rank 0 holds the first slice
    assert np.abs(got_table - ref).max() < 1e-12
    assert uniq > 0 and np.isfinite(mean)


@pytest.mark.parametrize("world", [2, 3])
def test_overlapped_schedule_equals_single_process_result(world):
    """ShardedTrainer(overlap=True): rows of step s+1 that no rank touches in step s are fetched BEFORE step s runs.
    Same result as the single-process reference, and some rows really were early."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker_overlap, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    deadline, msg = time.time() + 240, None
    while time.time() < deadline:
        if not q.empty():
            msg = q.get()
            break
        if not any(p.is_alive() for p in procs):
            break
        time.sleep(0.1)
    if msg is None and not q.empty():
        msg = q.get()
    for p in procs:
        p.join(60)
        if p.is_alive():
            p.kill()
    assert msg is not None, f"no result from the workers (exit codes {[p.exitcode for p in procs]})"
    assert all(p.exitcode == 0 for p in procs)
    got_table, got_losses, early, total = msg
    table, id_to_type, offsets, type_ids, pos = _problem()
    B = len(pos)
    ref = table.copy()
    for step in range(3):
        negs = [O.corrupt_batch(pos[r * B // world:(r + 1) * B // world], id_to_type, offsets, type_ids, 9,
                                step * world + r, 32, 0) for r in range(world)]
        ref, loss = O.sgd_step(ref, pos, np.concatenate(negs, 0), lr=0.05, margin=0.2)
        assert np.abs(got_losses[step] - loss[:B // world]).max() < 1e-6
    assert np.abs(got_table - ref).max() < 1e-12
    assert 0 < early < total          # steps 1 and 2 had early rows; step 0 (and every touched row) had none


def test_single_rank_degenerates_to_plain_step():
    from graphembeddings_amd import sharded as S
    table, id_to_type, offsets, type_ids, pos = _problem()
    tt = SimpleNamespace(id_to_type=id_to_type, type_offsets=offsets, type_ids=type_ids, padded_size=32)
    tr = S.ShardedTrainer(torch.as_tensor(table.copy()), table.shape[0], tt, seed=1, kernels=OracleKernels())
    neg = O.corrupt_batch(pos, id_to_type, offsets, type_ids, 1, 0, 32, 0)
    loss = tr.step(torch.as_tensor(pos), lr=0.1)
    ref, rloss = O.sgd_step(table, pos, neg, lr=0.1, margin=0.2)
    assert np.abs(tr.shard.numpy() - ref).max() < 1e-12
    assert np.abs(loss.numpy() - rloss).max() < 1e-6


def test_product_default_kernels_need_the_gpu():
    from graphembeddings_amd import sharded as S
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    table, id_to_type, offsets, type_ids, pos = _problem(np.float32)
    tt = SimpleNamespace(id_to_type=id_to_type, type_offsets=offsets, type_ids=type_ids, padded_size=32)
    tr = S.ShardedTrainer(torch.as_tensor(table), table.shape[0], tt)
    with pytest.raises(RuntimeError):
        tr.step(torch.as_tensor(pos), lr=0.1)


def _worker_static(rank, world, port, q):
    """The same three chunks through the exact schedule, the equal-split one with a roomy capacity, with a capacity that
    one chunk overflows (re-planned exactly), and with capacity="auto"."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from graphembeddings_amd import sharded as S
        table, id_to_type, offsets, type_ids, pos = _problem()
        tt = SimpleNamespace(id_to_type=id_to_type, type_offsets=offsets, type_ids=type_ids, padded_size=32)
        B = len(pos)
        mine = torch.as_tensor(pos[rank * B // world:(rank + 1) * B // world]).to(torch.int32)
        few = mine.clone()
        few[:, 0] = few[0, 0]; few[:, 1] = few[0, 1]            # a chunk that touches very few rows (fits a tiny capacity)
        chunks = [torch.stack([mine, mine], 0).contiguous(), torch.stack([few], 0).contiguous(), torch.stack([mine], 0).contiguous()]
        # the largest per-peer request of each chunk, over all ranks (exact plans of a throwaway trainer): the tight capacity
        # is what the sparse chunk needs, so that one fits and the chunks that need more are re-planned
        probe = S.ShardedTrainer(S.shard_rows(torch.as_tensor(table.copy()), rank, world), table.shape[0], tt, margin=0.2, seed=9,
                                 kernels=OracleKernels())
        need, first = [], 0
        for c in chunks:
            pl = probe.plan_chunk(c, probe.sample_negatives(c, first).to(torch.int32))
            t = torch.tensor([max(max(max(r) for r in pl.sc), max(max(r) for r in pl.rc))])
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            need.append(int(t))
            first += c.shape[0]
        tight = need[1]
        outs, info = [], {"need": need}
        for name, cap in (("exact", None), ("roomy", 256), ("tight", tight), ("auto", "auto"), ("depth1", None), ("roomy_depth1", 256),
                          ("two_calls", None), ("roomy_two_calls", 256)):
            full = torch.as_tensor(table.copy())
            tr = S.ShardedTrainer(S.shard_rows(full, rank, world), full.shape[0], tt, margin=0.2, seed=9,
                                  kernels=OracleKernels(), capacity=cap)
            if name.endswith("depth1"):
                tr.plan_depth = 1                                  # plans one chunk ahead of the steps (rounds 2-3) instead of two
            if name.endswith("two_calls"):
                # the first call plans the second call's two chunks as its look-ahead; the second adopts both plans
                l0 = tr.run_pipelined(chunks[:1], lambda gs: 0.05, lookahead=chunks[1:])
                assert len(tr._pending) == 2 and tr._pending[0][1] == chunks[0].shape[0]
                l1 = tr.run_pipelined(chunks[1:], lambda gs: 0.05)
                losses = torch.cat([l0, l1], 0)
            else:
                losses = tr.run_pipelined(chunks, lambda gs: 0.05)
            outs.append((tr.gather_full_table().numpy(), losses.numpy()))
            info[name] = (tr.replanned_chunks, tr.capacity, tr.stats.bytes_sent)
        if rank == 0:
            q.put((outs, info))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_equal_split_schedule_is_bitwise_the_exact_one(world):
    """ShardedTrainer(capacity=C): every all-to-all moves C rows per peer (unused slots: id -1, zero rows), no split size
    reaches the host.  Tables and losses are BITWISE the exact schedule's -- with room to spare, when a chunk overflows
    the capacity (that chunk is re-planned exactly, on every rank, the others stay equal-split) and with the capacity
    taken from the first chunk; and whether plans run two chunks ahead of the steps (default) or one."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker_static, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    deadline, msg = time.time() + 240, None
    while time.time() < deadline:
        if not q.empty():
            msg = q.get()
            break
        if not any(p.is_alive() for p in procs):
            break
        time.sleep(0.1)
    if msg is None and not q.empty():
        msg = q.get()
    for p in procs:
        p.join(60)
        if p.is_alive():
            p.kill()
    assert msg is not None, f"no result from the workers (exit codes {[p.exitcode for p in procs]})"
    assert all(p.exitcode == 0 for p in procs)
    outs, info = msg
    for name, (tab, loss) in zip(("roomy", "tight", "auto", "depth1", "roomy_depth1", "two_calls", "roomy_two_calls"), outs[1:]):
        assert np.array_equal(tab, outs[0][0]) and np.array_equal(loss, outs[0][1]), name
    assert info["exact"][0] == 0 and info["roomy"][0] == 0
    need = info["need"]
    assert need[0] > need[1] and need[2] > need[1]   # the sparse chunk is the one that fits the tight capacity ...
    assert info["tight"][0] == 2                     # ... and exactly the two others were re-planned
    assert info["auto"][0] == 0 and info["auto"][1] % 64 == 0 and info["auto"][1] >= 64
    d = 16
    assert info["roomy"][2] == (world - 1) * 256 * (2 * d * 4 + 4)      # padded bytes on the links are what is reported
