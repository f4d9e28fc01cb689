"""bench.py's N>1 leg: BASELINE config[3] -- synthetic 1.2 M entities / 30 M triples, d=200, the table
row-sharded over the N GPUs of one node, rows and gradient sums routed by RCCL all-to-all over xGMI.
Weak scaling: every rank trains B positives per step; value = 2*B*N*K / max-over-ranks time."""
from __future__ import annotations

import json
import os
import time

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0


def run(args, emit=True):
    from . import data as D
    from . import hole as H
    from . import sharded as S

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    # rehearsal knobs (one-GPU boxes): GE_DIST_BACKEND=gloo GE_SINGLE_DEVICE=1 run every rank on cuda:0
    backend = os.environ.get("GE_DIST_BACKEND", "nccl")
    if os.environ.get("GE_SINGLE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    d, B, K, W = args.dim, args.batch, args.steps, args.warmup
    n_rel, n_ent = 18, args.entities
    N = n_rel + n_ent

    # type tables (replicated, small next to the table) and this rank's slice of the triples
    data, _ = D.synthetic_large(n_entities=n_ent, n_triples=1, seed=1234)
    names, id_to_type, offsets, ids = D.synthetic_large_type_arrays(data)
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024, device=dev)
    # Triples are partitioned BY THE OWNER OF THEIR HEAD (rank r trains the triples whose head row it
    # holds): a third of a step's entity rows -- and every head-corrupted negative's type lookups aside --
    # are then local, and the row all-to-all shrinks accordingly.  Heads Zipf(0.8) over the rank's own rows,
    # tails Zipf(0.8) over all entities, relations uniform.
    n_loc = max(args.triples // world, B * 8)
    rng = np.random.default_rng(1234 + 7919 * rank)
    if world > 1 and getattr(args, "partition", "head") == "head":
        # Zipf(0.8) over the entity rows this rank owns, drawn directly (rejection from the global distribution
        # costs `world` times the samples: half a minute of host time per rank at 8 ranks)
        own = np.flatnonzero((n_rel + np.arange(n_ent)) % world == rank)
        head = n_rel + own[D._zipf_sample(rng, len(own), n_loc, 0.8)]
    else:
        head = n_rel + D._zipf_sample(rng, n_ent, n_loc, 0.8)
    tail = n_rel + D._zipf_sample(rng, n_ent, n_loc, 0.8)
    rel = rng.integers(0, n_rel, size=n_loc)
    dtri = torch.as_tensor(np.stack([head, tail, rel], 1).astype(np.int32)).to(dev)
    del head, tail, rel

    # this rank's shard, initialised in place (truncated normal, sigma = sqrt(2.6/(N+d)), holE.py:263-264)
    rows = S.shard_num_rows(N, rank, world)
    std = float(np.sqrt(2.6 / (N + d)))
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)
    shard = torch.empty(rows, d, device=dev)
    torch.nn.init.trunc_normal_(shard, 0.0, std, -2 * std, 2 * std, generator=gen)

    # The planner's two collectives may get their own communicator (--plan-group) so that they overlap the
    # data path's all-to-alls.  Default off: two RCCL communicators whose kernels are enqueued from two
    # streams have no cross-rank launch order, the classic multi-communicator hang, and this path has not
    # run on a multi-GPU node yet; on ONE communicator every rank enqueues in program order.
    plan_group = dist.new_group(backend=backend) if (world > 1 and getattr(args, "plan_group", False)) else None
    tr = S.ShardedTrainer(shard, N, tt, margin=0.2, model=args.model, seed=0, plan_group=plan_group,
                          peer_mapped=bool(getattr(args, "peer_mapped", False)), overlap=bool(getattr(args, "overlap", False)),
                          capacity=(None if getattr(args, "capacity", None) in (None, "") else
                                    ("auto" if args.capacity == "auto" else int(args.capacity))))
    batch_count = args.triples // (B * world)
    decay_steps = 32.0 * batch_count
    ev = H.Events(2)
    grad_ms = []
    orig_grad = tr.k.grad

    def timed_grad(*a, **kw):
        ev.record(0)
        out = orig_grad(*a, **kw)
        ev.record(1)
        timed_grad.pending = True
        return out
    timed_grad.pending = False

    CHUNK = 64   # steps per exchange plan (one sort / count exchange / few host syncs per chunk; the plan of
                 # chunk c+1 is built on a side stream while chunk c trains).  On ONE communicator the plan's count
                 # exchange queues behind the training chunk's all-to-alls, so the host learns the split sizes only
                 # when that chunk is done and the device waits for the next chunk's first launch once per chunk:
                 # long chunks keep that boundary rare (64 steps x 65,536 pairs: 0.5 GB of plan records)

    def lr_fn(gs):
        return H.inverse_time_decay(0.1, gs, decay_steps, 0.5)

    def make_chunks(first, n):
        chunks = []
        i = first
        while i < first + n:
            m = min(CHUNK, first + n - i)
            rows = [((j * B) % (n_loc - B)) for j in range(i, i + m)]
            chunks.append(torch.stack([dtri[r:r + B] for r in rows], 0).contiguous())
            i += m
        return chunks

    def one_step(i):
        return tr.run_pipelined(make_chunks(i, 1), lr_fn)[-1]

    # Steady state of a long run: every chunk's plan is built while the chunk before it trains.  The warm-up
    # call therefore plans the timed region's first chunk, and the timed call plans the chunk that would follow
    # it -- the timed region holds K steps and the planning of K steps, none of it on the critical path.
    warm, timed, after = make_chunks(0, W), make_chunks(W, K), make_chunks(W + K, min(2 * CHUNK, K))
    tr.run_pipelined(warm, lr_fn, lookahead=timed[:2])                   # (plans run two chunks ahead of the steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = tr.run_pipelined(timed, lr_fn, lookahead=after[:2])[-1]
    t_host = time.perf_counter() - t0                    # the host's share: enqueueing K steps and planning the chunk after them
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    stats = tr.stats                                     # of the timed region's last chunk
    # kernel timing pass (outside the timed region: the event sync would serialise the pipeline)
    tr.k.grad = timed_grad
    for i in range(W + K, W + K + 10):
        loss = one_step(i)
        torch.cuda.synchronize()
        grad_ms.append(ev.elapsed_ms(0, 1))
    tr.k.grad = orig_grad
    ev.close()
    t = torch.tensor([el], device=dev, dtype=torch.float64)
    seen = torch.ones(1, device=dev, dtype=torch.int32)   # one word per rank, summed: the ranks that really took part
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
    el = float(t.item())
    ranks_seen = int(seen.item())
    mean_loss = tr.mean_loss(loss)
    if rank == 0:
        kern_ms = float(np.median(grad_ms))
        alg = (24 * d + 28) * B
        achieved = alg / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "scored triples/sec/GPU (d=200)", "value": 2.0 * B * world * K / el,
            "unit": "scored triples/s", "n_gpus": world, "ranks_seen": ranks_seen,
            "backend": (backend if world > 1 else "none (one rank: no collective runs)"),
            "devices": ("all ranks on cuda:0 (GE_SINGLE_DEVICE=1 rehearsal)" if os.environ.get("GE_SINGLE_DEVICE") == "1" and world > 1
                        else "one device per rank"),
            "steps": K, "warmup": W,
            "ms_per_step": el / K * 1e3, "host_enqueue_ms_per_step": t_host / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic {n_ent} entities / {args.triples} triples, {args.model} d={d}, "
                                   f"table row-sharded (id % N) over {world} GPU(s), triples partitioned by head owner, "
                                   + ("rows read in place from IPC-mapped peer shards, all-to-all of gradient sums"
                                      if tr.peer_mapped else "RCCL all-to-all of ids/rows/gradient sums"),
                       "schedule": ("overlapped early fetch" if tr.overlap else
                                    f"equal splits, {tr.capacity} rows per peer and step ({tr.replanned_chunks} chunks re-planned exactly)"
                                    if tr.capacity is not None else "serial, exact splits"),
                       "batch_per_gpu": B, "embedding_dim": d, "table_rows": N,
                       "table_mb_per_gpu": round(rows * d * 4 / 1e6, 1), "parallelism": f"row-shard x{world}",
                       "per_gpu_value": 2.0 * B * K / el, "unique_rows_per_step": stats.unique_rows,
                       "remote_rows_per_step": stats.remote_rows, "early_rows_per_step": stats.early_rows,
                       "a2a_bytes_per_step_per_gpu": stats.bytes_sent,
                       "final_mean_hinge": round(mean_loss, 6)},
            "roofline": {"bound": "hbm", "kernel": "complex_hinge_grad_kernel (shard rows in place + staged rows)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": alg},
            "cpu_baseline": None,
        }
        if emit:
            print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out if rank == 0 else None
