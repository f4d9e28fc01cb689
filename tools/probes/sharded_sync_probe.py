"""How often the HOST waits per exchange-plan chunk of the row-sharded step: the exact-split schedule (per chunk one
read-back of the split sizes) against the equal-split one (ShardedTrainer(capacity="auto"): no split size is read back; one
wait for the plan stream's overflow flag).  Two ranks on this box's one GPU, collectives through gloo; run as

    GE_DIST_BACKEND=gloo GE_SINGLE_DEVICE=1 python tools/probes/sharded_sync_probe.py        (starts its own two ranks)

Counted per steady-state chunk, two ways: (1) the Python-level blocking calls made by sharded.py (Tensor.cpu / .item /
.tolist, Event.synchronize, torch.cuda.synchronize), (2) the runtime calls torch.profiler records on the host thread
(hipMemcpy* device-to-host, hipStreamSynchronize, hipEventSynchronize, hipDeviceSynchronize).  Rank 0 writes
gpurun_out/r04_sharded_host_syncs.json."""
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    if "WORLD_SIZE" not in os.environ:
        from graphembeddings_amd import launch
        os.environ.setdefault("GE_DIST_BACKEND", "gloo")
        os.environ.setdefault("GE_SINGLE_DEVICE", "1")
        sys.exit(launch.spawn_ranks(2, sys.argv[1:], script=os.path.abspath(__file__)))
    import numpy as np
    import torch
    import torch.distributed as dist
    from graphembeddings_amd import hole as H
    from graphembeddings_amd import sharded as S
    from graphembeddings_amd import sharded_train as ST
    rank, world, dev = ST.dist_setup()
    rng = np.random.default_rng(3 + rank)
    N, d, B, S_chunk, n_chunks = 200000, 200, 8192, 8, 6
    tt = H.TypeTables.from_host(np.zeros(N, np.int32), np.array([0, N], np.int64), np.arange(N, dtype=np.int32), padded_size=0, device=dev)
    tri = torch.as_tensor(np.stack([rng.integers(20, N, (n_chunks * S_chunk, B)), rng.integers(20, N, (n_chunks * S_chunk, B)),
                                    rng.integers(0, 20, (n_chunks * S_chunk, B))], 2).astype(np.int32)).to(dev)
    chunks = [tri[c * S_chunk:(c + 1) * S_chunk].contiguous() for c in range(n_chunks)]
    counts = collections.Counter()

    def wrap(obj, name, label):
        orig = getattr(obj, name)

        def f(*a, **k):
            counts[label] += 1
            return orig(*a, **k)
        setattr(obj, name, f)
        return orig
    out = {}
    for schedule, cap in (("exact splits", None), ("equal splits (capacity auto)", "auto")):
        shard = torch.zeros(S.shard_num_rows(N, rank, world), d, device=dev).normal_(0, 0.05)
        tr = S.ShardedTrainer(shard, N, tt, margin=0.2, seed=5, capacity=cap)
        tr.run_pipelined(chunks[:2], lambda gs: 0.1, lookahead=chunks[2])           # warm-up: capacity fixed, look-ahead primed
        torch.cuda.synchronize()
        saved = [(torch.Tensor, "cpu", wrap(torch.Tensor, "cpu", "Tensor.cpu")), (torch.Tensor, "item", wrap(torch.Tensor, "item", "Tensor.item")),
                 (torch.Tensor, "tolist", wrap(torch.Tensor, "tolist", "Tensor.tolist")),
                 (torch.cuda.Event, "synchronize", wrap(torch.cuda.Event, "synchronize", "Event.synchronize")),
                 (torch.cuda, "synchronize", wrap(torch.cuda, "synchronize", "cuda.synchronize"))]
        counts.clear()
        prof_counts = None
        try:
            from torch.profiler import ProfilerActivity, profile
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
                tr.run_pipelined(chunks[2:], lambda gs: 0.1)
            names = collections.Counter(e.name for e in prof.events())
            prof_counts = {k: v for k, v in names.items()
                           if any(t in k for t in ("Synchronize", "hipMemcpy", "cudaMemcpy", "cudaStreamSync", "cudaEventSync"))}
        except Exception as e:                                                       # the profiler is evidence, not the product
            prof_counts = {"error": f"{type(e).__name__}: {e}"}
            tr.run_pipelined(chunks[2:], lambda gs: 0.1)
        py = dict(counts)
        for obj, name, orig in saved:
            setattr(obj, name, orig)
        torch.cuda.synchronize()
        n = len(chunks) - 2
        out[schedule] = {"chunks_timed": n, "steps_per_chunk": S_chunk,
                         "python_level_blocking_calls_per_chunk": {k: v / n for k, v in py.items()},
                         "profiler_runtime_calls_per_chunk": ({k: v / n for k, v in prof_counts.items()} if "error" not in prof_counts else prof_counts),
                         "capacity": tr.capacity, "replanned_chunks": tr.replanned_chunks, "bytes_sent_per_step": tr.stats.bytes_sent}
    if rank == 0:
        out["_what"] = ("host-side waits per exchange-plan chunk, steady state (plan of chunk c+1 built on the side stream while chunk c trains), "
                        "world 2 on one GPU through gloo.  Exact splits: Tensor.cpu = the read-back of the split sizes, which on ONE RCCL "
                        "communicator waits for the training chunk's all-to-alls.  Equal splits: the only wait is Event.synchronize on the "
                        "PLAN stream's overflow flag (computed before any collective of the plan) + one CPU all-reduce of a word.")
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_sharded_host_syncs.json"), "w"), indent=1)
        print(json.dumps(out, indent=1))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
