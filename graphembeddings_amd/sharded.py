"""Row-sharded multi-GPU training step (SURVEY.md 8e / BASELINE config 4): one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm).

The reference is single-device (no tf.device / NCCL anywhere); what has to be kept is
the RESULT of one step of holE.py:287-296 on the global batch: every gradient evaluated against the
table as it was before the step, then table[row] -= lr * grad for every occurrence.

Layout: table rows are mod-sharded, owner(id) = id % G, local row = id // G (balances Zipfian
heads), each rank holds a [ceil(N/G), d] fp32 shard in its HBM.  Triples shard by rank (independent
units); type tables are replicated (small).  Per step each rank
  1. corrupts its B_loc positives (ge_corrupt_batch; step counter offset by rank so streams differ),
  2. dedups the row ids it needs, buckets them by owner, and exchanges the id lists
     (all_to_all_single of counts, then of ids),
  3. owners gather the requested rows from their shard (ge_gather_rows) and send them back
     (all_to_all_single, <= 4*B_loc*d*4 bytes per rank, spread over all 7 xGMI peers at once),
  4. runs the fused gather->score->hinge->grad kernel on the staged rows (ge_hinge_grad with pos/neg
     re-indexed into the staging buffer) and pre-reduces the IndexedSlices per staged row
     (ge_scatter_add_rows into a zeroed staging-shaped buffer),
  5. returns the per-row gradient sums to the owners (all_to_all_single, the reverse of 3), which
     apply them to their shard (ge_scatter_add_rows).
xGMI is point-to-point: the all-to-all drives all peer links concurrently, which is why the table
is never all-reduced.  The only other collective is the optional scalar loss all-reduce for logging.

The kernels are injected (`kernels=`): the product default is HipKernels (the C-ABI HIP path; it
raises without a GPU).  tests/ inject an oracle-backed double to cover the exchange logic with
world_size-2 gloo processes on CPU.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist


class HipKernels:
    """The product kernels: libge_hip.so through graphembeddings_amd.hole."""

    def __init__(self):
        from . import hole
        self.h = hole

    def corrupt_batch(self, tt, pos, seed, step, mode):
        return self.h.corrupt_batch(tt, 0, pos, seed=seed, step=step, mode=mode)

    def gather_rows(self, table, idx):
        return self.h.gather_rows(table, idx)

    def hinge_grad(self, rows, pos, neg, lr, margin, model, max_norm):
        return self.h.hinge_grad(rows, pos, neg, lr, margin=margin, model=model, max_norm=max_norm)

    def scatter_add_rows(self, table, idx, val):
        self.h.scatter_add_rows(table, idx, val)


def shard_rows(table: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """This rank's rows of a full table under owner(id) = id % world, local = id // world."""
    return table[rank::world].contiguous()


def shard_num_rows(n_rows: int, rank: int, world: int) -> int:
    return (n_rows - rank + world - 1) // world


@dataclass
class StepStats:
    unique_rows: int = 0
    remote_rows: int = 0
    bytes_sent: int = 0


class ShardedTrainer:
    def __init__(self, shard: torch.Tensor, n_rows: int, type_tables, *, margin=0.2, model="complex",
                 max_norm=1.0, seed=0, corrupt_mode=0, kernels=None, group=None):
        self.shard = shard
        self.N = int(n_rows)
        self.d = int(shard.shape[1])
        self.tt = type_tables
        self.margin, self.model, self.max_norm = float(margin), model, float(max_norm)
        self.seed, self.mode = int(seed), int(corrupt_mode)
        self.k = kernels if kernels is not None else HipKernels()
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        assert shard.shape[0] == shard_num_rows(self.N, self.rank, self.world)
        self.global_step = 0
        self.stats = StepStats()

    # -- exchange helpers ---------------------------------------------------------------------
    def _a2a(self, send: torch.Tensor, send_counts, recv_counts) -> torch.Tensor:
        """all_to_all_single with per-peer row counts (rows of `send` are grouped by destination)."""
        tail = tuple(send.shape[1:])
        recv = torch.empty((int(sum(recv_counts)),) + tail, dtype=send.dtype, device=send.device)
        if self.world == 1:
            recv.copy_(send)
            return recv
        dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=list(recv_counts),
                               input_split_sizes=list(send_counts), group=self.group)
        return recv

    def _fetch(self, ids: torch.Tensor):
        """Dedup the requested ids, bucket by owner, exchange id lists and fetch the rows.
        Returns (staged [U,d], remap: position of each requested id in `staged` or -1,
        req: local rows peers asked of me, sc/rc: per-peer counts)."""
        G, dev = self.world, ids.device
        ids = ids.reshape(-1).to(torch.int64)
        valid = (ids >= 0) & (ids < self.N)
        uniq, inverse = torch.unique(torch.where(valid, ids, torch.zeros_like(ids)), return_inverse=True)
        owner = uniq % G
        order = torch.argsort(owner, stable=True)           # staging order: grouped by owner
        staged_ids = uniq[order]
        send_counts = torch.bincount(owner, minlength=G)
        if G > 1:
            recv_counts = torch.empty_like(send_counts)
            dist.all_to_all_single(recv_counts, send_counts, group=self.group)
            sc, rc = send_counts.tolist(), recv_counts.tolist()   # host sync: split sizes
        else:
            sc = rc = send_counts.tolist()
        req = self._a2a((staged_ids // G).to(torch.int32), sc, rc)   # id lists to owners
        rows_out = self.k.gather_rows(self.shard, req)                # owners gather ...
        staged = self._a2a(rows_out, rc, sc)                          # ... and return rows, staging order
        pos_in_stage = torch.empty_like(order)
        pos_in_stage[order] = torch.arange(order.numel(), device=dev)
        remap = torch.where(valid, pos_in_stage[inverse], torch.full_like(inverse, -1)).to(torch.int32)
        self.stats = StepStats(unique_rows=int(uniq.numel()), remote_rows=int(uniq.numel() - sc[self.rank]),
                               bytes_sent=int((uniq.numel() - sc[self.rank]) * (2 * self.d * 4 + 4)))
        return staged, remap, req, sc, rc

    def step(self, pos: torch.Tensor, lr: float, neg: torch.Tensor = None) -> torch.Tensor:
        """One training step on this rank's positives [B_loc,3]; returns the local hinge [B_loc]."""
        G = self.world
        pos = pos.to(torch.int32).contiguous()
        if neg is None:
            # distinct counter per (global step, rank): keys the Philox stream of the sampler
            neg = self.k.corrupt_batch(self.tt, pos, self.seed, self.global_step * G + self.rank, self.mode)
        B = pos.shape[0]
        staged, remap, req, sc, rc = self._fetch(torch.cat([pos, neg], 0))
        remap = remap.view(2 * B, 3)   # triples re-indexed into the staging buffer (-1 stays invalid)
        # fused score/hinge/grad on staged rows, then pre-reduce the IndexedSlices per staged row
        loss, gi, gv = self.k.hinge_grad(staged, remap[:B].contiguous(), remap[B:].contiguous(), lr,
                                         self.margin, self.model, self.max_norm)
        gsum = torch.zeros_like(staged)
        self.k.scatter_add_rows(gsum, gi, gv)
        # gradient sums back to the owners, applied to the shard
        recv_g = self._a2a(gsum, sc, rc)
        self.k.scatter_add_rows(self.shard, req, recv_g)
        self.global_step += 1
        return loss

    def mean_loss(self, loss: torch.Tensor) -> float:
        """Scalar all-reduce (logging only)."""
        t = torch.stack([loss.sum(), torch.tensor(float(loss.numel()), device=loss.device)])
        if self.world > 1:
            dist.all_reduce(t, group=self.group)
        return float(t[0] / t[1])

    def gather_full_table(self) -> torch.Tensor:
        """All-gather the shards back into the [N,d] table (checkpointing / tests)."""
        G = self.world
        if G == 1:
            return self.shard.clone()
        rows = (self.N + G - 1) // G
        pad = torch.zeros(rows, self.d, dtype=self.shard.dtype, device=self.shard.device)
        pad[: self.shard.shape[0]] = self.shard
        parts = [torch.empty_like(pad) for _ in range(G)]
        dist.all_gather(parts, pad, group=self.group)
        full = torch.empty(rows * G, self.d, dtype=self.shard.dtype, device=self.shard.device)
        for g in range(G):
            full[g::G] = parts[g]
        return full[: self.N]
