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
     (all_to_all_single of counts, then of ids) -- planned for a whole CHUNK of steps at once,
     because negatives never depend on the table (plan_chunk),
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
class ChunkPlan:
    S: int
    B: int
    sc: list           # [S][G] rows this rank requests from each owner
    rc: list           # [S][G] rows each peer requests from this rank
    remap: torch.Tensor     # [S,2B,3] triples re-indexed into the step's staging buffer (-1 invalid)
    req_all: torch.Tensor   # local row indices peers asked of me, ordered (step, peer)
    req_start: list
    unique_rows: int = 0
    remote_rows: int = 0


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

    # -- exchange plans -----------------------------------------------------------------------
    # Negatives never depend on the table, so everything about the exchange except the rows and the
    # gradients themselves -- which ids each rank needs, who owns them, the all-to-all split sizes,
    # the re-indexing of the triples into the staging buffer -- is planned for a whole chunk of steps
    # at once: one dedup (torch.unique over step-tagged keys), one all-to-all of counts, one host
    # sync for the split sizes and one all-to-all of id lists per CHUNK instead of per step.
    def plan_chunk(self, pos: torch.Tensor, neg: torch.Tensor) -> "ChunkPlan":
        """pos, neg: [S,B,3] int32 (this rank's positives / negatives for S consecutive steps)."""
        G, N, dev = self.world, self.N, pos.device
        S, B = int(pos.shape[0]), int(pos.shape[1])
        ids = torch.cat([pos, neg], 1).reshape(S, 6 * B).to(torch.int64)
        valid = (ids >= 0) & (ids < N)
        step = torch.arange(S, device=dev).view(S, 1).expand(S, 6 * B)
        key = (step * N + torch.where(valid, ids, torch.zeros_like(ids))).reshape(-1)
        uniq, inverse = torch.unique(key, return_inverse=True)          # sorted by (step, id)
        u_step, u_id = uniq // N, uniq % N
        u_owner = u_id % G
        k2 = u_step * G + u_owner
        order = torch.argsort(k2, stable=True)                          # staging order (step, owner, id)
        staged_id, staged_owner = u_id[order], u_owner[order]
        counts = torch.bincount(k2, minlength=S * G).view(S, G)         # rows I need from owner g at step s
        pos_in = torch.empty_like(order)
        pos_in[order] = torch.arange(order.numel(), device=dev)
        per_step = counts.sum(1)
        step_start = torch.cumsum(per_step, 0) - per_step
        remap = pos_in[inverse] - step_start[step.reshape(-1)]
        remap = torch.where(valid.reshape(-1), remap, torch.full_like(remap, -1)).to(torch.int32).view(S, 2 * B, 3)
        # counts: what every peer wants from me, per step
        if G > 1:
            send_c = counts.t().contiguous()                            # [G,S]: row p -> peer p
            recv_c = torch.empty_like(send_c)
            dist.all_to_all_single(recv_c, send_c, group=self.group)
            sc = counts.cpu()                                           # the one host sync of the chunk
            rc = recv_c.t().contiguous().cpu()
        else:
            sc = rc = counts.cpu()
        # id lists, grouped by destination peer (then step): one all-to-all for the chunk
        order2 = torch.argsort(staged_owner, stable=True)
        send_ids = (staged_id // G).to(torch.int32)[order2]
        recv_ids = self._a2a(send_ids, sc.sum(0).tolist(), rc.sum(0).tolist())
        # received layout is (peer, step); per-step request lists need (step, peer)
        seg_len = rc.t().contiguous().reshape(-1).to(dev)               # [G*S] lengths in (peer, step) order
        seg = torch.repeat_interleave(torch.arange(G * S, device=dev), seg_len)
        key3 = (seg % S) * G + seg // S
        req_all = recv_ids[torch.argsort(key3, stable=True)]
        return ChunkPlan(S=S, B=B, sc=sc.tolist(), rc=rc.tolist(), remap=remap, req_all=req_all,
                         req_start=[0] + torch.cumsum(rc.sum(1), 0).tolist(),
                         unique_rows=int(per_step.sum()), remote_rows=int(per_step.sum() - counts[:, self.rank].sum()))

    def step_planned(self, plan: "ChunkPlan", s: int, lr: float) -> torch.Tensor:
        """Step s of a planned chunk: fetch rows (all-to-all), fused score/hinge/grad on the staging
        buffer, per-row pre-reduction, gradient sums back to the owners (all-to-all), apply."""
        B = plan.B
        sc, rc = plan.sc[s], plan.rc[s]
        req = plan.req_all[plan.req_start[s]:plan.req_start[s + 1]]
        rows_out = self.k.gather_rows(self.shard, req)                  # owners gather ...
        staged = self._a2a(rows_out, rc, sc)                            # ... rows arrive in staging order
        remap = plan.remap[s]
        loss, gi, gv = self.k.hinge_grad(staged, remap[:B].contiguous(), remap[B:].contiguous(), lr,
                                         self.margin, self.model, self.max_norm)
        gsum = torch.zeros_like(staged)
        self.k.scatter_add_rows(gsum, gi, gv)                           # pre-reduce per staged row
        recv_g = self._a2a(gsum, sc, rc)                                # sums back to the owners
        self.k.scatter_add_rows(self.shard, req, recv_g)
        self.global_step += 1
        return loss

    def sample_negatives(self, pos: torch.Tensor) -> torch.Tensor:
        """[S,B,3] negatives for S consecutive steps starting at global_step; the Philox stream is keyed
        by a counter that is distinct per (global step, rank)."""
        G = self.world
        return torch.stack([self.k.corrupt_batch(self.tt, pos[s].contiguous(), self.seed,
                                                 (self.global_step + s) * G + self.rank, self.mode)
                            for s in range(pos.shape[0])], 0)

    def run(self, pos: torch.Tensor, lr_fn, neg: torch.Tensor = None) -> torch.Tensor:
        """Train S consecutive steps on pos [S,B,3]; lr_fn(global_step) -> lr.  Returns losses [S,B]."""
        pos = pos.to(torch.int32).contiguous()
        if neg is None:
            neg = self.sample_negatives(pos)
        plan = self.plan_chunk(pos, neg.to(torch.int32))
        losses = [self.step_planned(plan, s, lr_fn(self.global_step)) for s in range(plan.S)]
        self.stats = StepStats(unique_rows=plan.unique_rows // plan.S, remote_rows=plan.remote_rows // plan.S,
                               bytes_sent=int(plan.remote_rows // plan.S * (2 * self.d * 4 + 4)))
        return torch.stack(losses, 0)

    def step(self, pos: torch.Tensor, lr: float, neg: torch.Tensor = None) -> torch.Tensor:
        """One training step on this rank's positives [B,3]; returns the local hinge [B]."""
        return self.run(pos.unsqueeze(0), lambda _gs: lr, None if neg is None else neg.unsqueeze(0))[0]

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
