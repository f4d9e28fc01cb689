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
     (ge_segment_sum_rows: which slots feed which staged row is part of the plan, so the sum is a
     segmented reduction without atomics; only rows with > 32 slots are split and combined atomically),
  5. returns the per-row gradient sums to the owners (all_to_all_single, the reverse of 3), which
     apply them to their shard (ge_segment_sum_rows again: one read-modify-write per distinct row).
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

    def segment_sum_rows(self, src, src_idx, order, begin, length, target, out, accumulate):
        self.h.segment_sum_rows(src, src_idx, order, begin, length, target, out, accumulate)

    native_planner = True   # ShardedTrainer.plan_chunk runs its local stages through ge_plan_* (csrc/ge_plan.hip)


def shard_rows(table: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """This rank's rows of a full table under owner(id) = id % world, local = id // world."""
    return table[rank::world].contiguous()


def shard_num_rows(n_rows: int, rank: int, world: int) -> int:
    return (n_rows - rank + world - 1) // world


@dataclass
class SegmentItems:
    """Work items of ge_segment_sum_rows for a whole chunk; step s owns items item_start[s]:item_start[s+1]."""
    order: torch.Tensor     # int32: source row (step-local) per sorted element, chunk-wide
    begin: torch.Tensor     # int32 [items]: offset into `order`
    length: torch.Tensor    # int32 [items]
    target: torch.Tensor    # int32 [items]: destination row, or ~row when the row is split over items
    item_start: list        # [S+1] host
    split_rows: torch.Tensor  # int64: destination rows that are split (to be zeroed in overwrite mode)
    split_start: list       # [S+1] host


MAX_ITEM = 32   # source rows per work item; longer segments are split and combined atomically


def _seg_of(starts: torch.Tensor, n: int) -> torch.Tensor:
    """Segment index of each of n consecutive elements, given the segments' (nondecreasing, exclusive)
    start offsets -- what repeat_interleave(arange, lengths) returns, by one binary search per element
    (empty segments are skipped correctly: the LAST segment starting at or before the element wins)."""
    return torch.searchsorted(starts, torch.arange(n, device=starts.device), right=True) - 1


def segment_items(cnt: torch.Tensor, seg_bounds: torch.Tensor, seg_row: torch.Tensor, order: torch.Tensor) -> SegmentItems:
    """Cut segments (cnt[i] consecutive elements of `order` each, ordered by step) into work items of
    <= MAX_ITEM elements.  seg_bounds [S+1]: index of each step's first segment (and the total);
    seg_row: destination row per segment.  One host sync (the per-step item ranges)."""
    dev = cnt.device
    n_seg = int(cnt.numel())
    n_it = (cnt + MAX_ITEM - 1) // MAX_ITEM
    split = n_it > 1
    cum_it = torch.cumsum(n_it, 0)
    cum_sp = torch.cumsum(split.to(torch.int64), 0)
    zero = torch.zeros(1, dtype=torch.int64, device=dev)
    host = torch.stack([torch.cat([zero, cum_it])[seg_bounds], torch.cat([zero, cum_sp])[seg_bounds]]).cpu()
    n_items, n_split = int(host[0, -1]), int(host[1, -1])
    seg_off = torch.cumsum(cnt, 0) - cnt
    it_start = cum_it - n_it
    item_seg = _seg_of(it_start, n_items)
    r = torch.arange(n_items, device=dev) - it_start[item_seg]
    begin = seg_off[item_seg] + r * MAX_ITEM
    length = torch.clamp(cnt[item_seg] - r * MAX_ITEM, max=MAX_ITEM)
    row = seg_row[item_seg]
    target = torch.where(split[item_seg], -row - 1, row)
    split_rows = seg_row[split] if n_split else torch.empty(0, dtype=torch.int64, device=dev)
    return SegmentItems(order=order.to(torch.int32).contiguous(), begin=begin.to(torch.int32),
                        length=length.to(torch.int32), target=target.to(torch.int32),
                        item_start=host[0].tolist(), split_rows=split_rows, split_start=host[1].tolist())


def native_segment_items(h, first_pos, n_runs, cap, seg_bounds, order, row_of=None, bucket=None, step_start=None,
                         world=1) -> SegmentItems:
    """segment_items on the ge_plan_* kernels: runs given by first_pos (plan_sorted_runs), cut into items of
    <= MAX_ITEM elements.  seg_bounds [S+1]: index of each step's first run (and the total).  One host sync."""
    n_it, split = h.plan_item_counts(first_pos, n_runs, cap, MAX_ITEM)
    it_incl, sp_incl = torch.cumsum(n_it, 0), torch.cumsum(split, 0)
    zero = torch.zeros(1, dtype=torch.int64, device=first_pos.device)
    host = torch.stack([torch.cat([zero, it_incl])[seg_bounds], torch.cat([zero, sp_incl])[seg_bounds]]).cpu()
    n_items, n_split = int(host[0, -1]), int(host[1, -1])
    begin, length, target, split_rows = h.plan_items(first_pos, n_runs, cap, it_incl, sp_incl, row_of, bucket, step_start,
                                                     world, MAX_ITEM, n_items, n_split)
    return SegmentItems(order=order, begin=begin, length=length, target=target, item_start=host[0].tolist(),
                        split_rows=split_rows, split_start=host[1].tolist())


def _regroup(n: int, counts_src_major: torch.Tensor) -> torch.Tensor:
    """Elements are grouped as segments (a, b) in a-major order with lengths counts_src_major[a, b];
    returns, per element, its position when the same segments are laid out b-major (a segmented
    transpose by index arithmetic -- no sort)."""
    dev = counts_src_major.device
    A, Bn = counts_src_major.shape
    flat = counts_src_major.reshape(-1)
    src_start = torch.cumsum(flat, 0) - flat
    tflat = counts_src_major.t().reshape(-1)
    dst_start = (torch.cumsum(tflat, 0) - tflat).view(Bn, A).t().reshape(-1)   # indexed by a*Bn + b
    seg = _seg_of(src_start, n)
    return dst_start[seg] + (torch.arange(n, device=dev) - src_start[seg])


@dataclass
class ChunkPlan:
    S: int
    B: int
    sc: list           # [S][G] rows this rank requests from each owner
    rc: list           # [S][G] rows each peer requests from this rank
    remap: torch.Tensor     # [S,2B,3] triples re-indexed into the step's staging buffer (-1 invalid)
    req_all: torch.Tensor   # local row indices peers asked of me, ordered (step, peer)
    req_start: list
    reduce_items: SegmentItems = None   # gradient slots -> staged rows (pre-reduction)
    apply_items: SegmentItems = None    # received gradient sums -> shard rows (owner apply)
    ready: object = None                # event recorded on the side stream when the plan was built there
    unique_rows: int = 0
    remote_rows: int = 0


@dataclass
class StepStats:
    unique_rows: int = 0
    remote_rows: int = 0
    bytes_sent: int = 0


class ShardedTrainer:
    def __init__(self, shard: torch.Tensor, n_rows: int, type_tables, *, margin=0.2, model="complex",
                 max_norm=1.0, seed=0, corrupt_mode=0, kernels=None, group=None, plan_group=None):
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
        # run_pipelined builds the NEXT chunk's negatives and exchange plan while the current chunk's steps
        # execute: on a side stream, with the plan's two collectives (request counts, id lists) on their own
        # process group (`plan_group`: its own RCCL communicator, so they do not queue behind the row / gradient
        # all-to-alls of the data path).  Without one the plan shares `group` and still runs on the side stream.
        self.plan_group = plan_group if plan_group is not None else group
        self._side = torch.cuda.Stream(device=shard.device) if shard.is_cuda else None
        self._pending = None    # (positives, first global step, plan) built ahead for the next run_pipelined call

    # -- exchange helpers ---------------------------------------------------------------------
    def _a2a(self, send: torch.Tensor, send_counts, recv_counts, group=None) -> torch.Tensor:
        """all_to_all_single with per-peer row counts (rows of `send` are grouped by destination)."""
        if self.world == 1:
            return send            # the exchange with oneself is the identity: no copy (callers only read the result)
        tail = tuple(send.shape[1:])
        recv = torch.empty((int(sum(recv_counts)),) + tail, dtype=send.dtype, device=send.device)
        dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=list(recv_counts),
                               input_split_sizes=list(send_counts), group=group if group is not None else self.group)
        return recv

    # -- exchange plans -----------------------------------------------------------------------
    # Negatives never depend on the table, so everything about the exchange except the rows and the
    # gradients themselves -- which ids each rank needs, who owns them, the all-to-all split sizes,
    # the re-indexing of the triples into the staging buffer -- is planned for a whole chunk of steps
    # at once: one dedup (torch.unique over step-tagged keys), one all-to-all of counts, one host
    # sync for the split sizes and one all-to-all of id lists per CHUNK instead of per step.
    def plan_chunk(self, pos: torch.Tensor, neg: torch.Tensor) -> "ChunkPlan":
        """pos, neg: [S,B,3] int32 (this rank's positives / negatives for S consecutive steps).
        One sort of the S*6B step-tagged ids gives everything: the staging order (step, owner, id),
        the re-indexed triples, the per-owner request counts and the slot lists per staged row."""
        if getattr(self.k, "native_planner", False) and pos.is_cuda:
            return self._plan_chunk_native(pos, neg)
        G, N, dev = self.world, self.N, pos.device
        S, B = int(pos.shape[0]), int(pos.shape[1])
        M6 = 6 * B
        ids = torch.cat([pos, neg], 1).reshape(S, M6).to(torch.int64)
        valid = (ids >= 0) & (ids < N)
        idz = torch.where(valid, ids, torch.zeros_like(ids))            # invalid ids alias row 0; their slots stay empty
        step = torch.arange(S, device=dev).view(S, 1)
        key = ((step * G + idz % G) * N + idz).reshape(-1)              # sorts as (step, owner, id)
        if S * G * N < 2 ** 31:
            key = key.to(torch.int32)                                   # 4 radix passes instead of 8
        key_sorted, perm = torch.sort(key)
        key_sorted = key_sorted.to(torch.int64)
        uniq, seg_sorted, cnt = torch.unique_consecutive(key_sorted, return_inverse=True, return_counts=True)
        U = int(uniq.numel())
        inverse = torch.empty_like(seg_sorted)
        inverse[perm] = seg_sorted                                      # staged position (chunk-wide) of every slot
        u_id = uniq % N
        bounds = torch.searchsorted(uniq, torch.arange(S * G + 1, device=dev) * N)
        counts = (bounds[1:] - bounds[:-1]).view(S, G)                  # rows I need from owner g at step s
        step_start = bounds[:-1:G]                                      # [S] first staged position of each step
        remap = inverse - step_start.view(S, 1).expand(S, M6).reshape(-1)
        remap = torch.where(valid.reshape(-1), remap, torch.full_like(remap, -1)).to(torch.int32).view(S, 2 * B, 3)
        # counts: what every peer wants from me, per step
        if G > 1:
            send_c = counts.t().contiguous()                            # [G,S]: row p -> peer p
            recv_c = torch.empty_like(send_c)
            dist.all_to_all_single(recv_c, send_c, group=self.plan_group)
            both = torch.stack([counts, recv_c.t()]).cpu()              # the host sync for the split sizes
            sc, rc = both[0], both[1]
        else:
            sc = rc = counts.cpu()
        # id lists grouped by destination peer (then step): one all-to-all for the chunk
        send_ids = (u_id // G).to(torch.int32)
        if G > 1:
            grouped = torch.empty_like(send_ids)
            grouped[_regroup(U, counts)] = send_ids
            send_ids = grouped
        recv_ids = self._a2a(send_ids, sc.sum(0).tolist(), rc.sum(0).tolist(), group=self.plan_group)
        n_req = int(recv_ids.numel())
        rc_dev = rc.to(dev)
        if G > 1:                                                       # received (peer, step) -> needed (step, peer)
            req_all = torch.empty_like(recv_ids)
            req_all[_regroup(n_req, rc_dev.t().contiguous())] = recv_ids
        else:
            req_all = recv_ids
        per_step_req = rc_dev.sum(1)
        req_start_dev = torch.cumsum(per_step_req, 0) - per_step_req
        req_start = [0] + torch.cumsum(rc.sum(1), 0).tolist()
        # pre-reduction items: the gradient slots (ge_hinge_grad order h+,t+,r+,h-,t-,r- per pair)
        # that feed each staged row.  Slots that turn out empty at run time (hinge inactive, merged
        # pos/neg rows, invalid ids) are skipped by the kernel through grad_idx < 0.
        f = torch.arange(M6, device=dev)
        tr, X = f // 3, f % 3
        gslot = (tr % B) * 6 + (tr // B) * 3 + X
        seg_bounds = torch.cat([step_start, torch.full((1,), U, dtype=torch.int64, device=dev)])
        u_step = _seg_of(step_start, U)
        reduce_items = segment_items(cnt, seg_bounds, torch.arange(U, device=dev) - step_start[u_step],
                                     gslot[perm % M6])
        # owner-apply items: the received gradient rows (one per requested (peer,row)) per shard row
        rows_local = int(self.shard.shape[0])
        req_step = _seg_of(req_start_dev, n_req)
        key4 = req_step * rows_local + req_all.to(torch.int64)
        if G > 1:
            key4_sorted, o2 = torch.sort(key4)
        else:                                                           # one peer: each step's list is already sorted
            key4_sorted, o2 = key4, torch.arange(n_req, device=dev)
        useg, cnt4 = torch.unique_consecutive(key4_sorted, return_counts=True)
        bounds4 = torch.searchsorted(useg, torch.arange(S + 1, device=dev) * rows_local)
        apply_items = segment_items(cnt4, bounds4, useg % rows_local, o2 - req_start_dev[req_step[o2]])
        own = int(sc[:, self.rank].sum())
        return ChunkPlan(S=S, B=B, sc=sc.tolist(), rc=rc.tolist(), remap=remap, req_all=req_all,
                         req_start=req_start, reduce_items=reduce_items, apply_items=apply_items,
                         unique_rows=U, remote_rows=U - own)

    def _plan_chunk_native(self, pos: torch.Tensor, neg: torch.Tensor) -> "ChunkPlan":
        """plan_chunk with its local stages on the ge_plan_* kernels (csrc/ge_plan.hip): the same plan, word for
        word (tests/test_gpu_sharded.py), in a dozen launches instead of two hundred.  The device sort, the scans
        and the collectives stay with torch."""
        h = self.k.h
        G, N, dev = self.world, self.N, pos.device
        pos, neg = pos.to(torch.int32).contiguous(), neg.to(torch.int32).contiguous()
        S, B = int(pos.shape[0]), int(pos.shape[1])
        key_sorted, perm = torch.sort(h.plan_keys(pos, neg, N, G))       # (step, owner, id)
        incl, first_pos, bucket, u_id = h.plan_sorted_runs(key_sorted, N)
        n_runs = incl[-1:]
        U = int(n_runs)                                                  # host sync: distinct (step, row) pairs
        bounds = torch.searchsorted(bucket[:U], torch.arange(S * G + 1, dtype=torch.int32, device=dev))
        counts = (bounds[1:] - bounds[:-1]).view(S, G)                  # rows I need from owner g at step s
        step_start = bounds[:-1:G].contiguous()                         # [S] first run of each step
        remap, order = h.plan_scatter(perm, incl, step_start, pos, neg, N)
        if G > 1:
            send_c = counts.t().contiguous()
            recv_c = torch.empty_like(send_c)
            dist.all_to_all_single(recv_c, send_c, group=self.plan_group)
            both = torch.stack([counts, recv_c.t()]).cpu()              # the host sync for the split sizes
            sc, rc = both[0], both[1]
        else:
            sc = rc = counts.cpu()
        send_ids = u_id[:U] // G if G > 1 else u_id[:U]                  # local row at its owner
        if G > 1:
            grouped = torch.empty_like(send_ids)
            grouped[_regroup(U, counts)] = send_ids
            send_ids = grouped
        recv_ids = self._a2a(send_ids, sc.sum(0).tolist(), rc.sum(0).tolist(), group=self.plan_group)
        n_req = int(recv_ids.numel())
        rc_dev = rc.to(dev)
        if G > 1:                                                       # received (peer, step) -> needed (step, peer)
            req_all = torch.empty_like(recv_ids)
            req_all[_regroup(n_req, rc_dev.t().contiguous())] = recv_ids
        else:
            req_all = recv_ids
        per_step_req = rc_dev.sum(1)
        req_start_dev = torch.cumsum(per_step_req, 0) - per_step_req
        req_start = [0] + torch.cumsum(rc.sum(1), 0).tolist()
        seg_bounds = torch.cat([step_start, torch.full((1,), U, dtype=torch.int64, device=dev)])
        reduce_items = native_segment_items(h, first_pos, n_runs, U, seg_bounds, order, bucket=bucket,
                                            step_start=step_start, world=G)
        # owner-apply items: the received gradient rows (one per requested (peer, row)) per shard row
        rows_local = int(self.shard.shape[0])
        req_step = _seg_of(req_start_dev, n_req)
        key4 = req_step * rows_local + req_all.to(torch.int64)
        if S * rows_local < 2 ** 31:
            key4 = key4.to(torch.int32)
        if G > 1:
            key4, o2 = torch.sort(key4)
            order4 = (o2 - req_start_dev[req_step[o2]]).to(torch.int32)
        else:                                                           # one peer: each step's list is already sorted
            order4 = (torch.arange(n_req, device=dev) - req_start_dev[req_step]).to(torch.int32)
        incl4, first4, step4, row4 = h.plan_sorted_runs(key4.contiguous(), rows_local)
        n_runs4 = incl4[-1:]
        U4 = int(n_runs4) if n_req else 0
        bounds4 = torch.searchsorted(step4[:U4], torch.arange(S + 1, dtype=torch.int32, device=dev))
        apply_items = native_segment_items(h, first4, n_runs4, U4, bounds4, order4, row_of=row4)
        own = int(sc[:, self.rank].sum())
        return ChunkPlan(S=S, B=B, sc=sc.tolist(), rc=rc.tolist(), remap=remap, req_all=req_all,
                         req_start=req_start, reduce_items=reduce_items, apply_items=apply_items,
                         unique_rows=U, remote_rows=U - own)

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
        gsum = torch.empty_like(staged)
        ri = plan.reduce_items
        i0, i1 = ri.item_start[s], ri.item_start[s + 1]
        z0, z1 = ri.split_start[s], ri.split_start[s + 1]
        if z1 > z0:
            gsum.index_fill_(0, ri.split_rows[z0:z1], 0.0)              # split rows are combined atomically
        self.k.segment_sum_rows(gv, gi, ri.order, ri.begin[i0:i1], ri.length[i0:i1], ri.target[i0:i1],
                                gsum, False)                            # pre-reduce per staged row
        recv_g = self._a2a(gsum, sc, rc)                                # sums back to the owners
        ai = plan.apply_items
        i0, i1 = ai.item_start[s], ai.item_start[s + 1]
        self.k.segment_sum_rows(recv_g, None, ai.order, ai.begin[i0:i1], ai.length[i0:i1], ai.target[i0:i1],
                                self.shard, True)
        self.global_step += 1
        return loss

    def sample_negatives(self, pos: torch.Tensor, first_step: int = None) -> torch.Tensor:
        """[S,B,3] negatives for S consecutive steps starting at first_step (default: global_step); the Philox
        stream is keyed by a counter that is distinct per (global step, rank)."""
        G = self.world
        s0 = self.global_step if first_step is None else int(first_step)
        return torch.stack([self.k.corrupt_batch(self.tt, pos[s].contiguous(), self.seed,
                                                 (s0 + s) * G + self.rank, self.mode)
                            for s in range(pos.shape[0])], 0)

    def _plan_ahead(self, pos: torch.Tensor, first_step: int, inputs_ready=None) -> "ChunkPlan":
        """Negatives + exchange plan of a chunk that starts at `first_step`, on the side stream.
        inputs_ready: event after which `pos` is valid (default: everything enqueued on the current stream so
        far -- which would put the plan BEHIND steps already enqueued, so run_pipelined passes the event it
        recorded before enqueuing any step)."""
        pos = pos.to(torch.int32).contiguous()
        if self._side is None:
            plan = self.plan_chunk(pos, self.sample_negatives(pos, first_step).to(torch.int32))
            plan.ready = None
            return plan
        if inputs_ready is None:
            self._side.wait_stream(torch.cuda.current_stream(pos.device))
        else:
            self._side.wait_event(inputs_ready)
        with torch.cuda.stream(self._side):
            plan = self.plan_chunk(pos, self.sample_negatives(pos, first_step).to(torch.int32))
            plan.ready = torch.cuda.Event()
            plan.ready.record(self._side)
        return plan

    def _adopt(self, plan: "ChunkPlan") -> None:
        """Make a plan built on the side stream usable on the current one: wait for it, and tell the caching
        allocator that its tensors are read here (they are freed while these reads may still be queued)."""
        if getattr(plan, "ready", None) is None:
            return
        cur = torch.cuda.current_stream(plan.remap.device)
        cur.wait_event(plan.ready)
        for t in (plan.remap, plan.req_all):
            t.record_stream(cur)
        for it in (plan.reduce_items, plan.apply_items):
            for t in (it.order, it.begin, it.length, it.target, it.split_rows):
                if t.is_cuda:
                    t.record_stream(cur)

    def run_pipelined(self, chunks, lr_fn, lookahead: torch.Tensor = None) -> torch.Tensor:
        """Train the chunks (a sequence of pos [S,B,3] tensors) back to back.  The steps of chunk c are
        enqueued first (asynchronously); chunk c+1's negatives and exchange plan -- which never depend on the
        table -- are then built on the side stream while those steps execute, so the host's waits inside the
        planner (its output sizes are data dependent) fall into time the device spends training.
        lookahead: the positives the NEXT call will start with (the same int32 contiguous tensor object, at the
        global step this call ends on): their plan is built while this call's last chunk executes and adopted by
        that call, so a training loop that calls this once per validation tick never plans on the critical path.
        Every rank must pass a lookahead, or none (the plan holds two collectives).
        Returns every step's losses [sum S, B]."""
        given = list(chunks)
        chunks = [c.to(torch.int32).contiguous() for c in given]
        ready = None
        if self._side is not None and (chunks or lookahead is not None):
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.shard.device))   # every chunk's positives exist from here on
        pend, self._pending = getattr(self, "_pending", None), None
        plan = None
        if chunks:
            if pend is not None and pend[0] is given[0] and pend[1] == self.global_step:
                plan = pend[2]                                           # planned during the previous call
            else:
                plan = self._plan_ahead(chunks[0], self.global_step, ready)
        losses = []
        for c in range(len(chunks)):
            self._adopt(plan)
            losses += [self.step_planned(plan, s, lr_fn(self.global_step)) for s in range(plan.S)]
            done = plan
            plan = self._plan_ahead(chunks[c + 1], self.global_step, ready) if c + 1 < len(chunks) else None
            self.stats = StepStats(unique_rows=done.unique_rows // done.S, remote_rows=done.remote_rows // done.S,
                                   bytes_sent=int(done.remote_rows // done.S * (2 * self.d * 4 + 4)))
        if lookahead is not None:
            self._pending = (lookahead, self.global_step, self._plan_ahead(lookahead, self.global_step, ready))
        return torch.stack(losses, 0) if losses else None

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
