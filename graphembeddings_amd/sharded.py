"""Row-sharded multi-GPU training step (SURVEY.md 8e / BASELINE config 4): one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm).

The reference is single-device (no tf.device / NCCL anywhere); what has to be kept is
the RESULT of one step of holE.py:287-296 on the global batch: every gradient evaluated against the
table as it was before the step, then table[row] -= lr * grad for every occurrence.

Layout: table rows are mod-sharded, owner(id) = id % G, local row = id // G (balances Zipfian
heads), each rank holds a [ceil(N/G), d] fp32 shard in its HBM.  Triples shard by rank (independent
units); type tables are replicated (small).  Negatives never depend on the table, so everything
about a step except its floating-point work is PLANNED for a whole chunk of steps at once
(plan_chunk; natively: csrc/ge_shard.hip):
  * requester side: the step's gradient slots sorted by (own rows first, then owner, row).  Own
    rows are read and updated IN PLACE, exactly like in the one-GPU loop (the same work items: one
    read-modify-write per distinct row, sole-slot rows by the producing pair).  The distinct rows
    of other owners, in sorted order, are the step's staging order: the request list sent to the
    owners (all_to_all_single of counts, then of ids, once per chunk), the order the fetched rows
    arrive in and the order the gradient sums go back in.
  * owner side: the received request lists sorted by row: work items that add the returned
    gradient sums to the shard, again one read-modify-write per distinct row.
Per step each rank then
  1. gathers the rows its peers asked for (ge_gather_rows) and sends them (all_to_all_single,
     spread over all 7 xGMI peers at once); nothing is gathered or copied for its own rows,
  2. runs the fused gather->score->hinge->grad kernel on two row stores -- its shard and the staging
     buffer (ge_shard_grad) -- then ONE kernel that updates its own rows and reduces the gradient
     rows of every staged row into the send buffer (ge_shard_apply),
  3. returns the per-row gradient sums to the owners (all_to_all_single, the reverse of 1), which
     add them to their shard (ge_shard_owner_apply).
At world size 1 steps 1 and 3 vanish and the step IS the one-GPU native step (two kernels).
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


@dataclass
class RequesterPlan:
    """What a kernel backend's plan_requester returns for S steps.  The trainer reads `counts` and `req_row`
    (the exchange) and hands the rest back to the backend's grad / apply."""
    S: int
    B: int
    counts: torch.Tensor       # int32 [S,G]: distinct rows needed from each owner; column `rank` = distinct own rows
    req_row: torch.Tensor      # int32 [S,cap]: entry u = staged row u's index in its owner's shard (grouped by owner)
    data: object = None        # backend-private (records, row sources ...)


class HipKernels:
    """The product kernels: libge_hip.so through graphembeddings_amd.hole."""

    def __init__(self):
        from . import hole
        self.h = hole
        self._grad_ws = None

    def corrupt_batch(self, tt, pos, seed, step, mode):
        return self.h.corrupt_batch(tt, 0, pos, seed=seed, step=step, mode=mode)

    def gather_rows(self, table, idx):
        return self.h.gather_rows(table, idx)

    peer_shards = None     # set by ShardedTrainer(peer_mapped=True): the shards of all ranks, mapped into this process

    def plan_requester(self, pos, neg, n_rows, world, rank) -> RequesterPlan:
        records, pos_src, neg_src, req_row, counts = self.h.shard_plan(pos, neg, n_rows, world, rank,
                                                                         peer_mapped=self.peer_shards is not None)
        return RequesterPlan(S=int(pos.shape[0]), B=int(pos.shape[1]), counts=counts, req_row=req_row,
                             data=(records, pos_src, neg_src, n_rows, world))

    def plan_owner(self, req_all, req_start, rows_local):
        return self.h.shard_owner_plan(req_all, req_start, rows_local)

    def _workspace(self, B, d, dev):
        if self._grad_ws is None or self._grad_ws[0].numel() < 6 * B or self._grad_ws[1].shape[1] != d:
            self._grad_ws = (torch.empty(6 * B, dtype=torch.int32, device=dev),
                             torch.empty(6 * B, d, dtype=torch.float32, device=dev))
        return self._grad_ws

    def _step_ptrs(self, shard, plan):
        """Addresses of a plan's per-step arrays, taken once per plan: the step loop is host time (5 launches and two
        collectives a step against ~0.1-0.3 ms of device work), and three tensor-indexing dispatches per launch were a
        third of it."""
        c = getattr(plan, "_ptrs", None)
        if c is None or c[0] is not shard or c[15] is not self._grad_ws:
            records, pos_src, neg_src, n_rows, world = plan.data
            self.h._table(shard)
            for t in (records, pos_src, neg_src):
                if t.dtype != torch.int32 or not t.is_contiguous():
                    raise ValueError("plan arrays must be contiguous int32")
            gi, gv = self._workspace(plan.B, shard.shape[1], shard.device)
            c = plan._ptrs = (shard, shard.data_ptr(), int(shard.shape[0]), int(shard.shape[1]),
                              records.data_ptr(), 4 * records.stride(0), pos_src.data_ptr(), 4 * pos_src.stride(0),
                              neg_src.data_ptr(), 4 * neg_src.stride(0), int(n_rows), int(world), gi.data_ptr(), gv.data_ptr(),
                              self.h._MODELS, self._grad_ws)
        return c

    def grad(self, shard, staged, plan, s, lr, margin, model, max_norm, gsum):
        if self.peer_shards is not None:
            records, pos_src, neg_src, n_rows, world = plan.data
            gi, gv = self._workspace(plan.B, shard.shape[1], shard.device)
            return self.h.shard_grad(shard, staged, pos_src[s], neg_src[s], records[s], plan.B, n_rows, world, lr, margin, model,
                                     max_norm, gi, gv, gsum, peer_shards=self.peer_shards)
        (_, sp, rows, d, rec, rec_st, ps, ps_st, ns, ns_st, n_rows, world, gi, gv, models, _ws) = self._step_ptrs(shard, plan)
        n_staged = 0 if staged is None else int(staged.shape[0])
        loss = torch.empty(plan.B, dtype=torch.float32, device=shard.device)
        self.h._lib.call("ge_shard_grad", sp, rows, d, staged.data_ptr() if n_staged else None, n_staged, ps + s * ps_st,
                         ns + s * ns_st, rec + s * rec_st, plan.B, n_rows, world, float(margin), float(lr), float(max_norm),
                         models[model], loss.data_ptr(), gi, gv, gsum.data_ptr() if n_staged else None, None, self.h._stream())
        return loss

    def apply(self, shard, plan, s, gsum):
        (_, sp, rows, d, rec, rec_st, _ps, _pst, _ns, _nst, n_rows, world, gi, gv, _m, _ws) = self._step_ptrs(shard, plan)
        self.h._lib.call("ge_shard_apply", sp, rows, d, rec + s * rec_st, plan.B, n_rows, world, gi, gv,
                         gsum.data_ptr() if gsum is not None and gsum.numel() else None, self.h._stream())

    def owner_apply(self, shard, oplan, s, recv):
        self.h.shard_owner_apply(shard, oplan, s, recv)

    def plan_tensors(self, plan, oplan):
        """Device tensors of a plan built on another stream (for record_stream)."""
        out = [plan.counts, plan.req_row, plan.data[0], plan.data[1], plan.data[2]]
        if oplan is not None:
            out.append(oplan[0])
        return out

    def to_spectral(self, shard):
        return self.h.hole_to_spectral(shard)

    def from_spectral(self, shard):
        return self.h.hole_from_spectral(shard)


class _NullContext:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def shard_rows(table: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """This rank's rows of a full table under owner(id) = id % world, local = id // world."""
    return table[rank::world].contiguous()


def shard_num_rows(n_rows: int, rank: int, world: int) -> int:
    return (n_rows - rank + world - 1) // world


def _seg_of(starts: torch.Tensor, n: int) -> torch.Tensor:
    """Segment index of each of n consecutive elements, given the segments' (nondecreasing, exclusive)
    start offsets -- what repeat_interleave(arange, lengths) returns, by one binary search per element
    (empty segments are skipped correctly: the LAST segment starting at or before the element wins)."""
    return torch.searchsorted(starts, torch.arange(n, device=starts.device), right=True) - 1


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
    sc: list           # [S][G] rows this rank requests from each owner (0 for itself)
    rc: list           # [S][G] rows each peer requests from this rank
    req: RequesterPlan = None
    req_all: torch.Tensor = None   # local row indices peers asked of me, ordered (step, peer)
    req_start: list = None
    owner: object = None           # backend-private owner-side plan (None at world size 1)
    pre: FetchSplit = None         # overlapped schedule only
    static: "StaticSplit" = None   # equal-split schedule only (ShardedTrainer(capacity=...))
    ready: object = None           # event recorded on the side stream when the plan was built there
    unique_rows: int = 0
    remote_rows: int = 0
    own_rows_dev: torch.Tensor = None   # world size 1: the chunk's distinct rows, still on the device (statistics, read lazily)

    def resolve(self) -> "ChunkPlan":
        """Statistics that were left on the device so that planning never made the host wait (a blocking read)."""
        if self.own_rows_dev is not None:
            self.unique_rows, self.own_rows_dev = int(self.own_rows_dev.item()), None
        return self


@dataclass
class FetchSplit:
    """The overlapped schedule's split of every step's fetch (ShardedTrainer(overlap=True)): EARLY rows are those no
    rank touches in the previous step (not an own row referenced there, not requested there), so their owner can
    gather and send them while that step still computes; LATE rows wait for its updates.  Requester side: the staging
    positions u of each kind; owner side: the request lists of each kind; split sizes per (step, peer)."""
    sc_e: list
    sc_l: list
    rc_e: list
    rc_l: list
    idx_early: torch.Tensor        # int64, step-major: staging positions filled by the early all-to-all
    idx_late: torch.Tensor
    ie_start: list                 # [S+1] host offsets into idx_early / idx_late
    il_start: list
    req_early: torch.Tensor        # int32, (step, peer) order: rows gathered for the early all-to-all
    req_late: torch.Tensor
    re_start: list
    rl_start: list
    staged: dict = None            # step -> staging buffer that the early all-to-all of that step filled
    done: dict = None              # step -> event recorded behind it on the communication stream


@dataclass
class StaticSplit:
    """The equal-split schedule (ShardedTrainer(capacity=C)): every all-to-all moves exactly C rows per peer and step, so
    no split size -- nothing data dependent -- ever has to reach the host.  Requester side: where staged row u sits in the
    padded receive buffer (stage_index) and where the padded send-back buffer takes its rows from (back_index; unused
    slots read the one all-zero row behind the gradient sums); the only read-back is the overflow flag, copied to pinned
    memory on the stream the plan was built on and looked at when the plan is adopted."""
    C: int
    stage_index: torch.Tensor      # int64 [S, G*C]: padded receive position of staged row u (u >= the step's count: 0)
    back_index: torch.Tensor       # int64 [S, G*C]: staged row sent back in padded slot (p, j); G*C = the zero row
    over_host: torch.Tensor        # pinned int32 [1]: 1 when some (step, owner) needs more than C rows
    over_ready: object = None      # event behind the copy into over_host
    used_rows: torch.Tensor = None  # int64 [1] on the device: rows really requested in the chunk (statistics, read lazily)


def _segment_sums(flags: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
    """Sum of `flags` over consecutive segments whose lengths are `lengths` (any shape, row-major)."""
    c = torch.cat([torch.zeros(1, dtype=torch.int64, device=flags.device), torch.cumsum(flags.to(torch.int64), 0)])
    ends = torch.cumsum(lengths.reshape(-1), 0)
    return (c[ends] - c[ends - lengths.reshape(-1)]).view(lengths.shape)


@dataclass
class StepStats:
    unique_rows: int = 0
    remote_rows: int = 0
    bytes_sent: int = 0
    early_rows: int = 0        # overlapped schedule: fetched rows per step that travelled beside the previous step


class ShardedTrainer:
    def __init__(self, shard: torch.Tensor, n_rows: int, type_tables, *, margin=0.2, model="complex",
                 max_norm=1.0, seed=0, corrupt_mode=0, kernels=None, group=None, plan_group=None, peer_mapped=False,
                 overlap=False, capacity=None, capacity_margin=1.25, control_group=None):
        """peer_mapped (EXPERIMENT, one node, world <= 8, HipKernels): every rank maps the other ranks' shards into
        its address space (CUDA IPC handles exchanged once) and the gradient kernel reads the other owners' rows IN
        PLACE over the fabric: no gather, no row all-to-all, no staging buffer.  What it costs instead: two
        stream-ordered cross-rank barriers per step (nobody may update rows another rank is still reading; nobody may
        read rows whose owner has not finished the previous step's updates).  Gradient sums still go back by
        all-to-all.  Rehearsed with the ranks sharing one device; on real peers it additionally needs peer access
        between the devices to be enabled, which this code does not do: do not enable it on hardware unverified.

        overlap (opt-in, not yet run on RCCL): the rows of step s+1 that NO rank touches in step s -- the plan knows
        them -- are gathered and sent on a communication stream while step s computes; only the remainder is fetched
        in front of step s+1.  Same arithmetic, bitwise the same table as the serial schedule
        (tests/test_gpu_sharded.py); one more flag exchange and one more size read-back per chunk.  All collectives
        stay on ONE communicator and are issued in the same host order on every rank.

        capacity (opt-in; "auto" or rows per peer and step): the EQUAL-SPLIT schedule.  The exact schedule sizes every
        all-to-all from the plan's per-peer counts, which the host reads back once per chunk -- and on one communicator
        that read-back waits behind the all-to-alls of the chunk that is training, so the device idles once per chunk
        while the host catches up.  With a static capacity C every all-to-all moves C rows per peer (unused slots: id -1,
        zero rows), nothing data dependent reaches the host and the whole run is enqueued ahead of the device; the same
        rows meet in the same order, so tables and losses are BITWISE the exact schedule's.  "auto": the first chunk is
        planned exactly and C = capacity_margin x its largest per-peer count (agreed across ranks, rounded up to 64).  A
        chunk that needs more than C somewhere is re-planned exactly: the overflow flag is computed before any of the
        plan's collectives, read from pinned memory (a wait for the PLAN's stream, never the training stream) and
        agreed by a one-word all-reduce on `control_group` -- a CPU (gloo) group, created here when none is given, so
        the agreement never queues on the GPU communicator.  Padded bytes are what `stats.bytes_sent` reports."""
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
        self.plan_depth = 2            # chunks planned ahead of the steps by run_pipelined (1: the schedule of rounds 2-3)
        # model="hole" (a real-valued HolE table): the shard is carried in the frequency domain, where HolE is the
        # ComplEx-shaped trilinear form (csrc/ge_complex_dev.h); gather_full_table() returns real rows again
        self._spectral_resident = False
        if model == "hole" and hasattr(self.k, "to_spectral"):
            if self.d & 1:
                raise ValueError("the row-sharded HolE step needs an even embedding_dim")
            self.k.to_spectral(self.shard)
            self.model, self._spectral_resident = "hole_spectral", True
        # run_pipelined builds the NEXT chunk's negatives and exchange plan while the current chunk's steps
        # execute: on a side stream, with the plan's two collectives (request counts, id lists) on their own
        # process group (`plan_group`: its own RCCL communicator, so they do not queue behind the row / gradient
        # all-to-alls of the data path).  Without one the plan shares `group`: a communicator runs its collectives
        # in issue order, so the look-ahead plan's two exchanges then wait behind the all-to-alls of the chunk that
        # is training (the host blocks in the plan's size read-back until that chunk is done).
        self.plan_group = plan_group if plan_group is not None else group
        self.peer_mapped = bool(peer_mapped) and self.world > 1
        self.overlap = bool(overlap) and self.world > 1 and not self.peer_mapped
        self._comm = torch.cuda.Stream(device=shard.device) if (self.overlap and shard.is_cuda) else None
        if self.peer_mapped:
            self._map_peer_shards()
        self._side = torch.cuda.Stream(device=shard.device) if shard.is_cuda else None
        self._pending = None    # [(positives, first global step, plan)] built ahead for the next run_pipelined call
        self.capacity = None                       # rows per peer and step of the equal-split schedule (None: exact splits)
        self._capacity_auto = False
        self.capacity_margin = float(capacity_margin)
        self.control_group = control_group
        self.replanned_chunks = 0                  # chunks that overflowed the capacity and ran on exact splits
        if capacity is not None and self.world > 1 and not self.peer_mapped and not self.overlap:
            if capacity == "auto":
                self._capacity_auto = True
            else:
                self.capacity = int(capacity)
            if self.control_group is None:
                # the agreement on "did anyone overflow" runs on the CPU: it must not queue behind GPU collectives
                self.control_group = dist.new_group(backend="gloo") if dist.get_backend(group) != "gloo" else group

    def _map_peer_shards(self):
        """Exchange CUDA IPC handles of the shards (torch's own reduction: hipIpcGetMemHandle / OpenMemHandle) and keep
        the mapped tensors alive; the kernel backend gets their addresses."""
        from torch.multiprocessing.reductions import reduce_tensor
        if self.world > 8 or not hasattr(self.k, "peer_shards"):
            raise ValueError("peer_mapped needs world <= 8 and the HIP kernel backend")
        fn, args = reduce_tensor(self.shard)
        box = [None] * self.world
        dist.all_gather_object(box, (fn, args), group=self.group)
        self._peer_tensors = [self.shard if r == self.rank else box[r][0](*box[r][1]) for r in range(self.world)]
        self.k.peer_shards = self._peer_tensors
        self._sync_word = torch.zeros(1, dtype=torch.float32, device=self.shard.device)

    def _rank_barrier(self):
        """Stream-ordered barrier across the ranks: a one-word all-reduce (it completes on a rank only after every
        rank has reached it on ITS stream; the host is not blocked)."""
        dist.all_reduce(self._sync_word, group=self.group)

    # -- exchange helpers ---------------------------------------------------------------------
    def _a2a(self, send: torch.Tensor, send_counts, recv_counts, group=None) -> torch.Tensor:
        """all_to_all_single with per-peer row counts (rows of `send` are grouped by destination)."""
        tail = tuple(send.shape[1:])
        recv = torch.empty((int(sum(recv_counts)),) + tail, dtype=send.dtype, device=send.device)
        dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=list(recv_counts),
                               input_split_sizes=list(send_counts), group=group if group is not None else self.group)
        return recv

    # -- exchange plans -----------------------------------------------------------------------
    def plan_chunk(self, pos: torch.Tensor, neg: torch.Tensor, exact: bool = False) -> "ChunkPlan":
        """pos, neg: [S,B,3] int32 (this rank's positives / negatives for S consecutive steps).  One sort of each
        step's gradient slots by (own rows, then owner, row) -- the backend's plan_requester -- gives the work items
        and the staging order; the request lists are exchanged once for the whole chunk (all_to_all_single of the
        per-(step, owner) counts, one host read-back for the split sizes, all_to_all_single of the id lists) and
        sorted by row at the owner (plan_owner)."""
        G, dev = self.world, pos.device
        pos, neg = pos.to(torch.int32).contiguous(), neg.to(torch.int32).contiguous()
        S, B = int(pos.shape[0]), int(pos.shape[1])
        rp = self.k.plan_requester(pos, neg, self.N, G, self.rank)
        counts = rp.counts.to(torch.int64)                              # [S,G]; column `rank` = distinct own rows
        if G > 1 and self.capacity is not None and not exact:
            return self._plan_chunk_static(rp, counts, S, B)
        if G == 1:
            # nothing data dependent reaches the host: the distinct-row count is statistics, read when somebody asks
            # (a read-back here made the host wait for the whole chunk's plan kernels before it could enqueue a step)
            zeros = [[0]] * S
            return ChunkPlan(S=S, B=B, sc=zeros, rc=zeros, req=rp, req_start=[0] * (S + 1), unique_rows=0, remote_rows=0,
                             own_rows_dev=counts.sum().view(1))
        ask = counts.clone()
        ask[:, self.rank] = 0
        send_c = ask.t().contiguous()                                   # [G,S]: row p -> peer p
        recv_c = torch.empty_like(send_c)
        dist.all_to_all_single(recv_c, send_c, group=self.plan_group)
        host = torch.stack([counts, ask, recv_c.t()]).cpu()             # the host sync for the split sizes
        own, sc, rc = int(host[0][:, self.rank].sum()), host[1], host[2]
        # request lists: step-major (owner, row) runs -> grouped by destination peer (then step): one all-to-all
        U = sc.sum(1)
        n = int(U.sum())
        cap = int(rp.req_row.shape[1])
        mask = torch.arange(cap, device=dev).view(1, -1) < U.to(dev).view(-1, 1)
        flat = rp.req_row[mask]
        send_ids = torch.empty_like(flat)
        rg = _regroup(n, ask) if n else None                            # flat position -> position in send order
        if n:
            send_ids[rg] = flat
        recv_ids = self._a2a(send_ids, sc.sum(0).tolist(), rc.sum(0).tolist(), group=self.plan_group)
        n_req = int(recv_ids.numel())
        req_all = torch.empty_like(recv_ids)                            # received (peer, step) -> needed (step, peer)
        rc_dev = rc.to(dev)
        rmap = _regroup(n_req, rc_dev.t().contiguous()) if n_req else None   # receive position -> position in req_all
        if n_req:
            req_all[rmap] = recv_ids
        req_start = [0] + torch.cumsum(rc.sum(1), 0).tolist()
        owner = self.k.plan_owner(req_all, req_start, int(self.shard.shape[0]))
        pre = self._split_fetch(pos, neg, ask, sc, rc, rc_dev, rg, rmap, req_all, n, n_req) if self.overlap else None
        return ChunkPlan(S=S, B=B, sc=sc.tolist(), rc=rc.tolist(), req=rp, req_all=req_all, req_start=req_start,
                         owner=owner, pre=pre, unique_rows=own + n, remote_rows=n)

    def _plan_chunk_static(self, rp: RequesterPlan, counts: torch.Tensor, S: int, B: int) -> "ChunkPlan":
        """The equal-split plan: no host read-back except the overflow flag (see StaticSplit).  Index arithmetic on the
        device only, static shapes throughout."""
        G, C, dev = self.world, int(self.capacity), counts.device
        ask = counts.clone()
        ask[:, self.rank] = 0                                            # [S,G] rows wanted from each owner
        over = (ask > C).any().to(torch.int32).view(1)
        over_host = torch.empty(1, dtype=torch.int32, pin_memory=True) if dev.type == "cuda" else torch.empty(1, dtype=torch.int32)
        over_host.copy_(over, non_blocking=True)
        over_ready = None
        if dev.type == "cuda":
            over_ready = torch.cuda.Event()
            over_ready.record(torch.cuda.current_stream(dev))
        ends = torch.cumsum(ask, 1)                                      # staging order: runs of owner 0, 1, ...
        off = ends - ask
        cap_r = int(rp.req_row.shape[1])
        j = torch.arange(C, device=dev).view(1, 1, C)
        u = (off.unsqueeze(2) + j).clamp_(max=cap_r - 1)                 # [S,G,C] staged row of padded slot (p, j)
        valid = j < ask.unsqueeze(2)
        ids = torch.where(valid, rp.req_row.gather(1, u.view(S, G * C)).view(S, G, C).to(torch.int32),
                          torch.full((), -1, dtype=torch.int32, device=dev))
        send_ids = ids.permute(1, 0, 2).contiguous()                     # block p -> peer p
        recv_ids = torch.empty_like(send_ids)
        dist.all_to_all_single(recv_ids, send_ids, group=self.plan_group)
        req_all = recv_ids.permute(1, 0, 2).reshape(-1).contiguous()     # per step: peer-major, C slots per peer
        req_start = [s * G * C for s in range(S + 1)]
        owner = self.k.plan_owner(req_all, req_start, int(self.shard.shape[0]))
        # staged row u of a step sits at padded position owner(u) * C + (u - first u of that owner)
        uu = torch.arange(G * C, device=dev).view(1, -1).expand(S, -1).contiguous()
        p_of = torch.searchsorted(ends.contiguous(), uu, right=True).clamp_(max=G - 1)
        stage_index = (p_of * C + (uu - off.gather(1, p_of))).clamp_(0, G * C - 1)
        stage_index = torch.where(uu < ends[:, -1:], stage_index, torch.zeros((), dtype=torch.int64, device=dev))
        back_index = torch.where(valid, off.unsqueeze(2) + j, torch.full((), G * C, dtype=torch.int64, device=dev)).view(S, G * C)
        st = StaticSplit(C=C, stage_index=stage_index, back_index=back_index.contiguous(), over_host=over_host,
                         over_ready=over_ready, used_rows=ask.sum().view(1))
        return ChunkPlan(S=S, B=B, sc=None, rc=None, req=rp, req_all=req_all, req_start=req_start, owner=owner, static=st,
                         unique_rows=0, remote_rows=S * (G - 1) * C)     # (the block a rank "sends" to itself never leaves it)

    def _overflowed(self, plan: "ChunkPlan") -> bool:
        """Did ANY rank's equal-split plan of this chunk need more than the capacity?  The local flag comes from pinned
        memory (a wait for the stream the plan was built on); the agreement is a CPU all-reduce."""
        st = plan.static
        if st.over_ready is not None:
            st.over_ready.synchronize()
        flag = torch.tensor([int(st.over_host.item())], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.control_group)
        return bool(flag.item())

    def _set_auto_capacity(self, plan: "ChunkPlan") -> None:
        """capacity="auto": from the first, exactly planned chunk -- the largest per-peer count anywhere, times the margin."""
        most = max([max(max(r) for r in plan.sc), max(max(r) for r in plan.rc), 1])
        t = torch.tensor([most], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.control_group)
        self.capacity = int(-(-int(t.item() * self.capacity_margin) // 64) * 64)
        self._capacity_auto = False

    def _split_fetch(self, pos, neg, ask, sc, rc, rc_dev, rg, rmap, req_all, n, n_req) -> "FetchSplit":
        """Early / late split of every step's fetch (see FetchSplit).  Owner side: which requested rows of step s no
        rank touches in step s-1 (step 0 of a chunk: none -- the previous chunk's last step is not looked at); the flags
        travel back to the requesters by one all-to-all of bytes; one host read-back for the split sizes."""
        G, dev, S = self.world, pos.device, int(pos.shape[0])
        rows_local = int(self.shard.shape[0])
        touched = torch.zeros(S * rows_local, dtype=torch.bool, device=dev)
        ids = torch.cat([pos, neg], 1).reshape(S, -1).to(torch.int64)
        mine = (ids >= 0) & (ids < self.N) & (ids % G == self.rank)      # every own row a step references may change in it
        base = torch.arange(S, device=dev).view(S, 1) * rows_local
        touched[(base + ids // G)[mine]] = True
        req_step = torch.repeat_interleave(torch.arange(S, device=dev), rc_dev.sum(1), output_size=n_req)
        req64 = req_all.to(torch.int64)
        touched[req_step * rows_local + req64] = True                   # ... and every row somebody fetches (its sum comes back)
        early_req = torch.zeros(n_req, dtype=torch.bool, device=dev)
        later = req_step >= 1
        early_req[later] = ~touched[(req_step[later] - 1) * rows_local + req64[later]]
        # flags back to the requesters, in the order their id lists arrived in; there back into staging order
        flags_send = early_req[rmap].to(torch.uint8) if n_req else torch.empty(0, dtype=torch.uint8, device=dev)
        flags_recv = self._a2a(flags_send, rc.sum(0).tolist(), sc.sum(0).tolist(), group=self.plan_group)
        flags = flags_recv[rg].bool() if n else torch.empty(0, dtype=torch.bool, device=dev)
        U_dev = ask.sum(1)
        step_of = torch.repeat_interleave(torch.arange(S, device=dev), U_dev, output_size=n)
        u_of = torch.arange(n, device=dev) - (torch.cumsum(U_dev, 0) - U_dev)[step_of]
        host = torch.stack([_segment_sums(flags, ask), _segment_sums(early_req, rc_dev)]).cpu()   # the split sizes
        sc_e, rc_e = host[0], host[1]
        sc_l, rc_l = sc - sc_e, rc - rc_e
        starts = lambda m: [0] + torch.cumsum(m.sum(1), 0).tolist()
        return FetchSplit(sc_e=sc_e.tolist(), sc_l=sc_l.tolist(), rc_e=rc_e.tolist(), rc_l=rc_l.tolist(),
                          idx_early=u_of[flags], idx_late=u_of[~flags], ie_start=starts(sc_e), il_start=starts(sc_l),
                          req_early=req_all[early_req], req_late=req_all[~early_req], re_start=starts(rc_e),
                          rl_start=starts(rc_l), staged={}, done={})

    def _prefetch(self, plan: "ChunkPlan", s: int) -> None:
        """The early rows of step s: gather at the owners, all-to-all, placed at their staging positions -- on the
        communication stream, behind everything enqueued so far (the previous step's updates, the plan), beside
        whatever the main stream does next (the current step's kernels, which touch none of these rows)."""
        pre = plan.pre
        main = torch.cuda.current_stream(self.shard.device) if self._comm is not None else None
        if main is not None:
            here = torch.cuda.Event()
            here.record(main)
            self._comm.wait_event(here)
        ctx = torch.cuda.stream(self._comm) if self._comm is not None else _NullContext()
        with ctx:
            rows = self.k.gather_rows(self.shard, pre.req_early[pre.re_start[s]:pre.re_start[s + 1]])
            recv = self._a2a(rows, pre.rc_e[s], pre.sc_e[s])
            staged = torch.empty(int(sum(plan.sc[s])), self.d, dtype=self.shard.dtype, device=self.shard.device)
            if recv.shape[0]:
                staged.index_copy_(0, pre.idx_early[pre.ie_start[s]:pre.ie_start[s + 1]], recv)
            done = None
            if self._comm is not None:
                done = torch.cuda.Event()
                done.record(self._comm)
                staged.record_stream(main)
        pre.staged[s], pre.done[s] = staged, done

    def step_planned(self, plan: "ChunkPlan", s: int, lr: float) -> torch.Tensor:
        """Step s of a planned chunk: fetch the other owners' rows (all-to-all), fused score/hinge/grad on the
        shard + staging buffer, own rows updated and staged rows reduced in one kernel, gradient sums back to the
        owners (all-to-all), added there."""
        staged = gsum = None
        if self.world > 1:
            sc, rc = (plan.sc[s], plan.rc[s]) if plan.sc is not None else (None, None)
            if self.peer_mapped:
                self._rank_barrier()                                    # every owner has finished the previous step's updates
                gsum = torch.zeros(int(sum(sc)), self.d, dtype=self.shard.dtype, device=self.shard.device)
            elif plan.pre is not None:                                  # overlapped schedule: this step's early rows are
                pre = plan.pre                                          # already on their way (or there); fetch the rest
                staged = pre.staged.pop(s, None)
                if staged is None:
                    staged = torch.empty(int(sum(sc)), self.d, dtype=self.shard.dtype, device=self.shard.device)
                rows_out = self.k.gather_rows(self.shard, pre.req_late[pre.rl_start[s]:pre.rl_start[s + 1]])
                recv = self._a2a(rows_out, pre.rc_l[s], pre.sc_l[s])
                if recv.shape[0]:
                    staged.index_copy_(0, pre.idx_late[pre.il_start[s]:pre.il_start[s + 1]], recv)
                if s + 1 < plan.S:
                    self._prefetch(plan, s + 1)                         # next step's early rows, beside this step's kernels
                done = pre.done.pop(s, None)
                if done is not None:
                    torch.cuda.current_stream(self.shard.device).wait_event(done)
                gsum = torch.zeros_like(staged)
            elif plan.static is not None:                               # equal splits: C rows per peer, nothing from the host
                st = plan.static
                req = plan.req_all[plan.req_start[s]:plan.req_start[s + 1]]
                rows_out = self.k.gather_rows(self.shard, req)          # [G*C, d]; unused slots (id -1): zero rows
                recv = torch.empty_like(rows_out)
                dist.all_to_all_single(recv, rows_out, group=self.group)
                staged = recv.index_select(0, st.stage_index[s])        # into staging order (rows past the step's count: unused)
                gsum = torch.zeros(staged.shape[0] + 1, self.d, dtype=self.shard.dtype, device=self.shard.device)   # + the zero row
            else:
                req = plan.req_all[plan.req_start[s]:plan.req_start[s + 1]]
                rows_out = self.k.gather_rows(self.shard, req)          # owners gather ...
                staged = self._a2a(rows_out, rc, sc)                    # ... rows arrive in staging order
                gsum = torch.zeros_like(staged)                         # rows with > 16 slots add atomically
        loss = self.k.grad(self.shard, staged, plan.req, s, lr, self.margin, self.model, self.max_norm, gsum)
        if self.peer_mapped:
            self._rank_barrier()                                        # nobody still reads rows that are about to change
        self.k.apply(self.shard, plan.req, s, gsum)                     # own rows in place, staged rows -> gsum
        if self.world > 1 and plan.static is not None:
            back = gsum.index_select(0, plan.static.back_index[s])      # padded slot (p, j) <- its staged row's sum, or zeros
            recv_g = torch.empty_like(back)
            dist.all_to_all_single(recv_g, back, group=self.group)
            self.k.owner_apply(self.shard, plan.owner, s, recv_g)
        elif self.world > 1:
            recv_g = self._a2a(gsum, sc, rc)                            # sums back to the owners
            self.k.owner_apply(self.shard, plan.owner, s, recv_g)
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
            neg = self.sample_negatives(pos, first_step).to(torch.int32)
            plan = self.plan_chunk(pos, neg)
            plan.ready = None
            plan.inputs = (pos, neg)
            return plan
        if inputs_ready is None:
            self._side.wait_stream(torch.cuda.current_stream(pos.device))
        else:
            self._side.wait_event(inputs_ready)
        with torch.cuda.stream(self._side):
            neg = self.sample_negatives(pos, first_step).to(torch.int32)
            plan = self.plan_chunk(pos, neg)
            plan.ready = torch.cuda.Event()
            plan.ready.record(self._side)
        plan.inputs = (pos, neg)
        return plan

    def _adopt(self, plan: "ChunkPlan") -> None:
        """Make a plan built on the side stream usable on the current one: wait for it, and tell the caching
        allocator that its tensors are read here (they are freed while these reads may still be queued)."""
        if getattr(plan, "ready", None) is None:
            return
        cur = torch.cuda.current_stream(self.shard.device)
        cur.wait_event(plan.ready)
        tensors = list(self.k.plan_tensors(plan.req, plan.owner)) if hasattr(self.k, "plan_tensors") else []
        if plan.req_all is not None:
            tensors.append(plan.req_all)
        if plan.own_rows_dev is not None:
            tensors.append(plan.own_rows_dev)
        if plan.static is not None:
            tensors += [plan.static.stage_index, plan.static.back_index]
        if plan.pre is not None:
            split = [plan.pre.idx_early, plan.pre.idx_late, plan.pre.req_early, plan.pre.req_late]
            tensors += split
            if self._comm is not None:
                for t in split:
                    if t.is_cuda:
                        t.record_stream(self._comm)
        for t in tensors:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(cur)

    def _settle(self, plan: "ChunkPlan") -> "ChunkPlan":
        """Equal-split bookkeeping once a plan is adopted: an auto capacity is fixed from the first (exact) chunk; an
        equal-split plan that overflowed the capacity on ANY rank is replaced by the exact plan of the same chunk."""
        if self._capacity_auto and plan.static is None and self.world > 1:
            self._set_auto_capacity(plan)
        if plan.static is not None and self._overflowed(plan):
            pos, neg = plan.inputs
            self.replanned_chunks += 1
            exact = self.plan_chunk(pos, neg, exact=True)               # (on the current stream: the rare path)
            exact.inputs, exact.ready = plan.inputs, None
            return exact
        return plan

    def run_pipelined(self, chunks, lr_fn, lookahead: torch.Tensor = None) -> torch.Tensor:
        """Train the chunks (a sequence of pos [S,B,3] tensors) back to back.  The steps of chunk c are
        enqueued first (asynchronously); chunk c+1's negatives and exchange plan -- which never depend on the
        table -- are then built on the side stream while those steps execute, so the host's wait inside the
        planner (the all-to-all split sizes are data dependent) falls into time the device spends training --
        with its own `plan_group`; on the data path's communicator the plan's collectives queue behind that chunk's
        all-to-alls (see __init__).
        lookahead: the positives the NEXT call will start with -- one chunk or a list of the first chunks, the same tensor
        objects, at the global step this call ends on: their plans are built while this call's last chunks execute and
        adopted by that call, so a training loop that calls this once per validation tick never plans on the critical
        path (with plans two chunks ahead, pass the next call's first two chunks).
        Every rank must pass a lookahead, or none (the plan holds two collectives).
        Returns every step's losses [sum S, B]."""
        given = list(chunks)
        chunks = [c.to(torch.int32).contiguous() for c in given]
        ready = None
        if self._side is not None and (chunks or lookahead is not None):
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.shard.device))   # every chunk's positives exist from here on
        pend, self._pending = getattr(self, "_pending", None) or [], None
        # Plans run `plan_depth` chunks ahead of the steps (they depend on the positives and the Philox keys only, never on
        # the table).  Depth 2, enqueue order P0 P1 S0 P2 S1 P3 ...: the collectives of plan c+2 queue behind the steps of
        # chunk c only, and what follows them (the owner-side sort of the received request lists) runs BESIDE chunk c+1 --
        # with depth 1 (S0 P1 S1 ...) chunk c+1 could not start before that sort, one device bubble per chunk; and a host
        # that waits inside a plan (exact splits: the split sizes) waits while one whole chunk of steps is still queued.
        ahead = [] if lookahead is None else ([lookahead] if isinstance(lookahead, torch.Tensor) else list(lookahead))
        seq = chunks + [t.to(torch.int32).contiguous() for t in ahead]
        orig = given + ahead                                             # the caller's objects (a pending plan is matched by identity)
        starts = [self.global_step]
        for ch in seq:
            starts.append(starts[-1] + int(ch.shape[0]))
        plans = [None] * len(seq)
        for i, (t, at, pl) in enumerate(pend):                           # planned during the previous call
            if i < len(seq) and t is orig[i] and at == starts[i]:
                plans[i] = pl
            else:
                break

        def plan_ahead(i):
            if i < len(seq) and plans[i] is None:
                plans[i] = self._plan_ahead(seq[i], starts[i], ready)

        depth = max(1, int(self.plan_depth))
        for i in range(depth):
            plan_ahead(i)
        losses = []
        for c in range(len(chunks)):
            plan, plans[c] = plans[c], None
            self._adopt(plan)
            plan = self._settle(plan)
            losses += [self.step_planned(plan, s, lr_fn(self.global_step)) for s in range(plan.S)]
            self._stats_of = plan                                        # (see `stats`: resolved when read)
            plan_ahead(c + depth)
        if ahead:
            for i in range(len(chunks), len(seq)):
                plan_ahead(i)
            self._pending = [(orig[i], starts[i], plans[i]) for i in range(len(chunks), len(seq))]
        return torch.stack(losses, 0) if losses else None

    @property
    def stats(self) -> StepStats:
        """Per-step statistics of the last chunk trained (equal splits: remote_rows counts the PADDED slots -- what the
        links really carry).  Reading them may wait for the device (world size 1 keeps its row count there)."""
        done = self._stats_of
        if done is not None:
            done.resolve()
            self._stats = StepStats(unique_rows=done.unique_rows // done.S, remote_rows=done.remote_rows // done.S,
                                    bytes_sent=int(done.remote_rows // done.S * (2 * self.d * 4 + 4)),
                                    early_rows=(sum(sum(r) for r in done.pre.sc_e) // done.S) if done.pre is not None else 0)
            self._stats_of = None
        return self._stats

    @stats.setter
    def stats(self, value: StepStats) -> None:
        self._stats, self._stats_of = value, None

    def run(self, pos: torch.Tensor, lr_fn, neg: torch.Tensor = None) -> torch.Tensor:
        """Train S consecutive steps on pos [S,B,3]; lr_fn(global_step) -> lr.  Returns losses [S,B]."""
        pos = pos.to(torch.int32).contiguous()
        if neg is None:
            neg = self.sample_negatives(pos)
        plan = self.plan_chunk(pos, neg.to(torch.int32))
        plan.inputs = (pos, neg.to(torch.int32))
        plan = self._settle(plan)
        losses = [self.step_planned(plan, s, lr_fn(self.global_step)) for s in range(plan.S)]
        self._stats_of = plan
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
        """All-gather the shards back into the [N,d] table (checkpointing / tests); real-valued rows also when the
        shard is carried in the frequency domain (model="hole")."""
        G = self.world
        mine = self.shard.clone()
        if self._spectral_resident:
            self.k.from_spectral(mine)
        if G == 1:
            return mine
        rows = (self.N + G - 1) // G
        pad = torch.zeros(rows, self.d, dtype=mine.dtype, device=mine.device)
        pad[: mine.shape[0]] = mine
        parts = [torch.empty_like(pad) for _ in range(G)]
        dist.all_gather(parts, pad, group=self.group)
        full = torch.empty(rows * G, self.d, dtype=mine.dtype, device=mine.device)
        for g in range(G):
            full[g::G] = parts[g]
        return full[: self.N]
