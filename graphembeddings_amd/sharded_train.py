"""The training / inference driver on G ranks: `python -m graphembeddings_amd.train --gpus G ...`.

The reference is one process on one device (holE.py:309: one tf.Session); what is kept is the BEHAVIOUR of its driver
(run_training, holE.py:249-370; --infer, holE.py:427-490, 564-575) on a table that is row-sharded over the ranks
(sharded.ShardedTrainer: owner(id) = id % G, rows and gradient sums routed by all-to-all):

  * batch_count = triple_count // batch_size and an epoch of batch_count - 1 steps, as on one GPU; --batch_size is the
    GLOBAL batch, every rank trains batch_size / G positives of it per step -- the triples whose HEAD row it owns
    (a third of a step's entity rows are then local), reshuffled per epoch;
  * lr = inverse_time_decay(global step) (holE.py:292-294);
  * 16 validation ticks per epoch (holE.py:351-354): the mean hinge of ONE random validation batch with fresh negatives --
    batch_size / G triples per rank, scored by a training step with learning rate 0 (same exchange, same kernels; a row
    plus -0 * gradient is the row), the means combined by ONE scalar all-reduce;
  * the "pocket" (holE.py:329, 357-360): when the validation loss improves on the best so far (2.0 at the start) every
    rank keeps a copy of its shard; the copies go to `model.ckpt.shard<r>-of-<G>.pt` once per epoch and every
    --checkpoint_seconds, and --resume_checkpoint reads them back (or slices a one-GPU `model.ckpt.pt`);
  * at the end rank 0 also writes the gathered best table as the one-GPU `model.ckpt.pt`, so --infer and
    --save_embeddings work on any number of GPUs.

--infer on G ranks (evaluate_sharded): every rank sweeps ITS OWN entity rows as candidates for every test triple.  A
position in the reference's heap is a count of candidates that pop before the true one, so the ranks' counts ADD: the
rank that owns a triple's true entity runs the ordinary sweep for it (its counts + the true triple's loss), the loss is
shared by one all-reduce, the other ranks sweep against that loss (ge_rank_1vK_vs_loss), and one more all-reduce sums
n_before / n_known_before.  Equal table rows give bit-equal losses on every rank (same planes, same MFMA order), so the
ranks equal evaluate.link_prediction_ranks' on one GPU exactly, ties included.
"""
from __future__ import annotations

import errno
import json
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import data as D
from . import evaluate as E
from . import hole as H
from . import sharded as S

CHUNK = 64          # steps per exchange plan


def dist_setup():
    """(rank, world, device) of this process; the process group is created from the launcher's environment
    (graphembeddings_amd/launch.py or torchrun).  GE_DIST_BACKEND=gloo GE_SINGLE_DEVICE=1 rehearse G ranks on one GPU."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if os.environ.get("GE_SINGLE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        backend = os.environ.get("GE_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, dev


def partition_by_head(triples: np.ndarray, rank: int, world: int) -> np.ndarray:
    """The triples whose head row this rank owns (owner(id) = id % world)."""
    t = np.asarray(triples)
    return np.ascontiguousarray(t[(t[:, 0] % world) == rank])


def shard_checkpoint_path(output_dir: str, rank: int, world: int) -> str:
    return os.path.join(output_dir, f"model.ckpt.shard{rank}-of-{world}.pt")


def save_shard(output_dir: str, shard: torch.Tensor, global_step: int, n_rows: int, rank: int, world: int) -> None:
    p = shard_checkpoint_path(output_dir, rank, world)
    torch.save({"shard": shard.detach().cpu(), "global_step": int(global_step), "n_rows": int(n_rows),
                "rank": int(rank), "world": int(world)}, p + ".tmp")
    os.replace(p + ".tmp", p)


def load_shard(output_dir: str, n_rows: int, d: int, rank: int, world: int, dev):
    """This rank's rows and the global step: from the shard files of a run on the same number of ranks, else sliced
    out of a one-GPU checkpoint."""
    p = shard_checkpoint_path(output_dir, rank, world)
    if os.path.exists(p):
        ck = torch.load(p, weights_only=True)
        if int(ck["world"]) != world or int(ck["rank"]) != rank or int(ck["n_rows"]) != n_rows:
            raise ValueError(f"{p} belongs to another sharding")
        shard, gs = ck["shard"].to(dev).contiguous(), int(ck["global_step"])
    else:
        from . import train as T
        ck = torch.load(T.checkpoint_path(output_dir), weights_only=True)
        if tuple(ck["embeddings"].shape) != (n_rows, d):
            raise ValueError("checkpoint shape does not match the data / --embedding_dim")
        shard, gs = ck["embeddings"][rank::world].to(dev).contiguous(), int(ck["global_step"])
    if tuple(shard.shape) != (S.shard_num_rows(n_rows, rank, world), d):
        raise ValueError("checkpoint shard shape does not match the data / --embedding_dim")
    return shard, gs


def _all_min_int(v: int, dev) -> int:
    t = torch.tensor([int(v)], dtype=torch.int64, device=dev)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def run_training_sharded(data: D.HolEData, FLAGS, log=print) -> dict:
    rank, world, dev = dist_setup()
    if FLAGS.log_loss:
        raise NotImplementedError("--log_loss runs on one GPU (ge_train_steps_logloss); the row-sharded step is the hinge step")
    if FLAGS.batch_size % world:
        raise ValueError(f"--batch_size {FLAGS.batch_size} (the global batch) must be a multiple of --gpus {world}")
    B = FLAGS.batch_size // world
    N, d = data.entity_count, FLAGS.embedding_dim
    batch_count = data.triple_count // FLAGS.batch_size
    say = log if rank == 0 else (lambda *a, **k: None)
    say('Embedding dimension: ', d, 'Batch size: ', FLAGS.batch_size, f'({B} per rank x {world})', 'Batch count: ', batch_count)
    if batch_count < 2:
        raise ValueError('need at least 2 batches of training triples')
    # the output directory must not exist unless --resume_checkpoint (holE.py:254-255): rank 0 decides for everyone
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    if rank == 0:
        if not FLAGS.resume_checkpoint and os.path.isdir(FLAGS.output_dir):
            flag += 1
        else:
            try:
                os.makedirs(FLAGS.output_dir)
            except OSError as e:
                if e.errno != errno.EEXIST:
                    raise
    if world > 1:
        dist.all_reduce(flag)
    if int(flag.item()):
        raise Exception("WARNING: " + FLAGS.output_dir + " already exists!")

    names, id_to_type, offsets, ids = data.type_arrays()
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=FLAGS.padded_size, device=dev)
    global_step = 0
    if FLAGS.resume_checkpoint:
        shard, global_step = load_shard(FLAGS.output_dir, N, d, rank, world, dev)
    else:
        # the one-GPU initializer with the one-GPU seed, sliced: the run starts from the table a one-GPU run would
        full = H.init_embeddings(N, d, device=dev, seed=FLAGS.seed)
        shard = full[rank::world].contiguous()
        del full
    mine = partition_by_head(data.triples, rank, world)
    if _all_min_int(len(mine), dev) < B:
        raise ValueError(f"a rank owns fewer than {B} training triples: lower --batch_size or --gpus")
    local = torch.as_tensor(mine.astype(np.int32)).to(dev)
    del mine
    trainer = S.ShardedTrainer(shard, N, tt, margin=FLAGS.margin, model=FLAGS.model, seed=FLAGS.seed)
    trainer.global_step = global_step
    decay_steps = FLAGS.learning_decay_steps * batch_count

    def lr_fn(gs):
        return H.inverse_time_decay(FLAGS.learning_rate, gs, decay_steps, FLAGS.learning_decay_rate)

    valid = None
    if data.validation_triples is not None and len(data.validation_triples) >= B:
        valid = torch.as_tensor(np.asarray(data.validation_triples).astype(np.int32)).to(dev)
    gen = torch.Generator(device=dev).manual_seed(FLAGS.seed * 1000003 + rank)
    vgen = torch.Generator(device=dev).manual_seed((FLAGS.seed ^ 0x5EED) * 1000003 + rank)

    tick = max(1, batch_count // 16)          # guard for the ZeroDivisionError of holE.py:351
    pocket_loss = 2.
    history = []
    state = {"best_saved": 2.0, "pocket_step": global_step, "last_write": time.time(), "pocket": None, "ticks": 0}

    def validation_tick():
        """Mean hinge of one random validation batch (B per rank) with fresh negatives: a step with learning rate 0."""
        nonlocal pocket_loss
        rows = torch.randint(0, valid.shape[0], (B,), device=dev, generator=vgen)
        vpos = valid[rows].unsqueeze(0).contiguous()
        gs = trainer.global_step
        neg = trainer.sample_negatives(vpos, first_step=(1 << 40) + state["ticks"])      # a Philox stream of its own
        state["ticks"] += 1
        loss = trainer.run(vpos, lambda _gs: 0.0, neg=neg)
        trainer.global_step = gs                                                          # not a training step
        vlm = trainer.mean_loss(loss[0])                                                  # the scalar all-reduce
        say('\tStep {} Validation Loss: {}...'.format(gs, vlm))
        history.append((gs, vlm))
        if vlm < pocket_loss:                 # the same number on every rank: every rank keeps its shard of the best table
            pocket_loss = vlm
            state["pocket"], state["pocket_step"] = trainer.shard.clone(), gs

    def write_pocket(epoch):
        state["last_write"] = time.time()
        if state["pocket"] is None or pocket_loss >= state["best_saved"]:
            return
        state["best_saved"] = pocket_loss
        pk = state["pocket"].clone()
        if trainer._spectral_resident:
            trainer.k.from_spectral(pk)                                                   # files hold the real-valued rows
        save_shard(FLAGS.output_dir, pk, state["pocket_step"], N, rank, world)
        if rank == 0:
            with open(os.path.join(FLAGS.output_dir, "model.ckpt.shards.json"), "w") as f:
                json.dump({"world": world, "n_rows": N, "embedding_dim": d, "global_step": state["pocket_step"],
                           "validation_loss": pocket_loss}, f)
        say('Epoch {}, (Model saved with loss {})'.format(epoch, pocket_loss))

    def chunks_of(first_batch, n):
        """n consecutive steps' positives starting at batch index first_batch of this epoch's shuffled local triples, as
        [S,B,3] chunks of at most CHUNK steps; a rank with fewer triples than the epoch has steps wraps around."""
        out = []
        per = local.shape[0] // B
        i = first_batch
        while i < first_batch + n:
            m = min(CHUNK, first_batch + n - i)
            out.append(torch.stack([local[((j % per) * B):((j % per) + 1) * B] for j in range(i, i + m)], 0).contiguous())
            i += m
        return out

    t_start = time.time()
    done = False
    last_loss = None
    for epoch in range(1, FLAGS.num_epochs + 1):
        say('Training epoch {}...'.format(epoch))
        local = local[torch.randperm(local.shape[0], device=dev, generator=gen)]
        batch = 1
        while batch < batch_count and not done:
            if batch % tick == 0 and valid is not None:
                validation_tick()
            nxt = min(batch_count, (batch // tick + 1) * tick)
            n = nxt - batch
            if FLAGS.max_steps:
                n = min(n, FLAGS.max_steps - (trainer.global_step - global_step))
            if n > 0:
                last_loss = trainer.run_pipelined(chunks_of(batch, n), lr_fn)[-1]
            if FLAGS.checkpoint_seconds > 0:
                # (every rank must take the same branch: the pocket write holds no collective, but keep the ranks in step)
                due = torch.tensor([1 if time.time() - state["last_write"] >= FLAGS.checkpoint_seconds else 0], device=dev)
                if world > 1:
                    dist.all_reduce(due, op=dist.ReduceOp.MAX)
                if int(due.item()):
                    write_pocket(epoch)
            batch += max(n, 0)
            if FLAGS.max_steps and trainer.global_step - global_step >= FLAGS.max_steps:
                done = True
            if n <= 0:
                break
        write_pocket(epoch)
        if done:
            break
    torch.cuda.synchronize()
    say('Done training -- epoch limit reached')
    # the one-GPU checkpoint of the best table (the final one when no validation tick ever improved on 2.0)
    if state["pocket"] is not None:
        trainer.shard.copy_(state["pocket"])
        final_step = state["pocket_step"]
    else:
        final_step = trainer.global_step
    full = trainer.gather_full_table()
    if state["pocket"] is None:
        pk = trainer.shard.clone()
        if trainer._spectral_resident:
            trainer.k.from_spectral(pk)
        save_shard(FLAGS.output_dir, pk, final_step, N, rank, world)
    if rank == 0:
        from . import train as T
        T.save_checkpoint(FLAGS.output_dir, full, final_step)
    mean_final = trainer.mean_loss(last_loss) if last_loss is not None else float("nan")
    if world > 1:
        dist.barrier()
    return {"steps": trainer.global_step - global_step, "seconds": time.time() - t_start, "pocket_loss": pocket_loss,
            "history": history, "global_step": trainer.global_step, "final_mean_hinge": mean_final, "world": world}


@torch.no_grad()
def evaluate_sharded(shard: torch.Tensor, n_rows: int, relation_count: int, test_triples: np.ndarray,
                     known_triples: np.ndarray = None, *, both_sides: bool = True, model: str = "complex",
                     max_norm: float = 1.0, infer_threshold: float = None, batch: int = 1 << 16, rank: int = None,
                     world: int = None, group=None):
    """Raw and filtered ranks of every test triple among ALL entity rows (ids >= relation_count), the candidates sharded like
    the table: rank r sweeps the rows it owns.  Returns (raw, filtered[, confident]) int64 arrays, tails then (both_sides)
    heads, identical on every rank and equal to evaluate.link_prediction_ranks on the gathered table.  model: "complex" or
    "hole_spectral" (shard in the frequency domain)."""
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    dev, d = shard.device, int(shard.shape[1])
    N, R = int(n_rows), int(relation_count)
    if d % 8 != 0 or d > H.rank_max_dim():
        raise ValueError("the sharded evaluation runs the fused sweep: embedding_dim a multiple of 8, <= %d" % H.rank_max_dim())
    test = np.asarray(test_triples, dtype=np.int64)

    def reduce_(t, op=dist.ReduceOp.SUM):
        if world > 1:
            dist.all_reduce(t, op=op, group=group)
        return t

    # ---- the rows every rank needs beside its own: heads, tails and relations of the test triples.  Each is contributed by
    # its owner (zeros elsewhere) and summed -- one all-reduce of [n_needed, d] (test sets are small next to the table)
    needed = np.unique(test.reshape(-1))
    needed_dev = torch.as_tensor(needed).to(dev)
    qrows = torch.zeros(len(needed), d, dtype=shard.dtype, device=dev)
    own = (needed_dev % world) == rank
    qrows[own] = shard[(needed_dev[own] // world)]
    reduce_(qrows)
    # ---- the table this rank's sweeps index: [its shard | the needed rows]; candidates = its entity rows
    n_loc = int(shard.shape[0])
    aug = torch.cat([shard, qrows], 0)
    del qrows
    pos_in_needed = torch.full((N,), -1, dtype=torch.int64, device=dev)
    pos_in_needed[needed_dev] = torch.arange(len(needed), device=dev)
    loc_ids = torch.arange(n_loc, device=dev, dtype=torch.int64)
    glob_of_loc = loc_ids * world + rank                                   # global id of local row l
    cand_loc = loc_ids[glob_of_loc >= R].to(torch.int32)                   # local rows that are entities: the candidates
    cand_glob = glob_of_loc[glob_of_loc >= R]
    K = int(cand_loc.numel())
    planes = H.RankPlanes(aug, cand_loc, max_norm=max_norm, model=model) if K else None
    # candidate position of a GLOBAL entity id in this rank's list (for the known-true cells)
    pos_of = torch.full((N,), -1, dtype=torch.int64, device=dev)
    pos_of[cand_glob] = torch.arange(K, device=dev)

    out_raw, out_fil, out_conf = [], [], []
    for side in (("tail", "head") if both_sides else ("tail",)):
        fixed_col, true_col = (0, 1) if side == "tail" else (1, 0)
        index = known_triples if isinstance(known_triples, dict) else None
        kidx = (index[side] if index is not None else E.KnownIndex(known_triples, N, side, dev))
        for s0 in range(0, len(test), batch):
            chunk = torch.as_tensor(test[s0:s0 + batch]).to(dev)
            fixed, rel, true_id = chunk[:, fixed_col], chunk[:, 2], chunk[:, true_col]
            nB = int(chunk.shape[0])
            hr = torch.stack([n_loc + pos_in_needed[fixed], n_loc + pos_in_needed[rel]], 1).to(torch.int32)
            n_before = torch.zeros(nB, dtype=torch.int32, device=dev)
            n_known = torch.zeros(nB, dtype=torch.int32, device=dev)
            true_loss = torch.zeros(nB, dtype=torch.float32, device=dev)
            mine = ((true_id % world) == rank)
            if (true_id < R).any():
                raise ValueError("a test triple's true entity is a relation row")

            def cells_for(rows_mask):
                """The known cells of a SUBSET of the rows (the kernel numbers rows 0.. within the call)."""
                sub = rows_mask.nonzero().view(-1)
                return sub, kidx.cells(fixed[sub], rel[sub], pos_of, K)
            # (a) rows whose true entity lives here: the ordinary sweep -- counts and the true triple's loss
            if K and bool(mine.any()):
                sub, (o1, r1) = cells_for(mine)
                a, b, tl = H.rank_candidates(aug, hr[sub], (true_id[sub] // world).to(torch.int32), cand_loc, known_off=o1,
                                             known_rc=r1, cand_is_head=(side == "head"), max_norm=max_norm,
                                             return_true_loss=True, model=model, planes=planes)
                n_before[sub], n_known[sub], true_loss[sub] = a, b, tl
            reduce_(true_loss)                                             # every row's loss, from its one owner
            # (b) the other rows: against that loss; ties pop in GLOBAL id order (holE.py:434)
            others = ~mine
            if K and bool(others.any()):
                sub, (o2, r2) = cells_for(others)
                # the kernel compares candidate ids of ITS list (local rows l = global // world) with ref_id: global order
                # g < g_true  <=>  (l, rank) < (l_true, rank_true) lexicographically; a reference id of l_true (+1 when this
                # rank's index is below the true owner's) turns "l < ref" into exactly that
                ref = (true_id[sub] // world) + (rank < (true_id[sub] % world)).to(torch.int64)
                a, b = H.rank_candidates_vs_loss(aug, hr[sub], ref.to(torch.int32), true_loss[sub], cand_loc, known_off=o2,
                                                 known_rc=r2, cand_is_head=(side == "head"), max_norm=max_norm, model=model,
                                                 planes=planes)
                n_before[sub], n_known[sub] = a, b
            cnt = torch.stack([n_before, n_known]).to(torch.int64)
            reduce_(cnt)
            raw = cnt[0] + 1
            fil = raw - cnt[1]
            if infer_threshold is not None:
                below = torch.zeros(nB, dtype=torch.int64, device=dev)
                if K:
                    below = H.confident_rows(aug, hr, cand_loc, infer_threshold, cand_is_head=(side == "head"), max_norm=max_norm,
                                             model=model, planes=planes).to(torch.int64)
                conf = reduce_(below) > 0                                  # some rank holds a candidate below the threshold
                raw, fil = raw[conf], fil[conf]
                out_conf.append(conf.cpu().numpy())
            out_raw.append(raw.cpu().numpy())
            out_fil.append(fil.cpu().numpy())
    res = (np.concatenate(out_raw), np.concatenate(out_fil))
    if infer_threshold is not None:
        res += (np.concatenate(out_conf),)
    return res


def infer_sharded(FLAGS, log=print) -> dict:
    """--infer on G ranks: the checkpoint's rows of this rank, every entity row of the table as a candidate."""
    rank, world, dev = dist_setup()
    data = D.init_inference_data(FLAGS.data_dir, min_mentions=None)
    N, d = data.entity_count, FLAGS.embedding_dim
    shard, _ = load_shard(FLAGS.output_dir, N, d, rank, world, dev)
    model = FLAGS.model
    if model == "hole":                        # the files hold real-valued rows; ranks use the HolE score on the half spectrum
        shard, model = H.hole_to_spectral(shard), "hole_spectral"
    parts = [a for a in (data.triples, data.validation_triples) if a is not None]
    known = np.concatenate(parts, 0) if parts else None
    res = evaluate_sharded(shard, N, data.relation_count, data.test_array, known, both_sides=True, model=model,
                           infer_threshold=FLAGS.infer_threshold, rank=rank, world=world)
    raw, fil = res[0], res[1]
    sweeps = 2 * len(data.test_array)
    if raw.size == 0:
        out = {k: float("nan") for k in ("raw_mrr", "mean_raw_pos", "filtered_mrr", "mean_filtered_pos", "hits1", "hits3", "hits10")}
    else:
        out = E.mrr_and_hits(raw, fil)
    out["recorded"], out["sweeps"], out["world"] = int(raw.size), int(sweeps), world
    if rank == 0:
        log(f"is_confident (lowest loss < {FLAGS.infer_threshold}): {raw.size} of {sweeps} sweeps recorded")
        log("raw MRR {raw_mrr:.6f} (mean rank {mean_raw_pos:.1f}); filtered MRR {filtered_mrr:.6f} "
            "(mean rank {mean_filtered_pos:.1f}); hits@1/3/10 {hits1:.2f} / {hits3:.2f} / {hits10:.2f} %".format(**out))
    if world > 1:
        dist.barrier()
    return out
