"""Training / inference driver with the reference's CLI contract (holE.py:585-622).

  python -m graphembeddings_amd.train --data_dir D --output_dir O [--embedding_dim 200 ...]

Same flag names, defaults and meaning as holE.py:598-621.  What is reproduced is the BEHAVIOUR of
run_training (holE.py:249-370): batch_count = triple_count // batch_size; an "epoch" is
batch_count-1 steps (holE.py:340); lr = inverse_time_decay over learning_decay_steps epochs
(holE.py:292-294); 16 times per epoch the mean hinge of ONE random validation batch with fresh random
negatives is printed (holE.py:351-354) and the table is saved when it improves on the pocket loss,
which starts at 2.0 (holE.py:329, 357-360); the output directory must not exist unless
--resume_checkpoint (holE.py:254-255).  TF mechanics (queues, sessions, summaries, the V2
checkpoint bundle) are not: the table is saved as `model.ckpt.pt` (embeddings + global_step).
Between validation ticks the steps are enqueued natively by ge_train_steps (hinge) or
ge_train_steps_logloss (--log_loss) -- no Python per step.
Extras: --model hole (README.md:42 score), --seed, --max_steps, --checkpoint_seconds.
"""
from __future__ import annotations

import argparse
import errno
import os
import sys
import time

import numpy as np
import torch

from . import data as D
from . import evaluate as E
from . import hole as H


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    p.add_argument('--learning_rate', type=float, default=0.1, help='Initial learning rate.')
    p.add_argument('--learning_decay_steps', type=float, default=32, help='Learning rate decay steps (in epochs).')
    p.add_argument('--learning_decay_rate', type=float, default=0.5, help='Learning decay rate.')
    p.add_argument('--batch_size', type=int, default=512, help='Batch size.')
    p.add_argument('--num_epochs', type=int, default=1000, help='Number of training epochs.')
    p.add_argument('--embedding_dim', type=int, default=128, help='Embedding dimension.')
    p.add_argument('--log_loss', action='store_true', help='Use logistic loss istead of pairwise ranking loss.')
    p.add_argument('--l2_regularization', type=float, default=0.1, help='L2 regularization weight (log loss only).')
    p.add_argument('--negative_ratio', type=int, default=1, help='Number of negative labels sampled in log_loss.')
    p.add_argument('--margin', type=float, default=0.2, help='Hinge loss margin.')
    p.add_argument('--padded_size', type=int, default=1024,
                   help='The maximum number of entities to use for each type while sampling corrupt triples.')
    p.add_argument('--output_dir', type=str, required=True, help='Output (checkpoint) directory.')
    p.add_argument('--data_dir', type=str, required=True, help='Input data directory.')
    p.add_argument('--reader_threads', type=int, default=4, help='Accepted for compatibility; ingest is one pass.')
    p.add_argument('--resume_checkpoint', action='store_true', help='Resume training on the checkpoint model.')
    p.add_argument('--save_embeddings', action='store_true', help='Output the embeddings to stdout.')
    p.add_argument('--infer', action='store_true', help='Link-prediction evaluation from the latest checkpoint.')
    p.add_argument('--infer_threshold', type=float, default=0.05, help='Max loss to save triples')
    p.add_argument('--min_mentions', type=int, default=50000,
                   help='The minimum number of mentions for an entity to be a viable candidate in inference.')
    # extensions
    p.add_argument('--model', choices=['complex', 'hole'], default='complex')
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--max_steps', type=int, default=0, help='Stop after this many steps (0 = epoch limit only).')
    p.add_argument('--checkpoint_seconds', type=float, default=300.,
                   help='Also write the best table found so far this often inside an epoch (0 = once per epoch only).')
    p.add_argument('--gpus', type=int, default=1,
                   help='Ranks (one per GPU) the table is row-sharded over; > 1 without a launcher: this program starts them '
                        '(graphembeddings_amd/launch.py).  --batch_size stays the GLOBAL batch.')
    return p


def checkpoint_path(output_dir: str) -> str:
    return os.path.join(output_dir, 'model.ckpt.pt')


def save_checkpoint(output_dir: str, embeddings: torch.Tensor, global_step: int) -> None:
    tmp = checkpoint_path(output_dir) + '.tmp'
    torch.save({'embeddings': embeddings.detach().cpu(), 'global_step': int(global_step)}, tmp)
    os.replace(tmp, checkpoint_path(output_dir))


def load_checkpoint(output_dir: str, device='cuda'):
    ck = torch.load(checkpoint_path(output_dir), weights_only=True)
    return ck['embeddings'].to(device).contiguous(), int(ck['global_step'])


def run_training(data: D.HolEData, FLAGS, log=print) -> dict:
    if FLAGS.log_loss and FLAGS.model != 'complex':
        raise NotImplementedError('--log_loss is implemented for the ComplEx score (holE.py:191-196)')
    batch_count = data.triple_count // FLAGS.batch_size
    log('Embedding dimension: ', FLAGS.embedding_dim, 'Batch size: ', FLAGS.batch_size, 'Batch count: ', batch_count)
    if batch_count < 2:
        raise ValueError('need at least 2 batches of training triples')
    if not FLAGS.resume_checkpoint and os.path.isdir(FLAGS.output_dir):
        raise Exception("WARNING: " + FLAGS.output_dir + " already exists!")   # holE.py:254-255
    try:
        os.makedirs(FLAGS.output_dir)
    except OSError as e:
        if e.errno != errno.EEXIST:
            raise
    names, id_to_type, offsets, ids = data.type_arrays()
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=FLAGS.padded_size)
    global_step = 0
    if FLAGS.resume_checkpoint:
        embeddings, global_step = load_checkpoint(FLAGS.output_dir)
        if tuple(embeddings.shape) != (data.entity_count, FLAGS.embedding_dim):
            raise ValueError('checkpoint shape does not match the data / --embedding_dim')
    else:
        embeddings = H.init_embeddings(data.entity_count, FLAGS.embedding_dim, seed=FLAGS.seed)
    triples = torch.as_tensor(np.ascontiguousarray(data.triples)).cuda()
    trainer = H.Trainer(embeddings, triples, tt, FLAGS.batch_size, margin=FLAGS.margin,
                        learning_rate=FLAGS.learning_rate,
                        decay_steps=FLAGS.learning_decay_steps * batch_count,
                        decay_rate=FLAGS.learning_decay_rate, model=FLAGS.model, seed=FLAGS.seed,
                        spectral_resident=(FLAGS.model == 'hole' and FLAGS.embedding_dim % 2 == 0 and not FLAGS.log_loss))
    # HolE: `embeddings` is held in the frequency domain while training (hole.Trainer); validation scores it
    # there (model 'hole_spectral'), checkpoints store the real-valued table
    eval_model = 'hole_spectral' if trainer.spectral else FLAGS.model
    trainer.global_step = global_step
    gen = torch.Generator(device='cuda').manual_seed(FLAGS.seed)
    valid = None
    if data.validation_triples is not None and len(data.validation_triples) >= FLAGS.batch_size:
        valid = torch.as_tensor(data.validation_triples).cuda()
    K = max(1, FLAGS.negative_ratio)
    if FLAGS.log_loss:
        # --log_loss (holE.py:206-220): the same native loop, K corrupted batches per step drawn in the prepare launch
        trainer.enable_log_loss(K, FLAGS.l2_regularization)

    # The reference validates 16 times per epoch and keeps the best table ("pocket", holE.py:351-360).  Reading the
    # validation loss on the host at every tick would stop the device every 7 steps at FB15k / B=4096 (and a
    # dozen tensor-op launches per tick cost more host time than the 7 steps take), so the tick is one native call
    # (hole.ValidationPocket / ge_validation_tick): batch selection, negatives, hinge, mean and the pocket copy all
    # happen on the device; the host reads the losses and writes the checkpoint FILE once per epoch.  What ends
    # up in the file is what the reference would have saved: the table and global step of the best tick.
    tick = max(1, batch_count // 16)          # guard for the ZeroDivisionError of holE.py:351
    pocket_loss = 2.
    history = []
    state = {'best_logged': 2.0, 'pocket_step': global_step, 'last_write': time.time()}
    vp = None
    if valid is not None:
        vp = H.ValidationPocket(embeddings, valid, tt, FLAGS.batch_size, margin=FLAGS.margin, model=eval_model,
                                seed=FLAGS.seed ^ 0x5EED, capacity=max(64, 2 * (batch_count // tick + 2)),
                                log_loss=(K, FLAGS.l2_regularization) if FLAGS.log_loss else None)

    def validation_tick():
        vp.tick(trainer.global_step, trainer.global_step)

    def drain():
        """Log the validation ticks since the last call, in order (one synchronisation).  The host sees every loss
        in tick order, so it knows which tick the device pocket holds: the first one with the lowest loss."""
        nonlocal pocket_loss
        ticks = vp.read() if vp is not None else []
        for step, vlm in ticks:
            log('\tStep {} Validation Loss: {}...'.format(step, vlm))
            history.append((step, vlm))
            if vlm < pocket_loss:
                pocket_loss = vlm
                state['pocket_step'] = step

    def write_pocket(epoch):
        """The checkpoint file follows the device pocket: once per epoch, every --checkpoint_seconds in between (the
        reference saves at every improving tick, holE.py:357-360: a killed run loses at most that much), and at the end."""
        drain()
        state['last_write'] = time.time()
        pocket = vp.pocket if vp is not None else None
        if pocket is None or pocket_loss >= state['best_logged']:
            return
        state['best_logged'] = pocket_loss
        table = pocket.clone()
        if trainer.spectral:
            H.hole_from_spectral(table)
        save_checkpoint(FLAGS.output_dir, table, state['pocket_step'])
        log('Epoch {}, (Model saved with loss {})'.format(epoch, pocket_loss))

    t_start = time.time()
    done = False
    epoch = 0
    for epoch in range(1, FLAGS.num_epochs + 1):
        log('Training epoch {}...'.format(epoch))
        trainer.reshuffle(gen)
        if epoch < FLAGS.num_epochs:
            trainer.prepare_reshuffle(gen)        # the next epoch's order is drawn beside this epoch's steps
        batch = 1
        while batch < batch_count and not done:
            if batch % tick == 0 and valid is not None:
                validation_tick()
            # steps up to the next validation tick (or the end of the epoch), enqueued natively
            nxt = min(batch_count, (batch // tick + 1) * tick)
            n = nxt - batch
            if FLAGS.max_steps:
                n = min(n, FLAGS.max_steps - (trainer.global_step - global_step))
            if n > 0:
                trainer.run(n)
            if FLAGS.checkpoint_seconds > 0 and time.time() - state['last_write'] >= FLAGS.checkpoint_seconds:
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
    log('Done training -- epoch limit reached')
    trainer.to_real()
    if not os.path.exists(checkpoint_path(FLAGS.output_dir)):
        save_checkpoint(FLAGS.output_dir, embeddings, trainer.global_step)
    steps = trainer.global_step - global_step
    return {'steps': steps, 'seconds': time.time() - t_start, 'pocket_loss': pocket_loss, 'history': history,
            'global_step': trainer.global_step, 'final_mean_hinge': float(trainer.last_loss.mean())}


def save_embeddings(FLAGS, log=print):
    """holE.py:501-527: print every row's complex embedding (clipped, as get_embedding returns it)."""
    data = D.init_inference_data(FLAGS.data_dir, min_mentions=None)
    emb, _ = load_checkpoint(FLAGS.output_dir)
    k = emb.shape[1] // 2
    norm = emb.norm(dim=1, keepdim=True).clamp_min(1.0)
    y = (emb / norm).cpu().numpy()
    for entity in range(data.entity_count):
        log(data.id_to_metadata.get(entity, str(entity)), y[entity, :k] + 1j * y[entity, k:])


def infer_triples(FLAGS, log=print) -> dict:
    """--infer: the ranking / MRR semantics of holE.py:427-490 as a filtered 1-vs-all link-prediction
    evaluation over every entity row (the reference's candidate lists are Diffbot-specific,
    holE.py:534-541)."""
    data = D.init_inference_data(FLAGS.data_dir, min_mentions=None)
    emb, _ = load_checkpoint(FLAGS.output_dir)
    # --model hole: the checkpoint holds the real-valued table; ranks use the HolE score (README.md:42), not ComplEx
    # positions are recorded for confident sweeps only: lowest loss < --infer_threshold (holE.py:436-438, 464-466, 616)
    return E.evaluate_fb15k_style(emb, data, both_sides=True, model=FLAGS.model, infer_threshold=FLAGS.infer_threshold)


def main(argv=None):
    FLAGS, _unparsed = build_parser().parse_known_args(argv)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if FLAGS.gpus > 1 and 'WORLD_SIZE' not in os.environ and not FLAGS.save_embeddings:
        # no launcher: this process (which has not touched the GPU) becomes the parent of --gpus ranks of itself
        from . import launch
        sys.exit(launch.spawn_ranks(FLAGS.gpus, list(sys.argv[1:] if argv is None else argv), module='graphembeddings_amd.train'))
    if world > 1 and FLAGS.gpus != world:
        raise SystemExit(f'--gpus {FLAGS.gpus} but the launcher started {world} ranks')
    if world > 1 and not FLAGS.save_embeddings:
        from . import sharded_train as ST
        if FLAGS.infer:
            ST.infer_sharded(FLAGS)
        else:
            training_data = D.init_data(FLAGS.data_dir, cache=True)
            if int(os.environ.get('RANK', '0')) == 0:
                print('Entities: ', training_data.entity_count - training_data.relation_count, 'Relations: ',
                      training_data.relation_count, 'Triples: ', training_data.triple_count)
            ST.run_training_sharded(training_data, FLAGS)
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return
    if FLAGS.save_embeddings:
        save_embeddings(FLAGS)
    elif FLAGS.infer:
        infer_triples(FLAGS)
    else:
        training_data = D.init_data(FLAGS.data_dir, cache=True)
        print('Entities: ', training_data.entity_count - training_data.relation_count, 'Relations: ',
              training_data.relation_count, 'Triples: ', training_data.triple_count)
        run_training(training_data, FLAGS)


if __name__ == '__main__':
    main(sys.argv[1:])
