"""Times ge_train_steps at large batch sizes on the 1.2 M-row (960 MB) table: per-step time, grad / apply kernel
times (HIP events on the dispatch), algorithmic bytes per step (72d+28 per pair) against the 8 TB/s peak."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import os
if os.environ.get("GE_LIB"):       # experiments only (tools/dev/build_variant.py): time a variant build of the library
    from graphembeddings_amd import _lib as _L
    _L.LIB_PATH = os.path.abspath(os.environ["GE_LIB"])
from graphembeddings_amd import data as D
from graphembeddings_amd import hole as H


def main():
    d = 200
    n_ent = int(sys.argv[1]) if len(sys.argv) > 1 else 1_200_000
    batches = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4096, 16384, 65536]
    zipf_s = float(sys.argv[3]) if len(sys.argv) > 3 else 0.8       # 0 = uniform heads / tails (no hot rows)
    margin = float(sys.argv[4]) if len(sys.argv) > 4 else 0.2       # -1 = no pair is hinge-active (the update kernel's floor)
    data, tri = D.synthetic_large(n_entities=n_ent, n_triples=4_000_000, seed=1234, zipf_s=zipf_s)
    names, id_to_type, offsets, ids = D.synthetic_large_type_arrays(data)
    tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
    dtri = torch.as_tensor(tri).cuda()
    for B in batches:
        emb = H.init_embeddings(data.entity_count, d, seed=0)
        tr = H.Trainer(emb, dtri, tt, B, margin=margin, learning_rate=0.1, decay_steps=1e5, seed=0)
        steps = max(8, min(64, (1 << 21) // B))
        tr.run(steps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.run(steps)
        t_host = (time.perf_counter() - t0) / steps      # the host's enqueue time: close to us_per_step = launch-bound
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps
        out = {"B": B, "zipf_s": zipf_s, "margin": margin, "us_per_step": el * 1e6, "host_enqueue_us_per_step": t_host * 1e6, "scored_per_s": 2 * B / el,
               "step_alg_frac_of_8TBs": (72 * d + 28) * B / el / 8e12}
        for kern, name in ((1, "grad_us"), (2, "apply_us")):
            ev = H.Events(2 * steps)
            tr.run(steps, events=ev.handles, ev_kernel=kern)
            torch.cuda.synchronize()
            out[name] = 1e3 * float(np.median([ev.elapsed_ms(2 * i, 2 * i + 1) for i in range(steps)]))
            ev.close()
        out["mean_loss"] = float(tr.last_loss.mean())
        print(json.dumps(out), flush=True)
        tr.close()
        del emb, tr


if __name__ == "__main__":
    main()
