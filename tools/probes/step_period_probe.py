"""Start-to-start period of consecutive steps' gradient kernels inside one native call (HIP events riding on the dispatches, no
profiler attached): where inside a run of the 65,536-pair step the time between kernels goes (chunk boundaries every 8 steps)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graphembeddings_amd import data as D
from graphembeddings_amd import hole as H

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
data, tri = D.synthetic_large(n_entities=1_200_000, n_triples=4_000_000, seed=1234)
names, id_to_type, offsets, ids = D.synthetic_large_type_arrays(data)
tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
dtri = torch.as_tensor(tri).cuda()
emb = H.init_embeddings(data.entity_count, 200, seed=0)
tr = H.Trainer(emb, dtri, tt, B, margin=0.2, learning_rate=0.1, decay_steps=1e5, seed=0)
steps = 48
tr.run(steps)
torch.cuda.synchronize()
ev = H.Events(2 * steps)
tr.run(steps, events=ev.handles, ev_kernel=1)
torch.cuda.synchronize()
period = [1e3 * ev.elapsed_ms(2 * i, 2 * (i + 1)) for i in range(steps - 1)]
dur = [1e3 * ev.elapsed_ms(2 * i, 2 * i + 1) for i in range(steps)]
print(json.dumps({"B": B, "grad_start_to_next_grad_start_us": [round(p, 1) for p in period], "grad_us": [round(x, 1) for x in dur],
                  "median_period": float(np.median(period)), "mean_period": float(np.mean(period))}))
