#!/bin/bash
# On the GPU box: the headline step (config 2) with the product library and each variant named, alternated twice.
OUT=gpurun_out/ab2_$1.log; shift
: > $OUT
cat > /tmp/c2probe.py <<'PY'
import os, sys, time, json
sys.path.insert(0, ".")
if os.environ.get("GE_LIB"):
    from graphembeddings_amd import _lib as _L
    _L.LIB_PATH = os.path.abspath(os.environ["GE_LIB"])
import numpy as np, torch
from graphembeddings_amd import data as D, hole as H
fb = D.fb15k_shape()
names, id_to_type, offsets, ids = fb.type_arrays()
tt = H.TypeTables.from_host(id_to_type, offsets, ids, padded_size=1024)
tri = torch.as_tensor(D.synthetic_fb15k_triples(fb, n_triples=483142, seed=0)).cuda()
emb = H.init_embeddings(fb.entity_count, 200, seed=0)
tr = H.Trainer(emb, tri, tt, 4096, margin=0.2, learning_rate=0.1, decay_steps=32.0 * 117, seed=0)
tr.run(200); torch.cuda.synchronize()
res = []
for rep in range(5):
    t0 = time.perf_counter(); tr.run(400); torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 400 * 1e6)
out = {"us_per_step": float(np.median(res)), "min": min(res)}
for k, name in ((1, "grad_us"), (2, "apply_us")):
    ev = H.Events(2 * 64); tr.run(64, events=ev.handles, ev_kernel=k); torch.cuda.synchronize()
    out[name] = 1e3 * float(np.median([ev.elapsed_ms(2 * i, 2 * i + 1) for i in range(64)])); ev.close()
out["loss"] = float(tr.last_loss.mean())
print(json.dumps(out))
PY
for rep in 1 2; do
  for t in product "$@"; do
    if [ $t = product ]; then L=""; else L=graphembeddings_amd/_variants/libge_$t.so; fi
    echo "== $t (rep $rep)" >> $OUT
    GE_LIB=$L timeout -k 10 300 python /tmp/c2probe.py 2>/dev/null | grep '^{' >> $OUT || exit 1
  done
done
cat $OUT
