"""Times ge_hole_to_spectral / ge_hole_from_spectral (in-place row DFTs of a HolE table) on the FB15k-sized and the
1.2 M-row table, and checks the round trip."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphembeddings_amd import hole as H
for N in (16296, 1_200_018):
    t = torch.randn(N, 200, device="cuda") * 0.07
    ref = t.clone()
    H.hole_to_spectral(t); H.hole_from_spectral(t)
    torch.cuda.synchronize()
    err = float((t - ref).abs().max())
    ev = H.Events(3)
    ev.record(0)
    for _ in range(5):
        H.hole_to_spectral(t)
    ev.record(1)
    for _ in range(5):
        H.hole_from_spectral(t)
    ev.record(2)
    torch.cuda.synchronize()
    print(json.dumps({"rows": N, "d": 200, "to_spectral_ms": round(ev.elapsed_ms(0, 1) / 5, 4), "from_spectral_ms": round(ev.elapsed_ms(1, 2) / 5, 4),
                      "round_trip_max_abs_err": err}), flush=True)
