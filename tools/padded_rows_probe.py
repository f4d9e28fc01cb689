"""SURVEY.md 7 asked for it: does padding the 800-byte rows of a d=200 table (so that they start on 64-byte sector /
128-byte line boundaries) help the north-star kernel?  Times ge_complex_score (dense, ld = 200) against
ge_complex_score_strided on the same 1.2 M rows stored ld = 208 (832 B: every row sector-aligned) and ld = 224
(896 B: line-aligned), 4 M random triples per launch.  Algorithmic bytes are the same 12d+16 per triple."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphembeddings_amd import hole as H, _lib

N, d, T = 1_200_018, 200, 1 << 22
g = torch.Generator(device="cuda").manual_seed(0)
dense = torch.randn(N, d, device="cuda", generator=g) * 0.05
tr = torch.randint(0, N, (T, 3), device="cuda", generator=g, dtype=torch.int32)
out = torch.empty(T, device="cuda")
st = torch.cuda.current_stream().cuda_stream
ref = None
for ld in (200, 208, 224, 256):
    tab = torch.zeros(N, ld, device="cuda")
    tab[:, :d] = dense
    call = lambda: _lib.call("ge_complex_score_strided", tab.data_ptr(), N, d, ld, tr.data_ptr(), T, 1.0, 1, out.data_ptr(), st)
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    ev = H.Events(2)
    ev.record(0)
    for _ in range(10):
        call()
    ev.record(1)
    torch.cuda.synchronize()
    ms = ev.elapsed_ms(0, 1) / 10
    if ref is None:
        ref = out.clone()
    same = bool(torch.equal(ref, out))
    print(json.dumps({"ld": ld, "row_bytes": 4 * ld, "table_mb": round(N * ld * 4 / 1e6), "kernel_ms": round(ms, 4),
                      "algorithmic_GBs": round((12 * d + 16) * T / ms / 1e6, 1), "same_scores_as_dense": same}), flush=True)
    del tab
