"""ge_complex_score_1vK at sweep shapes: ms and TFLOP/s (fp32 MFMA), output [B,K] fp32 written to HBM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from graphembeddings_amd import hole as H
N, R = 16296, 1345
g = torch.Generator(device="cuda").manual_seed(0)
c = torch.arange(R, N, dtype=torch.int32, device="cuda")
for d, B in ((200, 16384), (200, 59071), (128, 16384), (104, 16384), (200, 4096)):
    emb = torch.randn(N, d, device="cuda", generator=g) * 0.1
    hr = torch.stack([torch.randint(R, N, (B,), device="cuda", generator=g), torch.randint(0, R, (B,), device="cuda", generator=g)], 1).int()
    cc = c if B != 4096 else c[:256]
    H.score_candidates(emb, hr, cc)
    ev = H.Events(2); ev.record(0)
    for _ in range(3):
        out = H.score_candidates(emb, hr, cc)
    ev.record(1); torch.cuda.synchronize()
    ms = ev.elapsed_ms(0, 1) / 3; ev.close()
    K = cc.numel()
    print(f"d={d} B={B} K={K}: {ms:.3f} ms  {2.0*B*K*d/(ms*1e-3)/1e12:.1f} TFLOP/s  {B*K*4/(ms*1e-3)/1e9:.0f} GB/s written")
    del out
