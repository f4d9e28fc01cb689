"""HolE forward score: the hand-written direct-correlation kernel vs the FFT formulation of README.md:42
(ifft(conj(fft(h)) * fft(t))) evaluated with torch.fft (rocFFT) on the same GPU, same rows already gathered
and clipped for the FFT path (i.e. the FFT path is given a head start: no gather, no clip, no fusion)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from graphembeddings_amd import hole as H

N, d = 16296, 200
emb = H.init_embeddings(N, d)
g = torch.Generator().manual_seed(0)


def timeit(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


for B in (4096, 65536, 1 << 20):
    tri = torch.stack([torch.randint(1345, N, (B,), generator=g), torch.randint(1345, N, (B,), generator=g),
                       torch.randint(0, 1345, (B,), generator=g)], 1).int().cuda()
    reps = 200 if B <= 65536 else 20
    t_direct, s_direct = timeit(lambda: H.evaluate_triples(tri, emb, model="hole", apply_sigmoid=False), reps)
    idx = tri.long()
    h, t, r = emb[idx[:, 0]], emb[idx[:, 1]], emb[idx[:, 2]]
    clip = lambda x: x * torch.clamp(torch.rsqrt((x * x).sum(1, keepdim=True)), max=1.0)
    h, t, r = clip(h), clip(t), clip(r)
    fft_score = lambda: (r * torch.fft.irfft(torch.conj(torch.fft.rfft(h, dim=1)) * torch.fft.rfft(t, dim=1), n=d, dim=1)).sum(1)
    t_fft, s_fft = timeit(fft_score, reps)
    err = (s_direct[:, 0] - s_fft).abs().max().item()
    print(json.dumps({"B": B, "direct_kernel_us": round(t_direct * 1e6, 1), "torch_rocfft_us": round(t_fft * 1e6, 1),
                      "ratio_fft_over_direct": round(t_fft / t_direct, 2), "max_abs_diff": err}))
