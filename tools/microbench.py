#!/usr/bin/env python3
"""Kernel microbenchmarks (not the bench.py contract): forward score / 1vK / HolE throughput vs the
HBM roofline at launch sizes large enough to be bandwidth-bound.  Prints one JSON object per case."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphembeddings_amd import hole as H  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = H.Events(2)
    ev.record(0)
    for _ in range(iters):
        fn()
    ev.record(1)
    torch.cuda.synchronize()
    ms = ev.elapsed_ms(0, 1) / iters
    ev.close()
    return ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="score,hole,1vk,step")
    args = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(0)
    d = 200
    for N, tag in ((16296, "fb15k 13MB (L2/MALL-resident)"), (1_200_018, "1.2M rows 960MB (HBM)")):
        table = (torch.randn(N, d, device="cuda", generator=g) * 0.05)
        for B in (4096, 65536, 1 << 20, 1 << 22):
            tr = torch.randint(0, N, (B, 3), device="cuda", generator=g, dtype=torch.int32)
            if "score" in args.cases:
                out = torch.empty(B, device="cuda")
                ms = timeit(lambda: H.evaluate_triples(tr, table))
                gbs = (12 * d + 16) * B / ms / 1e6
                print(json.dumps({"kernel": "complex_score", "table": tag, "B": B, "ms": round(ms, 4),
                                  "Mtriples_s": round(B / ms / 1e3, 1), "alg_GBs": round(gbs, 1),
                                  "frac_8TBs": round(gbs / 8000, 3)}), flush=True)
            if "hole" in args.cases and B <= (1 << 20):
                ms = timeit(lambda: H.evaluate_triples(tr, table, model="hole"), iters=5, warm=1)
                print(json.dumps({"kernel": "hole_score", "table": tag, "B": B, "ms": round(ms, 4),
                                  "Mtriples_s": round(B / ms / 1e3, 1)}), flush=True)
            if "step" in args.cases and B <= (1 << 20):
                neg = tr.clone()
                neg[:, 1] = torch.randint(0, N, (B,), device="cuda", generator=g, dtype=torch.int32)
                opt = H.HingeSGD(table.clone(), B)
                ms = timeit(lambda: opt.step(tr, neg, 0.01), iters=10, warm=2)
                gbs = (72 * d + 28) * B / ms / 1e6
                print(json.dumps({"kernel": "complex_hinge_step(uniform ids)", "table": tag, "B": B, "ms": round(ms, 4),
                                  "Mscored_s": round(2 * B / ms / 1e3, 1), "alg_GBs": round(gbs, 1),
                                  "frac_8TBs": round(gbs / 8000, 3)}), flush=True)
                del opt
        if "1vk" in args.cases:
            for (B, K) in ((4096, 256), (4096, 16384), (59071, 14951)):
                hr = torch.stack([torch.randint(0, N, (B,), device="cuda", generator=g),
                                  torch.randint(0, min(N, 1345), (B,), device="cuda", generator=g)], 1).int()
                cand = torch.randint(0, N, (K,), device="cuda", generator=g, dtype=torch.int32)
                ms = timeit(lambda: H.score_candidates(table, hr, cand), iters=5, warm=1)
                tf = 2.0 * B * K * d / ms / 1e9
                print(json.dumps({"kernel": "score_1vK", "table": tag, "B": B, "K": K, "ms": round(ms, 4),
                                  "TFLOPs": round(tf, 2), "frac_157TF": round(tf / 157.3, 3),
                                  "Gscores_s": round(B * K / ms / 1e6, 2)}), flush=True)
        del table


if __name__ == "__main__":
    main()
