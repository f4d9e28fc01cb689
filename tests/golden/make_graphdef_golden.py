#!/usr/bin/env python3
"""Generates tests/golden/graphdef_v1.npz and graphdef_facts.json from the reference's OWN recorded
training graph, /root/reference/holE-20170724/graph.pbtxt (only in the build container; the
reference never travels).  Nothing of the reference's text is stored: the .npz holds seeded inputs
and the outputs obtained by EXECUTING the recorded nodes with oracle/graphdef.py (forward, the
batch/gradients/* autodiff sub-graph, the IndexedSlices concat, the ScatterSub), the .json holds the
handful of structural facts (operand orders, constants, attributes) the oracle's decisions rest on.

    python tests/golden/make_graphdef_golden.py

The recorded graph computes the HISTORICAL score variant (complex FFT correlation, Sum(Re+Im), tanh;
graph.pbtxt:6221-6521) around exactly the machinery today's holE.py still uses: embedding_lookup with
max_norm (clip_by_norm chain), the pairwise hinge, minimize() of the loss VECTOR, IndexedSlices concat
and ScatterSub.  oracle/hole_oracle.py restates that variant as model="graph20170724" on top of the
SAME helpers (clip_scale, _clip_backward, the hinge mask, sgd_step) that its ComplEx / HolE models use.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import graphdef as GD  # noqa: E402

REF = "/root/reference/holE-20170724/graph.pbtxt"
REF_0714 = "/root/reference/holE-20170714/graph.pbtxt"
HERE = os.path.dirname(os.path.abspath(__file__))
FETCH = ["batch/Maximum", "batch/gradients/concat_1", "batch/GradientDescent/update_embeddings/mul",
         "batch/GradientDescent/update_embeddings/ScatterSub", "batch/InverseTimeDecay", "batch/train/eval/Tanh"]
N, D, B = 300, 128, 512          # B and d are baked into the recorded graph; the table height is free


def inputs(seed):
    rng = np.random.default_rng(seed)
    table = rng.standard_normal((N, D)) * 0.08
    table[::3] *= 4.0                                  # a third of the rows outside the unit ball
    table[12] = 0.0
    table[12, 0] = 1.0                                 # |x| == max_norm exactly: the Minimum tie
    table[15] = 0.0                                    # all-zero row (rsqrt(0) = inf); only the zero_row case gathers it
    pos = np.stack([rng.integers(20, N, B), rng.integers(20, N, B), rng.integers(0, 10, B)], 1).astype(np.int32)
    pos[0] = [11, 12, 3]
    pos[1] = [12, 13, 3]
    neg = pos.copy()
    col = seed % 2                                     # whole batch corrupts one side (holE.py:137-140)
    neg[:, col] = rng.integers(20, N, B)
    neg[3] = pos[3]                                    # corrupted == original: pre-activation == margin
    neg[4] = pos[4]
    return table, pos, neg


def cases():
    # (name, seed, margin override or None = the recorded constant, global step, touch the zero row?)
    return [("recorded_margin", 1, None, 7, False), ("margin_0p05", 2, 0.05, 100, False),
            ("margin_0_ties", 3, 0.0, 4096, False), ("zero_row", 4, 0.2, 0, True)]


def run_case(g, seed, margin, step, zero_row):
    table, pos, neg = inputs(seed)
    if zero_row:
        pos[2] = [15, 16, 1]                           # the all-zero row as a head
        neg[2] = [15, 17, 1] if seed % 2 else [18, 16, 1]
    feeds = {"embeddings": table, "batch/shuffle_batch": pos, "batch/corrupt/cond/Merge": neg,
             "batch/Variable": np.int32(step)}
    if margin is not None:
        feeds["batch/add/y"] = np.float64(margin)
    with np.errstate(invalid="ignore"):
        loss, idx, upd, new, lr, act = g.run(FETCH, feeds)
    w = np.random.default_rng(99).standard_normal(D)
    return {"table": table, "pos": pos, "neg": neg, "loss": loss[:, 0], "idx": idx, "upd_proj": upd @ w,
            "upd_abs_sum": np.abs(upd).sum(1), "new_table": new, "lr": np.float64(lr), "act_pos": act[:, 0],
            "margin": np.float64(g["batch/add/y"].a("value") if margin is None else margin), "step": np.int64(step)}


def facts(g, g0714):
    def node(name):
        n = g[name]
        return {"op": n.op, "inputs": n.inputs}
    def const(name, gg=g):
        v = gg[name].a("value")
        return v.tolist() if hasattr(v, "tolist") else v
    f = {"source": "holE-20170724/graph.pbtxt", "producer_nodes": len(g.nodes)}
    lookups = {"batch/train/embedding_lookup": "h+", "batch/train/embedding_lookup_1": "t+", "batch/train/embedding_lookup_2": "r+",
               "batch/corrupt/embedding_lookup": "h-", "batch/corrupt/embedding_lookup_1": "t-", "batch/corrupt/embedding_lookup_2": "r-"}
    f["lookup_ids"] = {v: g[k].inputs[1] for k, v in lookups.items()}
    f["id_columns"] = {nm: const(g[nm].inputs[1])for nm in ("batch/train/h_id", "batch/train/t_id", "batch/train/r_id")}
    chain = {}
    base = "batch/train/embedding_lookup/clip_by_norm"
    for suffix in ("/mul", "/Sum", "/Rsqrt", "/mul_1", "/truediv", "/Minimum", "/mul_2", ""):
        n = g[base + suffix]
        chain[suffix or "/"] = {"op": n.op, "inputs": [i.replace(base, "~").replace("batch/train/embedding_lookup", "X") for i in n.inputs]}
    f["clip_chain"] = chain
    f["clip_consts"] = {"mul_1/y": const(base + "/mul_1/y"), "Const": const(base + "/Const"), "truediv/y": const(base + "/truediv/y"),
                        "Sum/reduction_indices": const(base + "/Sum/reduction_indices"), "Sum/keep_dims": bool(g[base + "/Sum"].a("keep_dims"))}
    f["clip_chain_identical_for_all_six"] = all(
        [g[k + "/clip_by_norm" + s].op for s in ("/mul", "/Sum", "/Rsqrt", "/mul_1", "/truediv", "/Minimum", "/mul_2")] ==
        [g[base + s].op for s in ("/mul", "/Sum", "/Rsqrt", "/mul_1", "/truediv", "/Minimum", "/mul_2")] for k in lookups)
    mg = "batch/gradients/" + base + "/Minimum_grad"
    f["minimum_grad"] = {"compare": node(mg + "/LessEqual"), "select_x": node(mg + "/Select"), "select_y": node(mg + "/Select_1")}
    f["split_real_imag"] = {"reshape": const("batch/train/Reshape/shape"), "re_begin": const("batch/train/Slice/begin"),
                            "re_size": const("batch/train/Slice/size"), "im_begin": const("batch/train/Slice_1/begin"),
                            "im_size": const("batch/train/Slice_1/size"), "complex": node("batch/train/Complex")}
    f["score_chain"] = {k: node("batch/train/eval/" + k) for k in ("FFT", "Conj", "FFT_1", "Mul", "IFFT", "Mul_1", "Real", "Imag", "add", "Sum", "Tanh")}
    f["hinge"] = {"sub": node("batch/sub"), "add": node("batch/add"), "margin_recorded": const("batch/add/y"),
                  "maximum": node("batch/Maximum"), "maximum_y": const("batch/Maximum/y")}
    f["grad_seed"] = {"fill": node("batch/gradients/Fill"), "value": const("batch/gradients/Const"),
                      "shape_of": g["batch/gradients/Shape"].inputs}
    f["maximum_grad"] = {"compare": node("batch/gradients/batch/Maximum_grad/GreaterEqual"), "select": node("batch/gradients/batch/Maximum_grad/Select")}
    f["concat_indices"] = g["batch/gradients/concat_1"].inputs
    f["concat_values"] = g["batch/gradients/concat"].inputs
    ss = g["batch/GradientDescent/update_embeddings/ScatterSub"]
    f["scatter_sub"] = {"inputs": ss.inputs, "use_locking": bool(ss.a("use_locking")), "update": node("batch/GradientDescent/update_embeddings/mul")}
    f["embeddings_shape"] = [d["size"][0] for d in g["embeddings"].a("shape")["dim"]]
    dec = "batch/InverseTimeDecay"
    for tag, gg in (("decay_20170724", g), ("decay_20170714", g0714)):
        f[tag] = {"chain": {s or "/": {"op": gg[dec + s].op, "inputs": gg[dec + s].inputs} for s in ("/Cast", "/Cast_1", "/truediv", "/Cast_3", "/Mul", "/Add", "")},
                  "learning_rate": const(dec + "/learning_rate", gg), "decay_steps": const(dec + "/Cast_1/x", gg),
                  "decay_rate": const(dec + "/Cast_2/x", gg), "one": const(dec + "/Const", gg),
                  "embeddings_shape": [d["size"][0] for d in gg["embeddings"].a("shape")["dim"]],
                  "margin_recorded": const("batch/add/y", gg), "batch": const("batch/train/h_id/size", gg)}
    return f


def main():
    g = GD.Graph(REF)
    g0714 = GD.Graph(REF_0714)
    out = {}
    for name, seed, margin, step, zero_row in cases():
        for k, v in run_case(g, seed, margin, step, zero_row).items():
            out[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "graphdef_v1.npz"), **out)
    with open(os.path.join(HERE, "graphdef_facts.json"), "w") as f:
        json.dump(facts(g, g0714), f, indent=1, sort_keys=True)
    print("wrote graphdef_v1.npz", os.path.getsize(os.path.join(HERE, "graphdef_v1.npz")), "bytes and graphdef_facts.json")


if __name__ == "__main__":
    main()
