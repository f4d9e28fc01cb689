"""Pins the oracle against the reference's OWN recorded training graph (holE-20170724/graph.pbtxt).

tests/golden/graphdef_v1.npz holds seeded inputs and the outputs obtained by executing the recorded
GraphDef node by node (oracle/graphdef.py; generator tests/golden/make_graphdef_golden.py): the
forward pass, TensorFlow's own autodiff sub-graph, the IndexedSlices concat and the ScatterSub.
oracle/hole_oracle.py restates the graph's (historical) score as model="graph20170724" on top of the
same clip / hinge / scatter helpers its ComplEx and HolE models use, and must reproduce every vector.
graphdef_facts.json holds the operand orders, constants and attributes the oracle's decisions cite.
When /root/reference is present (build container) the fixture is also re-derived from the file.
"""
import json
import os

import numpy as np
import pytest

from oracle import hole_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/holE-20170724/graph.pbtxt"
CASES = ("recorded_margin", "margin_0p05", "margin_0_ties", "zero_row")


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(HERE, "golden", "graphdef_v1.npz"))


@pytest.fixture(scope="module")
def F():
    return json.load(open(os.path.join(HERE, "golden", "graphdef_facts.json")))


def test_clip_chain_wiring(F):
    """tf.nn.embedding_lookup(max_norm=1) as recorded (graph.pbtxt:3108-3596): y = (x * c) * min(rsqrt(sum x^2), 1/c),
    sum over every axis but the first with keep_dims, Minimum(x = rsqrt, y = 1/c) in that operand order,
    and MinimumGrad = Select(LessEqual(rsqrt, 1/c)): at |x| == c the gradient goes to the rsqrt branch."""
    ch = F["clip_chain"]
    assert ch["/mul"] == {"op": "Mul", "inputs": ["X", "X"]}
    assert ch["/Sum"]["op"] == "Sum" and ch["/Sum"]["inputs"][0] == "~/mul"
    assert F["clip_consts"]["Sum/reduction_indices"] == [1, 2] and F["clip_consts"]["Sum/keep_dims"] is True
    assert ch["/Rsqrt"] == {"op": "Rsqrt", "inputs": ["~/Sum"]}
    assert ch["/mul_1"] == {"op": "Mul", "inputs": ["X", "~/mul_1/y"]}
    assert ch["/truediv"] == {"op": "RealDiv", "inputs": ["~/Const", "~/truediv/y"]}
    assert ch["/Minimum"] == {"op": "Minimum", "inputs": ["~/Rsqrt", "~/truediv"]}
    assert ch["/mul_2"] == {"op": "Mul", "inputs": ["~/mul_1", "~/Minimum"]}
    assert ch["/"] == {"op": "Identity", "inputs": ["~/mul_2"]}
    assert F["clip_consts"]["mul_1/y"] == 1.0 and F["clip_consts"]["Const"] == 1.0 and F["clip_consts"]["truediv/y"] == 1.0
    assert F["clip_chain_identical_for_all_six"] is True
    mg = F["minimum_grad"]
    assert mg["compare"]["op"] == "LessEqual"
    assert [i.rsplit("/", 1)[1] for i in mg["compare"]["inputs"]] == ["Rsqrt", "truediv"]
    assert mg["select_x"]["inputs"][0].endswith("Minimum_grad/LessEqual") and mg["select_y"]["inputs"][0].endswith("Minimum_grad/LogicalNot")
    # the oracle's statement of the same thing
    x = np.array([[3.0, 4.0], [0.6, 0.8], [0.3, 0.4], [0.0, 0.0]])
    assert np.allclose(O.clip_scale(x)[:, 0], [0.2, 1.0, 1.0, 1.0])
    gy = np.ones_like(x)
    gx = O._clip_backward(x, gy)
    assert np.allclose(gx[2], gy[2])                                     # inside the ball: identity
    assert np.allclose(gx[1], gy[1] - x[1] * (gy[1] @ x[1]))            # |x| == 1: the rsqrt branch (LessEqual)


def test_triple_columns_and_complex_split(F):
    """holE.py:181-185 / 161-168 as recorded: columns (head, tail, relation); first half of a row = Re, second = Im."""
    assert F["id_columns"] == {"batch/train/h_id": [0, 0], "batch/train/t_id": [0, 1], "batch/train/r_id": [0, 2]}
    sp = F["split_real_imag"]
    assert sp["reshape"] == [-1, 128] and sp["re_begin"] == [0, 0] and sp["re_size"] == [-1, 64]
    assert sp["im_begin"] == [0, 64] and sp["im_size"] == [-1, 64]
    assert sp["complex"]["op"] == "Complex" and sp["complex"]["inputs"][0].endswith("/Slice")


def test_hinge_seed_concat_and_scatter_wiring(F):
    """holE.py:231 + 296 as recorded: max((E(pos) - E(neg)) + margin, 0); MaximumGrad mask is >=; the gradient
    seed is ones of the loss shape (gradient of the SUM of the loss vector); the IndexedSlices are
    concatenated r+, r-, t+, t-, h+, h- and applied by ONE ScatterSub(embeddings, indices, lr * values),
    use_locking = false (graph.pbtxt:47850-48001)."""
    h = F["hinge"]
    assert h["sub"] == {"op": "Sub", "inputs": ["batch/train/eval/Tanh", "batch/corrupt/eval/Tanh"]}
    assert h["add"] == {"op": "Add", "inputs": ["batch/sub", "batch/add/y"]}
    assert h["maximum"] == {"op": "Maximum", "inputs": ["batch/add", "batch/Maximum/y"]} and h["maximum_y"] == 0.0
    assert h["margin_recorded"] == 1.0
    assert F["maximum_grad"]["compare"] == {"op": "GreaterEqual", "inputs": ["batch/add", "batch/Maximum/y"]}
    assert F["maximum_grad"]["select"]["inputs"][1] == "batch/gradients/Fill"
    assert F["grad_seed"]["fill"]["op"] == "Fill" and F["grad_seed"]["value"] == 1.0
    order = [i.split("/")[3] + "/" + i.split("/")[4].replace("_grad", "") for i in F["concat_indices"][:6]]
    assert order == ["train/embedding_lookup_2", "corrupt/embedding_lookup_2", "train/embedding_lookup_1",
                     "corrupt/embedding_lookup_1", "train/embedding_lookup", "corrupt/embedding_lookup"]
    assert F["lookup_ids"] == {"h+": "batch/train/h_id", "t+": "batch/train/t_id", "r+": "batch/train/r_id",
                               "h-": "batch/corrupt/h_id", "t-": "batch/corrupt/t_id", "r-": "batch/corrupt/r_id"}
    ss = F["scatter_sub"]
    assert ss["inputs"] == ["embeddings", "batch/gradients/concat_1", "batch/GradientDescent/update_embeddings/mul"]
    assert ss["use_locking"] is False
    assert ss["update"] == {"op": "Mul", "inputs": ["batch/gradients/concat", "batch/InverseTimeDecay"]}


def test_decay_chain_and_recorded_constants(F):
    """tf.train.inverse_time_decay as recorded: lr / (1 + rate * (float(step) / float(decay_steps))), plus the
    run constants of both dumps (holE.py:291-295)."""
    for tag in ("decay_20170714", "decay_20170724"):
        c = F[tag]["chain"]
        assert c["/truediv"] == {"op": "RealDiv", "inputs": ["batch/InverseTimeDecay/Cast", "batch/InverseTimeDecay/Cast_1"]}
        assert c["/Mul"] == {"op": "Mul", "inputs": ["batch/InverseTimeDecay/Cast_2/x", "batch/InverseTimeDecay/truediv"]}
        assert c["/Add"] == {"op": "Add", "inputs": ["batch/InverseTimeDecay/Cast_3", "batch/InverseTimeDecay/Mul"]}
        assert c["/"] == {"op": "RealDiv", "inputs": ["batch/InverseTimeDecay/learning_rate", "batch/InverseTimeDecay/Add"]}
        assert c["/Cast"]["inputs"] == ["batch/Variable/read"] and F[tag]["one"] == 1
        assert F[tag]["decay_rate"] == 0.5 and abs(F[tag]["learning_rate"] - 0.01) < 1e-9 and F[tag]["margin_recorded"] == 1.0
    assert F["decay_20170714"]["decay_steps"] == 6192 and F["decay_20170714"]["embeddings_shape"] == [35910, 64]
    assert F["decay_20170724"]["embeddings_shape"] == [1134637, 128] == F["embeddings_shape"]
    assert F["decay_20170724"]["batch"] == [-1, 1]


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_the_executed_graph(G, F, case):
    """loss vector, IndexedSlices (indices exactly, values through two projections) and the table after the
    ScatterSub, fp64, 1e-12: the oracle's clip forward/backward, hinge mask, SUM seed, slot order and
    duplicate accumulation ARE the recorded graph's (N = 300 rows for 3,072 slots: every row is hit many times)."""
    g = lambda k: G[f"{case}/{k}"]
    table, pos, neg, margin = g("table"), g("pos"), g("neg"), float(g("margin"))
    lr = O.inverse_time_decay(F["decay_20170724"]["learning_rate"], int(g("step")), F["decay_20170724"]["decay_steps"],
                              F["decay_20170724"]["decay_rate"])
    assert abs(lr - float(g("lr"))) < 1e-15
    assert np.abs(O.hole_graph20170724_evaluate(pos, table)[:, 0] - g("act_pos")).max() < 1e-13
    idx, val, loss = O.hinge_grads(pos, neg, table, margin=margin, model="graph20170724")
    assert np.abs(loss - g("loss")).max() < 1e-13
    assert np.array_equal(idx, g("idx"))
    new, _ = O.sgd_step(table, pos, neg, lr=lr, margin=margin, model="graph20170724")
    w = np.random.default_rng(99).standard_normal(table.shape[1])
    ok = np.isfinite(g("upd_proj"))
    if case != "zero_row":
        assert ok.all() and np.isfinite(g("new_table")).all()
    assert np.abs((lr * val) @ w - g("upd_proj"))[ok].max() < 1e-12
    assert np.abs(np.abs(lr * val).sum(1) - g("upd_abs_sum"))[ok].max() < 1e-12
    fin = np.isfinite(g("new_table"))
    assert np.abs(new - g("new_table"))[fin].max() < 1e-12
    assert (g("loss") == 0).sum() == (loss == 0).sum()
    if case == "margin_0_ties":
        # corrupted == original at margin 0: pre-activation exactly 0, GreaterEqual is TRUE, both sides get
        # (opposite) non-zero slices
        tie = np.nonzero((pos == neg).all(1))[0]
        assert len(tie) >= 2 and (g("loss")[tie] == 0).all()
        B = len(pos)
        assert (g("upd_abs_sum")[tie] > 0).all() and (g("upd_abs_sum")[B + tie] > 0).all()
    if case == "recorded_margin":
        assert margin == 1.0 and (loss > 0).all()


def test_all_zero_row_quirk(G):
    """Decided quirk: for an all-zero row the recorded graph produces NaN -- MinimumGrad hands the rsqrt branch a
    zero gradient and RsqrtGrad multiplies it by rsqrt(0)^3 = inf -- so TensorFlow would write NaN into that
    table row.  The oracle (and the kernels) treat the inactive branch's gradient as exactly zero and keep the
    row finite.  Xavier-initialised tables never hold such a row; every other row agrees (test above)."""
    new_g = G["zero_row/new_table"]
    bad = np.unique(np.nonzero(~np.isfinite(new_g))[0])
    assert bad.tolist() == [15]                                          # the one all-zero row, nothing else
    table, pos, neg = G["zero_row/table"], G["zero_row/pos"], G["zero_row/neg"]
    new, _ = O.sgd_step(table, pos, neg, lr=float(G["zero_row/lr"]), margin=float(G["zero_row/margin"]), model="graph20170724")
    assert np.isfinite(new).all()


@pytest.mark.skipif(not os.path.exists(REF), reason="the reference tree exists only in the build container")
def test_fixture_is_what_the_reference_file_yields(G, F):
    """Re-derive facts and vectors from /root/reference/holE-20170724/graph.pbtxt and compare with the committed fixture."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(HERE, "golden", "make_graphdef_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    g, g0 = mk.GD.Graph(mk.REF), mk.GD.Graph(mk.REF_0714)
    assert json.loads(json.dumps(mk.facts(g, g0))) == F
    for name, seed, margin, step, zero_row in mk.cases():
        got = mk.run_case(g, seed, margin, step, zero_row)
        for k, v in got.items():
            assert np.array_equal(np.asarray(v), G[f"{name}/{k}"], equal_nan=True), (name, k)
    # the recorded graph really is a HolE-by-FFT graph around a hinge, nothing else feeds the loss
    ops = {n.op for n in g.nodes.values()}
    assert {"FFT", "IFFT", "Conj", "Tanh", "ScatterSub", "Gather", "Minimum", "Maximum"} <= ops


def test_helpers_shared_between_the_pinned_variant_and_the_product_models():
    """The pinned model differs from ComplEx / HolE only inside _side_grads' score-gradient branch and the
    activation: same clip (forward + backward), hinge mask, slot order and scatter.  Guard that wiring."""
    import inspect
    src = inspect.getsource(O._side_grads)
    assert src.count("_clip_backward(") == 3 and src.count("clip_scale(") == 3
    src = inspect.getsource(O.hinge_grads)
    assert "pre >= 0" in src and src.count("_side_grads(") == 2
    assert "np.subtract.at" in inspect.getsource(O.sgd_step)
