"""A NumPy interpreter for the reference's recorded TensorFlow GraphDefs -- TEST INFRASTRUCTURE.

The reference cannot run here (no TensorFlow), but it ships what TensorFlow built from holE.py:
`holE-20170714/graph.pbtxt` and `holE-20170724/graph.pbtxt`, the complete training graphs (forward,
the autodiff sub-graph `batch/gradients/*`, the IndexedSlices concat and the ScatterSub of
GradientDescentOptimizer) as text protobufs.  This module reads that text format (hand-rolled, no
protobuf / TF import) and EXECUTES the recorded nodes with NumPy, op by op, following TensorFlow's
documented op semantics.  The op wiring, operand order, constants and attributes all come from the
reference's own file; only the ~50 primitive op definitions are restated here (each a few lines).

Used by tests/golden/make_graphdef_golden.py (in this container, where /root/reference exists) to
generate golden vectors -- loss vector, IndexedSlices and updated table for seeded inputs -- that pin
oracle/hole_oracle.py's clip chain (forward and backward incl. the norm == max_norm tie), hinge mask,
gradient-of-the-SUM seed, concat order and duplicate-accumulating ScatterSub.  Arithmetic runs in
float64 / complex128 (DT_FLOAT tensors are widened) so that agreement with the fp64 oracle can be
checked to ~1e-12.
"""
from __future__ import annotations

import codecs
import re

import numpy as np

_DT = {"DT_FLOAT": np.float64, "DT_DOUBLE": np.float64, "DT_INT32": np.int32, "DT_INT64": np.int64,
       "DT_BOOL": np.bool_, "DT_COMPLEX64": np.complex128, "DT_STRING": object}
_RAW = {"DT_FLOAT": np.float32, "DT_DOUBLE": np.float64, "DT_INT32": np.int32, "DT_INT64": np.int64,
        "DT_BOOL": np.bool_, "DT_COMPLEX64": np.complex64}


# ------------------------------------------------------------------ text-format protobuf reader
def _scalar(tok: str):
    if tok.startswith('"'):
        return codecs.escape_decode(tok[1:-1].encode("latin1"))[0]        # bytes
    if tok in ("true", "false"):
        return tok == "true"
    try:
        return int(tok)
    except ValueError:
        pass
    try:
        return float(tok)
    except ValueError:
        return tok                                                          # enum identifier


def parse_pbtxt(path: str) -> dict:
    """-> nested dict: field name -> list of values (scalars or dicts), in file order."""
    root: dict = {}
    stack = [root]
    with open(path, "r", encoding="latin1") as f:
        for line in f:
            t = line.strip()
            if not t:
                continue
            if t == "}":
                stack.pop()
            elif t.endswith("{"):
                key = t[:-1].strip().rstrip(":").strip()
                child: dict = {}
                stack[-1].setdefault(key, []).append(child)
                stack.append(child)
            else:
                key, _, val = t.partition(":")
                stack[-1].setdefault(key.strip(), []).append(_scalar(val.strip()))
    assert len(stack) == 1, "unbalanced braces"
    return root


def _tensor(t: dict):
    dt = t["dtype"][0]
    dims = [d.get("size", [0])[0] for d in t.get("tensor_shape", [{}])[0].get("dim", [])]   # absent = proto default 0
    n = int(np.prod(dims)) if dims else 1
    if "tensor_content" in t:
        arr = np.frombuffer(t["tensor_content"][0], dtype=_RAW[dt]).astype(_DT[dt])
    elif dt == "DT_STRING":
        arr = np.array(t.get("string_val", []), dtype=object)
    else:
        key = {"DT_FLOAT": "float_val", "DT_DOUBLE": "double_val", "DT_INT32": "int_val", "DT_INT64": "int64_val",
               "DT_BOOL": "bool_val"}[dt]
        vals = t.get(key, [0] if n else [])
        arr = np.array(vals, dtype=_DT[dt])
        if n == 0:
            arr = arr[:0]
        elif arr.size == 1 and n != 1:
            arr = np.full(n, arr[0], dtype=_DT[dt])                         # splat encoding
    return arr.reshape(dims)


class Node:
    __slots__ = ("name", "op", "inputs", "ctrl", "attr")

    def __init__(self, d: dict):
        self.name = d["name"][0].decode()
        self.op = d["op"][0].decode()
        ins = [i.decode() for i in d.get("input", [])]
        self.inputs = [i for i in ins if not i.startswith("^")]
        self.ctrl = [i[1:] for i in ins if i.startswith("^")]
        self.attr = {a["key"][0].decode(): a["value"][0] for a in d.get("attr", [])}

    def a(self, key, default=None):
        v = self.attr.get(key)
        if v is None:
            return default
        for k in ("i", "f", "b", "s", "type"):
            if k in v:
                return v[k][0]
        if "tensor" in v:
            return _tensor(v["tensor"][0])
        if "list" in v:
            return v["list"][0]
        if "shape" in v:
            return v["shape"][0]
        return default


class Graph:
    def __init__(self, path: str):
        self.nodes = {}
        self.order = []
        for d in parse_pbtxt(path)["node"]:
            n = Node(d)
            self.nodes[n.name] = n
            self.order.append(n.name)

    def __getitem__(self, name: str) -> Node:
        return self.nodes[name]

    def consumers(self, name: str):
        return [n for n in self.nodes.values() if any(i.split(":")[0] == name for i in n.inputs)]

    # -------------------------------------------------------------- execution
    def run(self, fetches, feeds: dict):
        """Evaluate `fetches` (node names, optionally name:k) given `feeds` {node name: ndarray}."""
        memo = {k: (v if isinstance(v, tuple) else (np.asarray(v),)) for k, v in feeds.items()}
        return [self._value(f, memo) for f in fetches]

    def _value(self, ref: str, memo):
        name, _, idx = ref.partition(":")
        k = int(idx) if idx else 0
        if name not in memo:
            # iterative post-order to stay clear of Python's recursion limit on long chains
            stack = [name]
            while stack:
                cur = stack[-1]
                if cur in memo:
                    stack.pop()
                    continue
                node = self.nodes[cur]
                missing = [i.split(":")[0] for i in node.inputs if i.split(":")[0] not in memo]
                if missing:
                    stack.extend(missing)
                    continue
                args = []
                for i in node.inputs:
                    nm, _, ix = i.partition(":")
                    args.append(memo[nm][int(ix) if ix else 0])
                out = _OPS[node.op](node, *args)
                memo[cur] = out if isinstance(out, tuple) else (out,)
                stack.pop()
        return memo[name][k]


# ------------------------------------------------------------------ op definitions (TensorFlow 1.2 semantics)
_OPS = {}


def op(*names):
    def deco(fn):
        for n in names:
            _OPS[n] = fn
        return fn
    return deco


@op("Const")
def _const(n):
    return n.a("value")


@op("Identity", "StopGradient", "PreventGradient")
def _identity(n, x):
    return x


@op("Gather")
def _gather(n, params, indices):
    return params[np.asarray(indices)]


@op("Mul")
def _mul(n, x, y): return x * y
@op("Add")
def _add(n, x, y): return x + y
@op("Sub")
def _sub(n, x, y): return x - y
@op("RealDiv")
def _div(n, x, y): return x / y
@op("Neg")
def _neg(n, x): return -x
@op("Square")
def _square(n, x): return x * x
@op("Rsqrt")
def _rsqrt(n, x):
    with np.errstate(divide="ignore"):
        return 1.0 / np.sqrt(x)
@op("RsqrtGrad")
def _rsqrt_grad(n, y, dy): return dy * -0.5 * y * y * y            # math_grad: dy * -0.5 * y^3
@op("Tanh")
def _tanh(n, x): return np.tanh(x)
@op("TanhGrad")
def _tanh_grad(n, y, dy): return dy * (1.0 - y * y)
@op("Sigmoid")
def _sigmoid(n, x): return 1.0 / (1.0 + np.exp(-x))
@op("SigmoidGrad")
def _sigmoid_grad(n, y, dy): return dy * y * (1.0 - y)
@op("Minimum")
def _minimum(n, x, y): return np.minimum(x, y)
@op("Maximum")
def _maximum(n, x, y): return np.maximum(x, y)
@op("GreaterEqual")
def _ge(n, x, y): return x >= y
@op("LessEqual")
def _le(n, x, y): return x <= y
@op("Greater")
def _gt(n, x, y): return x > y
@op("Less")
def _lt(n, x, y): return x < y
@op("LogicalNot")
def _not(n, x): return np.logical_not(x)
@op("Select")
def _select(n, c, x, y): return np.where(c, x, y)
@op("ZerosLike")
def _zeros_like(n, x): return np.zeros_like(x)
@op("FloorDiv")
def _floordiv(n, x, y): return np.floor_divide(x, y)
@op("FloorMod")
def _floormod(n, x, y): return np.mod(x, y)


@op("Complex")
def _complex(n, re, im): return re + 1j * im
@op("Real")
def _real(n, x): return np.real(x)
@op("Imag")
def _imag(n, x): return np.imag(x)
@op("Conj")
def _conj(n, x): return np.conj(x)
@op("FFT")
def _fft(n, x): return np.fft.fft(x, axis=-1)
@op("IFFT")
def _ifft(n, x): return np.fft.ifft(x, axis=-1)


@op("Cast")
def _cast(n, x):
    dst = n.a("DstT").decode() if isinstance(n.a("DstT"), bytes) else n.a("DstT")
    return np.asarray(x).astype(_DT[dst])


@op("Shape")
def _shape(n, x): return np.array(np.shape(x), dtype=np.int32)
@op("Size")
def _size(n, x): return np.array(np.size(x), dtype=np.int32)
@op("Rank")
def _rank(n, x): return np.array(np.ndim(x), dtype=np.int32)


@op("Fill")
def _fill(n, dims, value): return np.full(tuple(int(v) for v in np.atleast_1d(dims)), value)


@op("Reshape")
def _reshape(n, x, shape): return np.reshape(x, tuple(int(v) for v in np.atleast_1d(shape)))


@op("ExpandDims")
def _expand(n, x, axis): return np.expand_dims(x, int(axis))


@op("Squeeze")
def _squeeze(n, x):
    dims = n.a("squeeze_dims")
    ax = tuple(int(v) for v in dims.get("i", [])) if isinstance(dims, dict) else None
    return np.squeeze(x, axis=ax if ax else None)


@op("Slice")
def _slice(n, x, begin, size):
    sl = tuple(slice(int(b), None if int(s) == -1 else int(b) + int(s)) for b, s in zip(begin, size))
    return x[sl]


@op("StridedSlice")
def _strided_slice(n, x, begin, end, strides):
    bm, em, sm = int(n.a("begin_mask", 0)), int(n.a("end_mask", 0)), int(n.a("shrink_axis_mask", 0))
    assert int(n.a("ellipsis_mask", 0)) == 0 and int(n.a("new_axis_mask", 0)) == 0
    idx = []
    for i, (b, e, s) in enumerate(zip(begin, end, strides)):
        if (sm >> i) & 1:
            idx.append(int(b))
        else:
            idx.append(slice(None if (bm >> i) & 1 else int(b), None if (em >> i) & 1 else int(e), int(s)))
    return np.asarray(x)[tuple(idx)]


@op("Pack")
def _pack(n, *xs): return np.stack(xs, axis=int(n.a("axis", 0)))


@op("Unpack")
def _unpack(n, x): return tuple(np.moveaxis(x, int(n.a("axis", 0)), 0))


@op("ConcatV2")
def _concat(n, *args): return np.concatenate(args[:-1], axis=int(args[-1]))


@op("Tile")
def _tile(n, x, mult): return np.tile(x, tuple(int(v) for v in mult))


@op("Range")
def _range(n, start, limit, delta): return np.arange(int(start), int(limit), int(delta), dtype=np.int32)


@op("Pad")
def _pad(n, x, paddings): return np.pad(x, [(int(a), int(b)) for a, b in paddings])


def _reduce(fn):
    def run(n, x, axes):
        ax = tuple(int(v) % max(np.ndim(x), 1) for v in np.atleast_1d(axes)) if np.size(axes) else ()
        if np.ndim(x) == 0 or (np.size(axes) == 0):
            return x if np.size(axes) == 0 else fn(x)
        return fn(x, axis=ax, keepdims=bool(n.a("keep_dims", False)))
    return run


_OPS["Sum"] = _reduce(np.sum)
_OPS["Prod"] = _reduce(np.prod)
_OPS["Mean"] = _reduce(np.mean)
_OPS["Max"] = _reduce(np.max)
_OPS["Min"] = _reduce(np.min)


@op("AddN")
def _addn(n, *xs):
    out = xs[0]
    for x in xs[1:]:
        out = out + x
    return out


@op("BroadcastGradientArgs")
def _bga(n, s0, s1):
    """Reduction axes that undo NumPy-style broadcasting of shapes s0, s1 (ops/array_ops BCast)."""
    s0, s1 = [int(v) for v in s0], [int(v) for v in s1]
    r = max(len(s0), len(s1))
    p0, p1 = [1] * (r - len(s0)) + s0, [1] * (r - len(s1)) + s1
    r0 = [i for i in range(r) if p0[i] == 1 and (p1[i] != 1 or i < r - len(s0))]
    r1 = [i for i in range(r) if p1[i] == 1 and (p0[i] != 1 or i < r - len(s1))]
    # TF also lists axes where BOTH are 1 for both sides; summing a size-1 axis is a no-op either way
    both = [i for i in range(r) if p0[i] == 1 and p1[i] == 1]
    return (np.array(sorted(set(r0 + both)), dtype=np.int32), np.array(sorted(set(r1 + both)), dtype=np.int32))


@op("DynamicStitch")
def _dynamic_stitch(n, *args):
    k = len(args) // 2
    idx, data = args[:k], args[k:]
    size = max(int(np.max(i)) for i in idx if np.size(i)) + 1
    first = next(d for i, d in zip(idx, data) if np.size(i))
    tail = np.shape(first)[np.ndim(next(i for i in idx if np.size(i))):]
    out = np.zeros((size,) + tuple(tail), dtype=np.asarray(first).dtype)
    for i, d in zip(idx, data):          # later inputs win, as documented
        i = np.asarray(i)
        out[i.reshape(-1)] = np.asarray(d).reshape((-1,) + tuple(tail))
    return out


@op("InvertPermutation")
def _invperm(n, x):
    out = np.empty_like(x)
    out[x] = np.arange(len(x), dtype=x.dtype)
    return out


@op("Transpose")
def _transpose(n, x, perm): return np.transpose(x, tuple(int(v) for v in perm))


@op("ScatterSub")
def _scatter_sub(n, ref, indices, updates):
    """ref[indices[i], ...] -= updates[i, ...]; duplicate indices all contribute (state_ops docs)."""
    out = np.array(ref, copy=True)
    ind = np.asarray(indices).reshape(-1)
    np.subtract.at(out, ind, np.asarray(updates).reshape((len(ind),) + out.shape[1:]))
    return out


@op("NoOp")
def _noop(n, *a): return np.array(0)
