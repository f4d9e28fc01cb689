"""Triple / metadata ingest with the reference's on-disk contract (holE.py:44-94, 381-424).

Files in --data_dir (any of them may also be gzip-compressed with a `.gz` suffix):
  entity_metadata.tsv        header line, then `index \\t id \\t name \\t type [\\t mentions \\t is_tail]`
                             (holE.py:59 unpacks 6 columns; every file the reference ships has 4 --
                             both are accepted).  Relation rows come first (type RELATION).
  relation_ids.txt           one line per relation (holE.py:52)
  triples.txt                `head \\t tail \\t relation` decimal ints (holE.py:76-81)
  triples-valid.txt          same format (holE.py:83-92)
  test_positive_triples.txt  same format (holE.py:405-411)
Row ids index the single shared table; relation ids are 0..R-1 and entity ids start at R.

Also provides the seeded synthetic workloads of SURVEY.md section 8(d) (FB15k-shaped and the
1.2 M-entity Diffbot-like set) for benchmarks: there is no network, and the FB15k train split is not
shipped with the reference (.MISSING_LARGE_BLOBS:1-3).
"""
from __future__ import annotations

import gzip
import io
import os
from collections import defaultdict
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

PACKAGE_FB15K_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "fb15k")


def _resolve(path: str) -> Optional[str]:
    if os.path.exists(path):
        return path
    if os.path.exists(path + ".gz"):
        return path + ".gz"
    return None


def _open_text(path: str):
    if path.endswith(".gz"):
        return io.TextIOWrapper(gzip.open(path, "rb"), encoding="utf-8", newline="")
    return open(path, "r", encoding="utf-8", newline="")


def count_lines(path: str) -> int:
    """`sum(1 for line in open(file))` of holE.py:52-53."""
    with _open_text(path) as f:
        return sum(1 for _ in f)


def read_triples(path: str) -> np.ndarray:
    """TSV `head\\ttail\\trelation` -> int32 [T,3] (the decode_csv of holE.py:76-81)."""
    import pandas as pd
    df = pd.read_csv(path, sep="\t", header=None, dtype=np.int32, names=["h", "t", "r"],
                     compression="gzip" if path.endswith(".gz") else None)
    return np.ascontiguousarray(df.to_numpy(dtype=np.int32))


def read_triples_cached(path: str) -> np.ndarray:
    """read_triples with a binary side-car: `<file>.npy` (int32 [T,3]) is written next to the TSV on
    first use and memory-mapped afterwards when it is newer than the TSV.  Parsing 30 M-250 M-triple
    TSVs (README.md:60,120) is otherwise the start-up bottleneck once the kernels are fast."""
    cache = path + ".npy"
    try:
        if os.path.exists(cache) and os.path.getmtime(cache) >= os.path.getmtime(path):
            arr = np.load(cache, mmap_mode="r")
            if arr.ndim == 2 and arr.shape[1] == 3 and arr.dtype == np.int32:
                return arr
    except Exception:
        pass
    arr = read_triples(path)
    try:
        tmp = cache + f".tmp{os.getpid()}.npy"          # (several ranks may build the side-car at once: one file each, last rename wins)
        np.save(tmp, arr)
        os.replace(tmp, cache)
    except OSError:
        pass  # read-only data dir: keep going without the cache
    return arr


def write_triples(path: str, triples: np.ndarray) -> None:
    np.savetxt(path, np.asarray(triples, dtype=np.int64), fmt="%d", delimiter="\t")


@dataclass
class HolEData:
    """Pre-processing data used during training and inference (class HolEData, holE.py:25-34),
    plus the flat arrays the GPU sampler consumes."""
    type_to_ids: Dict[str, List[int]] = field(default_factory=lambda: defaultdict(list))
    id_to_type: Dict[int, str] = field(default_factory=dict)
    entity_count: int = 0      # table rows: relations + entities (holE.py:58)
    relation_count: int = 0
    triple_count: int = 0
    triples: Optional[np.ndarray] = None
    validation_triples: Optional[np.ndarray] = None
    id_to_metadata: Dict[int, str] = field(default_factory=dict)

    def type_arrays(self):
        """(type_names, id_to_type int32 [entity_count], type_offsets int64 [T+1], type_ids int32):
        dense coding of the two dicts, types numbered in first-appearance order."""
        names = list(self.type_to_ids.keys())
        code = {n: i for i, n in enumerate(names)}
        id_to_type = np.full(self.entity_count, -1, dtype=np.int32)
        for idx, ty in self.id_to_type.items():
            if 0 <= idx < self.entity_count:
                id_to_type[idx] = code[ty]
        offsets = np.zeros(len(names) + 1, dtype=np.int64)
        for i, n in enumerate(names):
            offsets[i + 1] = offsets[i] + len(self.type_to_ids[n])
        ids = np.empty(int(offsets[-1]), dtype=np.int32)
        for i, n in enumerate(names):
            ids[offsets[i]:offsets[i + 1]] = self.type_to_ids[n]
        return names, id_to_type, offsets, ids


def _read_metadata(entity_file: str, data: HolEData, min_mentions: Optional[int] = None):
    with _open_text(entity_file) as f:
        next(f)  # skip header (holE.py:56)
        for line in f:
            cols = line.rstrip("\r\n").split("\t")
            if len(cols) == 6:
                index, ent_id, name, ent_type, mentions, _is_tail = cols
            elif len(cols) == 4:
                index, ent_id, name, ent_type = cols
                mentions = None
            else:
                raise ValueError(f"{entity_file}: expected 4 or 6 tab-separated columns, got {len(cols)}")
            data.entity_count += 1
            index = int(index)
            keep = True
            if min_mentions is not None:  # inference-time candidate filter (holE.py:397)
                keep = (mentions is not None and int(mentions) >= min_mentions) or ent_id.startswith("P")
            if keep:
                data.type_to_ids[ent_type].append(index)
            data.id_to_type[index] = ent_type
            data.id_to_metadata[index] = ent_id + " " + name


def init_data(data_dir: str, require_train: bool = True, cache: bool = False) -> HolEData:
    """Model pre-processing (init_data, holE.py:44-94) without the TF input queues: the triple files
    are parsed once into int32 arrays; batching is done by TripleBatcher."""
    entity_file = _resolve(os.path.join(data_dir, "entity_metadata.tsv"))
    relation_file = _resolve(os.path.join(data_dir, "relation_ids.txt"))
    train_file = _resolve(os.path.join(data_dir, "triples.txt"))
    valid_file = _resolve(os.path.join(data_dir, "triples-valid.txt"))
    for name, p in (("entity_metadata.tsv", entity_file), ("relation_ids.txt", relation_file)):
        if p is None:
            raise FileNotFoundError(os.path.join(data_dir, name))
    if train_file is None and require_train:
        raise FileNotFoundError(os.path.join(data_dir, "triples.txt"))
    data = HolEData()
    data.relation_count = count_lines(relation_file)
    _read_metadata(entity_file, data)
    if train_file is not None:
        data.triples = (read_triples_cached if cache else read_triples)(train_file)
        data.triple_count = int(data.triples.shape[0])
    if valid_file is not None:
        data.validation_triples = read_triples(valid_file)
    for name, arr in (("triples.txt", data.triples), ("triples-valid.txt", data.validation_triples)):
        if arr is not None and arr.size and (arr.min() < 0 or arr.max() >= data.entity_count):
            raise ValueError(f"{name}: id outside [0, {data.entity_count})")
    return data


@dataclass
class HolEInferenceData(HolEData):
    """class HolEInferenceData (holE.py:373-378)."""
    true_triples: dict = field(default_factory=lambda: defaultdict(lambda: defaultdict(set)))
    test_triples: dict = field(default_factory=lambda: defaultdict(lambda: defaultdict(set)))
    test_array: Optional[np.ndarray] = None


def init_inference_data(data_dir: str, min_mentions: Optional[int] = None) -> HolEInferenceData:
    """init_inference_data (holE.py:381-424): metadata, test triples as {head:{rel:{tails}}}, and the
    train/valid triples that share a (head, relation) with a test triple as true_triples."""
    data = HolEInferenceData()
    entity_file = _resolve(os.path.join(data_dir, "entity_metadata.tsv"))
    relation_file = _resolve(os.path.join(data_dir, "relation_ids.txt"))
    test_file = _resolve(os.path.join(data_dir, "test_positive_triples.txt"))
    if entity_file is None or relation_file is None or test_file is None:
        raise FileNotFoundError(f"{data_dir}: entity_metadata.tsv, relation_ids.txt and "
                                "test_positive_triples.txt are required")
    _read_metadata(entity_file, data, min_mentions=min_mentions)
    data.relation_count = count_lines(relation_file)
    data.test_array = read_triples(test_file)
    for h, t, r in data.test_array:
        data.test_triples[int(h)][int(r)].add(int(t))
    for fname in ("triples.txt", "triples-valid.txt"):
        p = _resolve(os.path.join(data_dir, fname))
        if p is None:
            continue
        arr = read_triples(p)
        if fname == "triples.txt":
            data.triples, data.triple_count = arr, int(arr.shape[0])
        else:
            data.validation_triples = arr
        for h, t, r in arr:
            if int(r) in data.test_triples[int(h)]:
                data.true_triples[int(h)][int(r)].add(int(t))
    return data


class TripleBatcher:
    """Batch source standing in for tf.train.shuffle_batch (holE.py:281-283): uniformly shuffled
    batches of exactly batch_size (allow_smaller_final_batch=False), reshuffled every pass."""

    def __init__(self, triples: np.ndarray, batch_size: int, seed: int = 0):
        self.triples = np.ascontiguousarray(triples, dtype=np.int32)
        self.batch_size = int(batch_size)
        self.rng = np.random.default_rng(seed)
        self._perm = None
        self._pos = 0

    def next(self) -> np.ndarray:
        n = len(self.triples)
        if self._perm is None or self._pos + self.batch_size > n:
            self._perm = self.rng.permutation(n)
            self._pos = 0
        sel = self._perm[self._pos:self._pos + self.batch_size]
        self._pos += self.batch_size
        return self.triples[sel]


# ----------------------------------------------------------------------------- synthetic workloads

def _zipf_sample(rng, n_items: int, size: int, s: float) -> np.ndarray:
    """Indices in [0, n_items) with P(i) ~ 1/(i+1)^s over a random permutation of the items."""
    w = 1.0 / np.power(np.arange(1, n_items + 1, dtype=np.float64), s)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    ranks = np.searchsorted(cdf, rng.random(size), side="right")
    perm = rng.permutation(n_items)
    return perm[np.minimum(ranks, n_items - 1)]


def fb15k_shape(data_dir: str = PACKAGE_FB15K_DIR) -> HolEData:
    """The real FB15k id space (16,296 rows = 1,345 relation rows + 14,951 entities, 815 types)
    and its valid split, from the files shipped with the package."""
    return init_data(data_dir, require_train=False)


def synthetic_fb15k_triples(data: HolEData, n_triples: int = 483142, seed: int = 0, zipf_s: float = 1.0) -> np.ndarray:
    """FB15k-shaped train triples when the real triples.txt is absent: relation ~
    empirical frequency of the valid split, head/tail Zipf(s=1.0) over the entity rows."""
    rng = np.random.default_rng(seed)
    R, N = data.relation_count, data.entity_count
    if data.validation_triples is not None and len(data.validation_triples):
        freq = np.bincount(data.validation_triples[:, 2], minlength=R).astype(np.float64) + 0.05
    else:
        freq = np.ones(R)
    rel = rng.choice(R, size=n_triples, p=freq / freq.sum())
    head = R + _zipf_sample(rng, N - R, n_triples, zipf_s)
    tail = R + _zipf_sample(rng, N - R, n_triples, zipf_s)
    return np.stack([head, tail, rel], axis=1).astype(np.int32)


def synthetic_large(n_entities: int = 1_200_000, n_relations: int = 18, n_types: int = 12,
                    n_triples: int = 30_000_000, seed: int = 1234, zipf_s: float = 0.8):
    """Diffbot-like synthetic set of SURVEY.md 8(d) / BASELINE config 4: `n_relations` relation rows
    first, then entities; 12 types with >99 % of the entities in one (README.md:120); head/tail
    Zipf(0.8), relation uniform.  Returns (HolEData with type tables filled, triples int32 [T,3])."""
    rng = np.random.default_rng(seed)
    data = HolEData()
    data.relation_count = n_relations
    data.entity_count = n_relations + n_entities
    small = max(1, n_entities // 100 // max(1, n_types - 1))
    sizes = [n_entities - small * (n_types - 1)] + [small] * (n_types - 1)
    type_of = np.repeat(np.arange(n_types), sizes)
    rng.shuffle(type_of)
    data.type_to_ids["RELATION"] = list(range(n_relations))
    for i in range(n_relations):
        data.id_to_type[i] = "RELATION"
    ent_ids = np.arange(n_relations, n_relations + n_entities)
    for ty in range(n_types):
        data.type_to_ids[f"T{ty}"] = ent_ids[type_of == ty]
    data._id_to_type_code = np.concatenate([np.zeros(n_relations, np.int32), 1 + type_of.astype(np.int32)])
    head = n_relations + _zipf_sample(rng, n_entities, n_triples, zipf_s)
    tail = n_relations + _zipf_sample(rng, n_entities, n_triples, zipf_s)
    rel = rng.integers(0, n_relations, size=n_triples)
    triples = np.stack([head, tail, rel], axis=1).astype(np.int32)
    data.triples, data.triple_count = triples, n_triples
    return data, triples


def synthetic_large_type_arrays(data: HolEData):
    """type arrays for synthetic_large without materialising 1.2 M dict entries."""
    names = list(data.type_to_ids.keys())
    offsets = np.zeros(len(names) + 1, dtype=np.int64)
    for i, n in enumerate(names):
        offsets[i + 1] = offsets[i] + len(data.type_to_ids[n])
    ids = np.concatenate([np.asarray(data.type_to_ids[n], dtype=np.int32) for n in names])
    return names, data._id_to_type_code.astype(np.int32), offsets, ids
