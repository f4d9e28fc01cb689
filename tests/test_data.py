"""File-format contract (holE.py:44-94, 381-424) and the FB15k id-file pin."""
import gzip
import os

import numpy as np
import pytest

from graphembeddings_amd import data as D


@pytest.fixture(scope="module")
def fb():
    return D.fb15k_shape()


def test_fb15k_counts(fb):
    # diffbot_data/FB15k/README:44-45: 14,951 mids, 1,345 relation types, 50,000 validation triplets
    assert fb.relation_count == 1345
    assert fb.entity_count == 16296
    assert fb.entity_count - fb.relation_count == 14951
    assert len(fb.type_to_ids) == 815
    assert len(fb.type_to_ids["RELATION"]) == 1345 and fb.type_to_ids["RELATION"][:3] == [0, 1, 2]
    assert len(fb.type_to_ids["human"]) == 4206
    assert fb.validation_triples.shape == (50000, 3)
    assert fb.validation_triples[0].tolist() == [2625, 15605, 993]
    assert fb.triples is None and fb.triple_count == 0


def test_fb15k_valid_equals_raw_freebase_mapped_through_id_files(fb, golden_dir):
    """Reference-held fixture: triples-valid.txt is freebase_mtr100_mte100-valid.txt
    (head, relation, tail as Freebase ids) mapped through entity_metadata.tsv / relation_ids.txt,
    written in (head, tail, relation) order (holE.py:76-81) -- line for line."""
    mid_to_idx, rel_to_idx = {}, {}
    with D._open_text(os.path.join(D.PACKAGE_FB15K_DIR, "entity_metadata.tsv.gz")) as f:
        next(f)
        for line in f:
            idx, fid, _name, ty = line.rstrip("\n").split("\t")
            (rel_to_idx if ty == "RELATION" else mid_to_idx)[fid] = int(idx)
    with D._open_text(os.path.join(D.PACKAGE_FB15K_DIR, "relation_ids.txt.gz")) as f:
        for line in f:
            name, rid = line.rstrip("\n").split("\t")
            assert rel_to_idx[name] == int(rid)
    with gzip.open(os.path.join(golden_dir, "fb15k_freebase_valid_head2000.txt.gz"), "rt") as f:
        raw = [l.rstrip("\n").split("\t") for l in f]
    mapped = np.array([[mid_to_idx[h], mid_to_idx[t], rel_to_idx[r]] for h, r, t in raw], dtype=np.int32)
    assert np.array_equal(mapped, fb.validation_triples[:2000])


def test_type_arrays_roundtrip(fb):
    names, id_to_type, offsets, ids = fb.type_arrays()
    assert len(names) == 815 and id_to_type.shape == (16296,) and offsets[-1] == 16296
    assert (id_to_type >= 0).all()
    for code in (0, 5, 100, 814):
        members = ids[offsets[code]:offsets[code + 1]]
        assert (id_to_type[members] == code).all()
        assert members.tolist() == fb.type_to_ids[names[code]]
    sizes = np.diff(offsets)
    assert (sizes == 1).sum() == 372          # singleton types


def _write_dir(tmp_path, six_columns):
    hdr = "Index\tId\tName\tType" + ("\tMentions\tIsTail" if six_columns else "")
    rows = [(0, "r0", "r0", "RELATION"), (1, "r1", "r1", "RELATION"), (2, "P1", "alice", "P"),
            (3, "P2", "bob", "P"), (4, "S1", "java", "S")]
    with open(tmp_path / "entity_metadata.tsv", "w") as f:
        f.write(hdr + "\n")
        for i, (idx, a, b, c) in enumerate(rows):
            f.write(f"{idx}\t{a}\t{b}\t{c}" + (f"\t{100 * i}\ttrue" if six_columns else "") + "\n")
    (tmp_path / "relation_ids.txt").write_text("r0\t0\nr1\t1\n")
    (tmp_path / "triples.txt").write_text("2\t4\t0\n3\t4\t1\n2\t3\t1\n")
    (tmp_path / "triples-valid.txt").write_text("3\t4\t0\n")
    (tmp_path / "test_positive_triples.txt").write_text("2\t4\t1\n")
    return str(tmp_path)


@pytest.mark.parametrize("six", [False, True])
def test_init_data_accepts_4_and_6_columns(tmp_path, six):
    d = D.init_data(_write_dir(tmp_path, six))
    assert (d.entity_count, d.relation_count, d.triple_count) == (5, 2, 3)
    assert d.triples.dtype == np.int32 and d.triples.tolist() == [[2, 4, 0], [3, 4, 1], [2, 3, 1]]
    assert d.type_to_ids["P"] == [2, 3] and d.id_to_type[4] == "S"


def test_init_inference_data(tmp_path):
    d = D.init_inference_data(_write_dir(tmp_path, True), min_mentions=250)
    assert d.test_triples[2][1] == {4}
    # true_triples keeps only train/valid triples sharing (head, relation) with a test triple (holE.py:421)
    assert d.true_triples[2][1] == {3} and 0 not in d.true_triples[2]
    # min_mentions filter keeps ids with mentions >= 250 or whose id starts with 'P' (holE.py:397)
    assert d.type_to_ids["P"] == [2, 3] and d.type_to_ids["S"] == [4] and d.type_to_ids["RELATION"] == []


def test_out_of_range_ids_rejected(tmp_path):
    p = _write_dir(tmp_path, False)
    (tmp_path / "triples.txt").write_text("2\t9\t0\n")
    with pytest.raises(ValueError):
        D.init_data(p)


def test_batcher_and_synthetic(fb):
    tr = D.synthetic_fb15k_triples(fb, n_triples=5000, seed=0)
    assert tr.shape == (5000, 3) and tr[:, 2].max() < 1345 and tr[:, :2].min() >= 1345 and tr.max() < 16296
    b = D.TripleBatcher(tr, 512, seed=1)
    seen = [b.next() for _ in range(12)]
    assert all(x.shape == (512, 3) for x in seen)
    data, tri = D.synthetic_large(n_entities=5000, n_triples=20000, seed=1)
    names, id_to_type, offsets, ids = D.synthetic_large_type_arrays(data)
    assert data.entity_count == 5018 and tri.shape == (20000, 3) and len(names) == 13
    assert (id_to_type[ids[offsets[1]:offsets[2]]] == 1).all()
    assert (np.diff(offsets)[1] / 5000) > 0.98


def test_binary_triple_cache(tmp_path):
    p = _write_dir(tmp_path, False)
    a = D.init_data(p, cache=True).triples
    assert (tmp_path / "triples.txt.npy").exists()
    b = D.init_data(p, cache=True).triples          # second load comes from the memory-mapped cache
    assert np.array_equal(a, b) and b.dtype == np.int32
    (tmp_path / "triples.txt").write_text("2\t4\t0\n")   # TSV newer than the cache -> re-parsed
    import os, time
    os.utime(tmp_path / "triples.txt", (time.time() + 5, time.time() + 5))
    assert D.init_data(p, cache=True).triples.tolist() == [[2, 4, 0]]


def test_read_triples_cached_side_car_stale_cache_and_read_only_dir(tmp_path):
    """data.read_triples_cached: the binary side-car is written on first use, served (memory-mapped)
    afterwards, REBUILT when the TSV is newer or the side-car is not an int32 [T,3] array, and a directory that cannot be
    written to costs nothing but the cache."""
    import os
    import time
    from graphembeddings_amd import data as D
    p = str(tmp_path / "triples.txt")
    a = np.array([[5, 6, 0], [7, 8, 1], [9, 10, 0]], dtype=np.int64)
    D.write_triples(p, a)
    got = D.read_triples_cached(p)
    assert np.array_equal(got, a) and os.path.exists(p + ".npy")
    again = D.read_triples_cached(p)
    assert isinstance(again, np.memmap) and again.dtype == np.int32 and np.array_equal(again, a)
    # stale: the TSV changes after the side-car was written
    b = np.array([[1, 2, 0], [3, 4, 1]], dtype=np.int64)
    D.write_triples(p, b)
    os.utime(p, (time.time() + 5, time.time() + 5))
    assert np.array_equal(D.read_triples_cached(p), b)
    os.utime(p + ".npy", (time.time() + 10, time.time() + 10))
    assert np.array_equal(D.read_triples_cached(p), b)          # and the rebuilt side-car is the new content
    # a side-car of the wrong shape / dtype is not trusted
    np.save(p + ".npy", np.zeros((4, 2), dtype=np.float32))
    os.utime(p + ".npy", (time.time() + 20, time.time() + 20))
    assert np.array_equal(D.read_triples_cached(p), b)
    # read-only directory (skipped when running as root, who may write anywhere)
    ro = tmp_path / "ro"
    ro.mkdir()
    q = str(ro / "triples.txt")
    D.write_triples(q, a)
    os.chmod(ro, 0o555)
    try:
        assert np.array_equal(D.read_triples_cached(q), a)
        if os.geteuid() != 0:
            assert not os.path.exists(q + ".npy")
    finally:
        os.chmod(ro, 0o755)
