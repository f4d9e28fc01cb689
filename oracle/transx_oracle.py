"""CPU oracle for the reference's native Bernoulli filtered sampler (init.cpp) -- TEST INFRASTRUCTURE.

Two things live here:

1. `InitCppRestatement`: a line-by-line restatement of /root/reference/init.cpp (`init()` 47-127,
   `randd`/`rand_max` 145-157, `corrupt_head` 159-189, `corrupt_tail` 191-221, `getBatch` 223-246),
   INCLUDING its global LCG stream and its two latent defects (the `memset(.., sizeof(pointer))` of
   init.cpp:94-95 and the `j < rigHead[i]` loop bound of init.cpp:112,117), so that its output can be
   compared bit for bit with the reference binary itself.  PINNED: tests/test_transx_oracle.py runs
   oracle/_ref/init.so (init.cpp compiled from where it lies in /root/reference, make.sh:1 flags) on
   generated ./data files and checks every batch equal.

2. `bernoulli_corrupt_batch`: the same corruption rule (filtered replacement via the
   skip-the-true-entities mapping, head/tail side chosen with probability hpt/(hpt+tph)) restated
   for a parallel sampler: per-row Philox4x32-10 draws instead of one sequential LCG, defects fixed,
   entity ids allowed to live in [ent_lo, ent_lo + n_ent) of the shared holE.py table.  This is the
   oracle of ge_bernoulli_corrupt_batch (bit-exact).  Its mapping function is the pinned one.
"""
from __future__ import annotations

import numpy as np

from .hole_oracle import philox4x32_10

MASK64 = (1 << 64) - 1


def _sorted_by(tri, keys):
    order = np.lexsort(tuple(tri[:, k] for k in reversed(keys)))
    return tri[order]


class InitCppRestatement:
    """State and behaviour of init.so.  Triples are (h, t, r) rows as read from triple2id.txt
    (init.cpp:71-73 reads h, then t, then r)."""

    def __init__(self, triples, entity_total, relation_total, reproduce_defects=True):
        tri = np.asarray(triples, dtype=np.int64)
        self.trainList = tri
        self.entityTotal, self.relationTotal, self.tripleTotal = int(entity_total), int(relation_total), len(tri)
        H, T, R = 0, 1, 2
        self.trainHead = _sorted_by(tri, (H, R, T))      # cmp_head: (h, r, t)   init.cpp:20-24
        self.trainTail = _sorted_by(tri, (T, R, H))      # cmp_tail: (t, r, h)   init.cpp:26-30
        E, n = self.entityTotal, self.tripleTotal
        lefHead, rigHead = np.zeros(E, np.int64), np.zeros(E, np.int64)   # calloc
        lefTail, rigTail = np.zeros(E, np.int64), np.zeros(E, np.int64)
        if reproduce_defects:
            rigHead[:2] = -1    # memset(rigHead, -1, sizeof(rigHead)): 8 bytes = two ints  (init.cpp:94)
            rigTail[:2] = -1
        else:
            rigHead[:] = -1
            rigTail[:] = -1
        th, tt = self.trainHead, self.trainTail
        for i in range(1, n):                              # init.cpp:96-105
            if tt[i, T] != tt[i - 1, T]:
                rigTail[tt[i - 1, T]] = i - 1
                lefTail[tt[i, T]] = i
            if th[i, H] != th[i - 1, H]:
                rigHead[th[i - 1, H]] = i - 1
                lefHead[th[i, H]] = i
        rigHead[th[n - 1, H]] = n - 1
        rigTail[tt[n - 1, T]] = n - 1
        self.lefHead, self.rigHead, self.lefTail, self.rigTail = lefHead, rigHead, lefTail, rigTail
        freqRel = np.bincount(tri[:, R], minlength=self.relationTotal).astype(np.float32)
        left_mean = np.zeros(self.relationTotal, np.float32)
        right_mean = np.zeros(self.relationTotal, np.float32)
        hi = (lambda r: r) if reproduce_defects else (lambda r: r + 1)   # `j < rigHead[i]` (init.cpp:112)
        for i in range(E):                                 # init.cpp:111-122
            for j in range(lefHead[i] + 1, hi(rigHead[i])):
                if th[j, R] != th[j - 1, R]:
                    left_mean[th[j, R]] += np.float32(1.0)
            if lefHead[i] <= rigHead[i]:
                left_mean[th[lefHead[i], R]] += np.float32(1.0)
            for j in range(lefTail[i] + 1, hi(rigTail[i])):
                if tt[j, R] != tt[j - 1, R]:
                    right_mean[tt[j, R]] += np.float32(1.0)
            if lefTail[i] <= rigTail[i]:
                right_mean[tt[lefTail[i], R]] += np.float32(1.0)
        with np.errstate(divide="ignore", invalid="ignore"):
            self.left_mean = (freqRel / left_mean).astype(np.float32)      # tails per head  (init.cpp:123-126)
            self.right_mean = (freqRel / right_mean).astype(np.float32)    # heads per tail
        self.next_random = 3                                # init.cpp:146

    # -- RNG (init.cpp:148-157) ------------------------------------------------------------------
    def randd(self):
        self.next_random = (self.next_random * 25214903917 + 11) & MASK64
        return self.next_random

    def rand_max(self, x):
        return int(self.randd() % x)

    # -- corrupt_head(h, r): a TAIL that is not a known tail of (h, r)  (init.cpp:159-189) --------
    def corrupt_head(self, h, r, tmp_fn=None):
        th = self.trainHead
        ll, rr = _range_of(th[:, 2], self.lefHead[h], self.rigHead[h], r)
        tmp = (tmp_fn or self.rand_max)(self.entityTotal - (rr - ll + 1))
        return _skip_known(th[:, 1], ll, rr, tmp)

    # -- corrupt_tail(t, r): a HEAD that is not a known head of (t, r)  (init.cpp:191-221) --------
    def corrupt_tail(self, t, r, tmp_fn=None):
        tt = self.trainTail
        ll, rr = _range_of(tt[:, 2], self.lefTail[t], self.rigTail[t], r)
        tmp = (tmp_fn or self.rand_max)(self.entityTotal - (rr - ll + 1))
        return _skip_known(tt[:, 0], ll, rr, tmp)

    def getBatch(self, batchSize):
        """init.cpp:223-246.  Returns ph, pt, pr, nh, nt, nr int32 arrays."""
        out = np.zeros((6, batchSize), dtype=np.int32)
        for b in range(batchSize):
            i = self.rand_max(self.tripleTotal)
            h, t, r = (int(v) for v in self.trainList[i])
            # float prob = 1000 * right_mean[r] / (right_mean[r] + left_mean[r]);   (fp32 arithmetic)
            prob = np.float32(1000) * self.right_mean[r] / (self.right_mean[r] + self.left_mean[r])
            # `randd(id) % 1000 < prob`: unsigned long long converted to float for the comparison
            if np.float32(self.randd() % 1000) < prob:
                j = self.corrupt_head(h, r)
                out[:, b] = (h, t, r, h, j, r)
            else:
                j = self.corrupt_tail(t, r)
                out[:, b] = (h, t, r, j, t, r)
        return tuple(out)


def _range_of(rel_col, lef0, rig0, r):
    """The two binary searches of init.cpp:161-176: [ll, rr] = positions of relation r inside
    [lef0, rig0] of the sorted array."""
    lef, rig = lef0 - 1, rig0
    while lef + 1 < rig:
        mid = (lef + rig) >> 1
        if rel_col[mid] >= r:
            rig = mid
        else:
            lef = mid
    ll = rig
    lef, rig = lef0, rig0 + 1
    while lef + 1 < rig:
        mid = (lef + rig) >> 1
        if rel_col[mid] <= r:
            lef = mid
        else:
            rig = mid
    rr = lef
    return int(ll), int(rr)


def _skip_known(ent_col, ll, rr, tmp):
    """init.cpp:177-188: map tmp in [0, E - cnt) onto the entities that are NOT among the sorted
    known entities ent_col[ll..rr]."""
    if tmp < ent_col[ll]:
        return int(tmp)
    if tmp > ent_col[rr] - rr + ll - 1:
        return int(tmp + rr - ll + 1)
    lef, rig = ll, rr + 1
    while lef + 1 < rig:
        mid = (lef + rig) >> 1
        if ent_col[mid] - mid + ll - 1 < tmp:
            lef = mid
        else:
            rig = mid
    return int(tmp + lef - ll + 1)


# ------------------------------------------------------------------------------------------------
# parallel restatement = oracle of ge_bernoulli_corrupt_batch
# ------------------------------------------------------------------------------------------------
TAG_BSIDE = 0x62736964  # 'bsid'
TAG_BPICK = 0x62706963  # 'bpic'


class BernoulliIndex:
    """Sorted known-triple index over the shared holE.py table: entities are rows
    [ent_lo, ent_lo + n_ent); triples are (h, t, r).  Arrays are what the GPU kernel consumes."""

    def __init__(self, triples, ent_lo, n_ent, n_rel):
        tri = np.unique(np.asarray(triples, dtype=np.int64), axis=0)
        self.ent_lo, self.n_ent, self.n_rel = int(ent_lo), int(n_ent), int(n_rel)
        bh = _sorted_by(tri, (0, 2, 1))      # by (h, r, t)
        bt = _sorted_by(tri, (1, 2, 0))      # by (t, r, h)
        self.bh_key = (bh[:, 0] * n_rel + bh[:, 2]).astype(np.int64)   # (h, r) packed
        self.bh_ent = bh[:, 1].astype(np.int32)                        # tails, ascending inside a key
        self.bt_key = (bt[:, 1] * n_rel + bt[:, 2]).astype(np.int64)
        self.bt_ent = bt[:, 0].astype(np.int32)
        # tails per head / heads per tail, per relation (Wang et al. 2014), defects of init.cpp fixed
        rel = tri[:, 2]
        freq = np.bincount(rel, minlength=n_rel).astype(np.float64)
        n_hr = np.bincount(np.unique(tri[:, [0, 2]], axis=0)[:, 1], minlength=n_rel).astype(np.float64)
        n_tr = np.bincount(np.unique(tri[:, [1, 2]], axis=0)[:, 1], minlength=n_rel).astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            tph = np.where(n_hr > 0, freq / n_hr, 1.0)    # left_mean
            hpt = np.where(n_tr > 0, freq / n_tr, 1.0)    # right_mean
        # P(corrupt the TAIL) = hpt / (hpt + tph), as a 32-bit threshold for a uniform word
        p = hpt / (hpt + tph)
        self.tail_threshold = np.minimum(np.floor(p * 4294967296.0), 4294967295.0).astype(np.uint32)


def bernoulli_corrupt_batch(pos, index: BernoulliIndex, seed: int, step: int):
    """neg[i] = pos[i] with the tail (prob hpt/(hpt+tph) of its relation) or the head replaced by a
    uniformly drawn entity that does NOT form a known triple (init.cpp:159-246 semantics)."""
    pos = np.asarray(pos, dtype=np.int32)
    B = len(pos)
    s_lo, s_hi = step & 0xFFFFFFFF, (step >> 32) & 0xFFFFFFFF
    k_lo, k_hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    rows = np.arange(B, dtype=np.uint64)
    w_side = philox4x32_10(s_lo, s_hi, rows & 0xFFFFFFFF, rows >> 32, k_lo ^ TAG_BSIDE, k_hi)[0]
    w_pick = philox4x32_10(s_lo, s_hi, rows & 0xFFFFFFFF, rows >> 32, k_lo ^ TAG_BPICK, k_hi)[0]
    neg = pos.copy()
    for i in range(B):
        h, t, r = (int(v) for v in pos[i])
        if not (0 <= r < index.n_rel):
            neg[i] = (-1, -1, -1)
            continue
        corrupt_tail_side = int(w_side[i]) < int(index.tail_threshold[r])
        if corrupt_tail_side:
            key_arr, ent_arr, key, col = index.bh_key, index.bh_ent, h * index.n_rel + r, 1
        else:
            key_arr, ent_arr, key, col = index.bt_key, index.bt_ent, t * index.n_rel + r, 0
        ll = int(np.searchsorted(key_arr, key, side="left"))
        rr = int(np.searchsorted(key_arr, key, side="right")) - 1
        cnt = max(rr - ll + 1, 0)
        free = index.n_ent - cnt
        if free <= 0:
            neg[i, col] = -1
            continue
        tmp = (int(w_pick[i]) * free) >> 32
        if cnt == 0:
            j = tmp
        else:
            j = _skip_known(ent_arr.astype(np.int64) - index.ent_lo, ll, rr, tmp)
        neg[i, col] = index.ent_lo + j
    return neg
