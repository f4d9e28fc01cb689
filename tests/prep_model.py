"""NumPy model of the prepared step record ge_train_prepare_steps builds (include/ge_hip.h): the
row-sorted list of a step's IndexedSlices slots cut into work items of <= 16 slots of one row.
Test infrastructure only."""
import numpy as np

ITEM_CAP = 16
SUB = 4096


def expected_record(pos, neg, N, direct, layout):
    """pos/neg [B,3] int32 -> dict(n_items[n_sub], items[n_sub] (k,2), islots[n_sub] (k,16), slot_item[6B] or None).
    The step's (row, slot) keys are sorted as ONE sequence; tile t of 4*S positions lists the items that start in it."""
    stride, n_sub, S, off_slot, off_sub, sub_stride, off_items, off_islots = [int(v) for v in layout]
    B = len(pos)
    P = 4 * S
    bad = ((pos < 0) | (pos >= N)).any(1) | ((neg < 0) | (neg >= N)).any(1)
    slot_item = np.full(6 * B, -1, np.int32) if direct else None
    out = {"n_items": [], "items": [[] for _ in range(n_sub)], "islots": [[] for _ in range(n_sub)], "slot_item": slot_item}
    rows, slots = [], []
    for i in range(B):
        if bad[i]:
            continue
        for X in range(3):
            rows.append(int(pos[i, X])); slots.append(6 * i + X)
        diff = np.nonzero(pos[i] != neg[i])[0]
        if len(diff):
            c = int(diff[0])
            rows.append(int(neg[i, c])); slots.append(6 * i + 3 + c)
    rows = np.asarray(rows, np.int64); slots = np.asarray(slots, np.int64)
    order = np.lexsort((slots, rows))
    rows, slots = rows[order], slots[order]
    k = 0
    while k < len(rows):
        e = k
        while e < len(rows) and rows[e] == rows[k]:
            e += 1
        run = e - k
        if direct and run == 1:
            slot_item[slots[k]] = -2
        else:
            for a in range(k, e, ITEM_CAP):
                b = min(e, a + ITEM_CAP)
                out["items"][a // P].append((rows[k], (b - a) | ((1 << 30) if run > ITEM_CAP else 0)))
                out["islots"][a // P].append(list(slots[a:b]) + [-1] * (ITEM_CAP - (b - a)))
        k = e
    out["n_items"] = [len(v) for v in out["items"]]
    out["items"] = [np.asarray(v, np.int32).reshape(-1, 2) for v in out["items"]]
    out["islots"] = [np.asarray(v, np.int32).reshape(-1, ITEM_CAP) for v in out["islots"]]
    return out


def parse_record(rec, B, layout):
    stride, n_sub, S, off_slot, off_sub, sub_stride, off_items, off_islots = [int(v) for v in layout]
    out = {"neg": rec[:3 * B].reshape(B, 3), "slot_item": rec[off_slot:off_slot + 6 * B], "n_items": [], "items": [], "islots": []}
    for sub in range(n_sub):
        s = rec[off_sub + sub * sub_stride: off_sub + (sub + 1) * sub_stride]
        n = int(s[0])
        out["n_items"].append(n)
        out["items"].append(s[off_items:off_items + 2 * n].reshape(n, 2))
        out["islots"].append(s[off_islots:off_islots + ITEM_CAP * n].reshape(n, ITEM_CAP))
    return out
