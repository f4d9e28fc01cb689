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


def expected_shard_plan(pos, neg, N, G, rank, layout):
    """NumPy model of ge_shard_plan for ONE step (include/ge_hip.h): the record (items over own rows + staged rows
    R + u, slot_item tags -2 / -3 - u), pos_src [B,3], neg_src [B], req_row [U], counts [G] (column `rank` = distinct
    own rows)."""
    stride, n_sub, S, off_slot, off_sub, sub_stride, off_items, off_islots = [int(v) for v in layout]
    B = len(pos)
    P = 4 * S
    R = (N + G - 1) // G
    vrow = lambda i: (i // G) if i % G == rank else R * (1 + i % G) + i // G
    slot_item = np.full(6 * B, -1, np.int32)
    pos_src = np.full((B, 3), -1, np.int32)
    neg_src = np.full(B, -1, np.int32)
    rows, slots = [], []
    for i in range(B):
        p, n = pos[i], neg[i]
        bad = (p < 0).any() or (p >= N).any() or n[0] < 0 or n[1] < 0 or n[0] >= N or n[1] >= N
        bad = bad or n[2] != p[2] or (n[0] != p[0] and n[1] != p[1])
        if bad:
            continue
        for X in range(3):
            rows.append(vrow(int(p[X]))); slots.append(6 * i + X)
        if n[0] != p[0]:
            rows.append(vrow(int(n[0]))); slots.append(6 * i + 3)
        elif n[1] != p[1]:
            rows.append(vrow(int(n[1]))); slots.append(6 * i + 4)
    rows = np.asarray(rows, np.int64); slots = np.asarray(slots, np.int64)
    order = np.lexsort((slots, rows))
    rows, slots = rows[order], slots[order]
    items = [[] for _ in range(n_sub)]
    islots = [[] for _ in range(n_sub)]
    req_row, counts = [], np.zeros(G, np.int64)
    k = 0
    while k < len(rows):
        e = k
        while e < len(rows) and rows[e] == rows[k]:
            e += 1
        run, r = e - k, int(rows[k])
        if r >= R:
            u = len(req_row)
            req_row.append(r % R)
            counts[r // R - 1] += 1
            src, tag = R + u, -3 - u
        else:
            counts[rank] += 1
            src, tag = r, -2
        for sl in slots[k:e]:
            i, X = divmod(int(sl), 6)
            if X < 3:
                pos_src[i, X] = src
            else:
                neg_src[i] = (src << 1) | (X - 3)
        if run == 1:
            slot_item[slots[k]] = tag
        else:
            for a in range(k, e, ITEM_CAP):
                b = min(e, a + ITEM_CAP)
                items[a // P].append((src, (b - a) | ((1 << 30) if run > ITEM_CAP else 0)))
                islots[a // P].append(list(slots[a:b]) + [-1] * (ITEM_CAP - (b - a)))
        k = e
    return {"n_items": [len(v) for v in items], "items": [np.asarray(v, np.int32).reshape(-1, 2) for v in items],
            "islots": [np.asarray(v, np.int32).reshape(-1, ITEM_CAP) for v in islots], "slot_item": slot_item,
            "pos_src": pos_src, "neg_src": neg_src, "req_row": np.asarray(req_row, np.int32), "counts": counts}


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
