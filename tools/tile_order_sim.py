#!/usr/bin/env python3
"""List-scheduling simulation of a launch on 7168 wave slots (1024 SIMDs x 7) with per-tile wave durations of the C3
frame (profiles/r02_tile_durations.npz: shader clocks per 8x8 tile for the camera at z + 0, 0.1, 0.2, 0.4, 0.8, recorded
on an MI355X by a build whose frame kernel stored one duration per tile): how long the launch takes for several tile
orders, with durations that are current and with durations recorded at another camera position (a stale order).
Result: a per-tile longest-first order is worth 11 % and worthless one camera step later; blocks of 16 x 16 tiles ordered
by their longest tile keep 9 % for two steps -- what rt_scene_set_tile_order does."""
import heapq, os, sys
import numpy as np
d = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_tile_durations.npz"))
SLOTS = 7168


def makespan(cost, order):
    h = [0] * SLOTS
    heapq.heapify(h)
    end = 0
    for i in order:
        t = heapq.heappop(h) + int(cost[i])
        end = max(end, t)
        heapq.heappush(h, t)
    return end


def lpt(c):
    return np.argsort(-c, kind="stable")


def classes(c, thr=(3.0, 1.5), dil=0, shape=None):
    """tiles in classes by duration relative to the mean (heaviest class first), row-major inside a class; dil: a tile
    takes the heaviest class found within +-dil tiles"""
    m = c.mean()
    k = np.zeros(c.shape, dtype=np.int64)
    for j, t in enumerate(sorted(thr)):
        k[c >= t * m] = j + 1
    if dil:
        k2 = k.reshape(shape).copy()
        kk = k.reshape(shape)
        for dy in range(-dil, dil + 1):
            for dx in range(-dil, dil + 1):
                sh = np.roll(np.roll(kk, dy, 0), dx, 1)
                k2 = np.maximum(k2, sh)
        k = k2.reshape(-1)
    return np.argsort(-k, kind="stable")


def blocks(c, shape, b=8):
    """blocks of b x b tiles ordered by their heaviest tile, row-major inside a block"""
    h, w = shape
    c2 = c.reshape(shape)
    by, bx = (h + b - 1) // b, (w + b - 1) // b
    key = np.zeros((by, bx))
    for y in range(by):
        for x in range(bx):
            key[y, x] = c2[y * b:(y + 1) * b, x * b:(x + 1) * b].max()
    order_b = np.argsort(-key.reshape(-1), kind="stable")
    idx = np.arange(h * w).reshape(shape)
    out = []
    for q in order_b:
        y, x = divmod(int(q), bx)
        out.append(idx[y * b:(y + 1) * b, x * b:(x + 1) * b].reshape(-1))
    return np.concatenate(out)


cur = d["dz0.0"]
shape = cur.shape
c0 = cur.reshape(-1).astype(np.int64)
ideal = c0.sum() / SLOTS
print("tiles", c0.size, "sum/slots", int(ideal), "max", c0.max())
print("grid order           ", makespan(c0, np.arange(c0.size)) / ideal)
for name in ("dz0.0", "dz0.1", "dz0.2", "dz0.4", "dz0.8"):
    src = d[name].reshape(-1).astype(np.int64)
    res = {"lpt": makespan(c0, lpt(src)) / ideal,
           "classes3": makespan(c0, classes(src)) / ideal,
           "classes3_dil2": makespan(c0, classes(src, dil=2, shape=shape)) / ideal,
           "classes3_dil4": makespan(c0, classes(src, dil=4, shape=shape)) / ideal,
           "classes(4,2,1.25)_dil3": makespan(c0, classes(src, thr=(4.0, 2.0, 1.25), dil=3, shape=shape)) / ideal,
           "blocks8": makespan(c0, blocks(src, shape, 8)) / ideal,
           "blocks4": makespan(c0, blocks(src, shape, 4)) / ideal}
    print("durations from", name, {k: round(v, 3) for k, v in res.items()})

print("-- block sizes, key = max / mean of the block")
def blocks_key(c, shape, b, keyf):
    h, w = shape
    c2 = c.reshape(shape)
    by, bx = (h + b - 1) // b, (w + b - 1) // b
    key = np.array([[keyf(c2[y * b:(y + 1) * b, x * b:(x + 1) * b]) for x in range(bx)] for y in range(by)])
    order_b = np.argsort(-key.reshape(-1), kind="stable")
    idx = np.arange(h * w).reshape(shape)
    return np.concatenate([idx[(q // bx) * b:(q // bx + 1) * b, (q % bx) * b:(q % bx + 1) * b].reshape(-1) for q in order_b])
for b in (4, 6, 8, 12, 16, 24):
    row = {}
    for name in ("dz0.0", "dz0.1", "dz0.2", "dz0.4"):
        src = d[name].reshape(-1).astype(np.int64)
        row[name] = (round(makespan(c0, blocks_key(src, shape, b, np.max)) / ideal, 3), round(makespan(c0, blocks_key(src, shape, b, np.mean)) / ideal, 3))
    print("block", b, row)
