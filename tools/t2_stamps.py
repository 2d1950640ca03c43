#!/usr/bin/env python3
"""Where a pair of column blocks spends its life in k_spgemm_tile2 (NTPOLY_AMD_T2_STAMPS=<file>): per sampled pair and wave,
s_memtime at 0 entry, 1 prologue done, 2/4 slab loop done, 3/5 slab epilogue done, 6 all slabs done, 7 exit."""
import sys
import numpy as np
st = np.fromfile(sys.argv[1], dtype=np.int64).reshape(128, 16, 8)
nw = 16 if st[0, 15, 0] != 0 else 12
st = st[:, :nw, :]
rows = []
for p in range(128):
    s = st[p]
    if s[0, 0] == 0 or s[:, 7].min() == 0:
        continue
    t0 = s[:, 0].min()
    total = s[:, 7].max() - t0
    prol = s[:, 1].max() - t0
    loop1 = (s[:, 2] - s[:, 1])
    epi1 = (s[:, 3] - s[:, 2])
    has2 = s[:, 4] > 0
    loop2 = np.where(has2, s[:, 4] - s[:, 3], 0)
    epi2 = np.where(has2, s[:, 5] - s[:, 4], 0)
    done = s[:, 6] - t0
    rows.append((total, prol, loop1.max(), loop1.mean(), epi1.mean(), loop2.max(), epi2[has2].mean() if has2.any() else 0, done.min(), done.max(), s[:, 7].max() - s[:, 6].max()))
a = np.array(rows, dtype=float)
names = ["total", "prologue", "loop1 max", "loop1 mean", "epi1 mean", "loop2 max", "epi2 mean", "first wave done", "last wave done", "end stage"]
print("%d pairs sampled (s_memtime ticks, 100 MHz: 1 tick = 10 ns ~ 24 cycles)" % len(a))
for i, nme in enumerate(names):
    print("%-16s median %8.0f  mean %8.0f" % (nme, np.median(a[:, i]), a[:, i].mean()))
if len(sys.argv) > 2:
    for p in (int(x) for x in sys.argv[2].split(",")):
        s = st[p]
        t0 = s[:, 0].min()
        print("pair sample %d: wave: entry, prologue done, [slab loop done, epilogue done] x2, all done, exit (cycles from the first entry)" % p)
        for w in range(nw):
            print("  wave %2d: %s" % (w, " ".join("%7d" % (x - t0 if x > 0 else -1) for x in s[w])))
