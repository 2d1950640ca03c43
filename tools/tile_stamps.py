#!/usr/bin/env python3
"""Decode the s_memtime stamps of a diagnostic build (NTPOLY_AMD_EXTRA_FLAGS=-DNTP_TILE_STAMPS, NTP_TILE_STAMPS_FILE=...)."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(64, 8, 64)
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = []
for blk in range(64):
    s = a[blk]
    if s[0, 0] == 0:
        continue
    t0 = s[:nw, 0].min()
    for w in range(nw):
        st = s[w]
        tiles = []
        i = 4
        while i + 3 < 64 and st[i] > 0:
            tiles.append((st[i] - t0, st[i + 1] - st[i], st[i + 2] - st[i + 1], st[i + 3] - st[i + 2]))
            i += 4
        rows.append((blk, w, st[1] - t0, st[2] - t0, st[3] - t0, tiles))
for blk, w, pro, loopend, bar, tiles in rows[: 4 * nw]:
    print("block %2d wave %d: prologue done %6d | tiles done %7d | barrier %7d | tiles (start, fill, loop, epilogue): %s" % (
        blk, w, pro, loopend, bar, " ".join("(%d,%d,%d,%d)" % t for t in tiles)))
pro = np.array([r[2] for r in rows]); le = np.array([r[3] for r in rows]); bar = np.array([r[4] for r in rows])
fill = np.array([t[1] for r in rows for t in r[5]]); loop = np.array([t[2] for r in rows for t in r[5]]); epi = np.array([t[3] for r in rows for t in r[5]])
print("prologue phases (B copy issued, records, group ranges, tile ranges):", [float((a[:, :nw, i] - a[:, :nw, 0])[a[:, :nw, 0] > 0].mean()) for i in (56, 57, 58, 60, 61, 62, 63)])
print("records requested / tile requested (stamps 50, 52 = the block's column extents have arrived, 51):", [float((a[:, :nw, i] - a[:, :nw, 0])[(a[:, :nw, 0] > 0) & (a[:, :nw, i] > 0)].mean()) if ((a[:, :nw, 0] > 0) & (a[:, :nw, i] > 0)).any() else None for i in (50, 52, 51)])
end = a[:, :nw, 55]; b3 = a[:, :nw, 3]
print("end-of-block phase (after the barrier): %.0f" % float((end - b3)[(end > 0) & (b3 > 0)].mean()))
print("mean: prologue %.0f, tiles done %.0f, barrier %.0f | per tile fill %.0f loop %.0f epilogue %.0f (n=%d)" % (
    pro.mean(), le.mean(), bar.mean(), fill.mean(), loop.mean(), epi.mean(), len(fill)))
