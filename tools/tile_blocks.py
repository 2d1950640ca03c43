#!/usr/bin/env python3
"""Per-block durations of the tile kernel from a diagnostic build (-DNTP_TILE_STAMPS, NTP_TILE_STAMPS_FILE=f -> f.blocks)."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 4)
ok = a[:, 1] > 0
d = (a[:, 1] - a[:, 0])[ok]
t0 = a[ok, 0].min()
print("blocks %d, kernel span %d ticks; block duration mean %.0f median %.0f p99 %.0f max %d" % (
    ok.sum(), a[ok, 1].max() - t0, d.mean(), np.median(d), np.percentile(d, 99), d.max()))
order = np.argsort(-d)[:12]
idx = np.nonzero(ok)[0]
for i in order:
    b = idx[i]
    print("  block %6d: %7d ticks, start %8d, deferred %3d, k range %d" % (b, d[i], a[b, 0] - t0, a[b, 2], a[b, 3]))
nd = a[ok, 2]
print("deferred per block: mean %.1f max %d; blocks with > 32: %d" % (nd.mean(), nd.max(), (nd > 32).sum()))
# how busy the device is over the kernel's span (blocks in flight per 5 % slice)
span = a[ok, 1].max() - t0
edges = np.linspace(0, span, 21)
s = a[ok, 0] - t0; e = a[ok, 1] - t0
print("blocks in flight at 5 % steps:", [int(((s <= x) & (e > x)).sum()) for x in edges[:-1]])
