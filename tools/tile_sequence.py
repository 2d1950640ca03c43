#!/usr/bin/env python3
"""The launches of the MFMA tile kernel in a rocprofv3 --kernel-trace database, in time order: variant and duration
(which steps of a solve run with eight waves per workgroup):  python tools/tile_sequence.py run_results.db"""
import re
import sqlite3
import sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels where name like '%k_spgemm_tile%' order by start").fetchall()
for i, (n, s, e) in enumerate(rows):
    m = re.search(r"k_spgemm_tile<([^>]*)>", n)
    print(i, (m.group(1) if m else n[:40]).replace(" ", ""), "%.1f us" % ((e - s) / 1e3))
