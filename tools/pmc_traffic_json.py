#!/usr/bin/env python3
"""HBM traffic per launch of one kernel from the two PMC passes of tools/collect_profiles.sh:
   python tools/pmc_traffic_json.py gpurun_out/<tag> k_spgemm_slab [out.json]
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE under-counts 64-byte reads by 2x (MI355X_MICROARCH.md, HBM
section; calibrated on k_scale, which reads and writes the same number of bytes)."""
import csv
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sources_sha16():
    """fingerprint of the kernel sources the figure belongs to (bench.py reports the traffic only for these sources)"""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "ntpoly_amd", "csrc")
    # every device source and every header / generated loop they include: a figure belongs to ALL of them
    for f in sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".hpp", ".inc"))):
        h.update(f.encode())
        with open(os.path.join(csrc, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def per_kernel(path):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        m = re.search(r"(k_[a-z0-9_]+|__amd_rocclr_\w+)", r["Kernel_Name"])
        acc[m.group(1) if m else r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
    return acc


def main():
    root, pat = sys.argv[1], sys.argv[2]
    f = per_kernel(root + "/pmc_FETCH_SIZE/run_counter_collection.csv")
    w = per_kernel(root + "/pmc_WRITE_SIZE/run_counter_collection.csv")
    # (a comma-separated list of candidates: the one with the most counted bytes is the dominant kernel of the run)
    cands = []
    for q in pat.split(","):
        cands += [q] if q in f else [k for k in f if q in k]
    if not cands:
        raise SystemExit("no kernel matching %r in the counter files" % pat)
    fk = max(cands, key=lambda k: sum(f[k]))
    fetch, write = f[fk][1:] or f[fk], w[fk][1:] or w[fk]   # drop the first (cold) launch
    fa, wa = sum(fetch) / len(fetch), sum(write) / len(write)
    cal = None
    if "k_scale" in f and "k_scale" in w:
        cal = (sum(w["k_scale"]) / len(w["k_scale"])) / (sum(f["k_scale"]) / len(f["k_scale"]))
    out = {"kernel": fk, "launches": len(fetch), "fetch_size_KB_avg": fa, "write_size_KB_avg": wa, "fetch_correction": 2.0,
           "k_scale_write_over_fetch": cal, "hbm_bytes_per_launch": (2.0 * fa + wa) * 1024.0,
           "sources_sha16": sources_sha16()}
    s = json.dumps(out, indent=1)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(s + "\n")
    print(s)


if __name__ == "__main__":
    main()
