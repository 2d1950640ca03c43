#!/usr/bin/env python3
"""A MODEL of the 1 / 2 / 4 / 8-GPU strong-scaling curve of the headline workload from measurements on ONE GPU (no
multi-GPU node is available to this project's boxes; the real curve is the driver's SCALE_rNN.json when it has one).

One rank of P owns N / P columns of every matrix.  Its per-step work is measured here for what it is on the device:
TRS2 steps on a banded operand of dimension N / P with the same half bandwidth (a panel of N / P columns of an N-wide
band has the work of an (N / P)-wide band, boundary columns aside), through the same entry point as bench.py, with every
launch and every host read-back of a one-rank step inside the timed region.  A panel step then adds, per step:

  * round 4: one more host round trip than the one-rank step (exchange layout + plan) -- measured as the wall time of a
    device-to-host read-back of one scalar on the idle stream.  Round 5 (option exchange_ahead, the default): a panel step
    prepares the next step's exchange on its own read-back, so the round trip is gone; what remains on the critical path is
    the preparation's five small launches (request, extents, scan, counts, plan statistics: 2 us each behind a running
    stream, MI355X_MICROARCH.md "boundary") and the all-gather below;
  * the halo: the runs of the columns within one bandwidth of the panel's two edges, sent over two xGMI links at once,
    bytes / (153 GB/s per link);
  * the all-gather of one 8-byte extent record per column of the whole matrix over a ring of P - 1 hops.

    python3 tools/rank_share.py [--n 262144] [--arithmetic fma] > profiles/r03_rank_share.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

XGMI_LINK_GBS = 153.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=262144)
    ap.add_argument("--halfband", type=int, default=100)
    ap.add_argument("--threshold", type=float, default=1e-8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--arithmetic", choices=("fma", "unfused"), default="fma")
    args = ap.parse_args()
    import ntpoly_amd as nt
    from gen import banded_triplets
    from bench import trs2_step
    nt.init_comm()
    nt.ConstructGlobalProcessGrid(1, 1, 1)
    nt.set_option("time_kernels", 1)
    nt.set_option("spgemm_fma", 1 if args.arithmetic == "fma" else 0)
    h, thr = args.halfband, args.threshold

    # one host round trip: a scalar read back from an idle stream
    tiny = nt.Matrix_ps(64)
    tiny.FillIdentity()
    tiny.Trace()
    nt.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        tiny.Trace()
    sync_us = (time.perf_counter() - t0) / 200 * 1e6

    rows = []
    for P in (1, 1, 2, 4, 8):   # (the first pass warms the allocator and the clocks up and is dropped)
        n = args.n // P
        col, row, val = banded_triplets(n, h)
        H = nt.Matrix_ps.from_triplets(n, col, row, val)
        del col, row, val
        e_min, e_max = nt.EigenBounds.GershgorinBounds(H)
        Ident = nt.Matrix_ps(n)
        Ident.FillIdentity()
        X = nt.Matrix_ps(H)
        X.Scale(-1.0)
        X.Increment(Ident, e_max, 0.0)
        X.Scale(1.0 / (e_max - e_min))
        X2 = nt.Matrix_ps(n)
        pool = nt.PMatrixMemoryPool(H)
        tr = None
        for _ in range(args.warmup):
            _, _, tr = trs2_step(nt, X, X2, H, pool, n / 2.0, thr, tr)
        nt.reset_spgemm_accum()
        s0 = nt.exchange_stats()[2]
        nt.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            _, energy, tr = trs2_step(nt, X, X2, H, pool, n / 2.0, thr, tr)
        nt.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        syncs = (nt.exchange_stats()[2] - s0) / args.steps
        acc = nt.spgemm_accum()
        per_col = X.GetSize() / n                       # entries per column of the iterate = rows of a run
        halo_bytes = 2.0 * (per_col / 2.0) * per_col * 8.0   # both edges: (bandwidth of X) columns x (run length) x 8 B
        t_halo_us = (halo_bytes / 2.0) / (XGMI_LINK_GBS * 1e3) if P > 1 else 0.0      # two links in parallel
        t_gather_us = (8.0 * args.n * (P - 1) / P) / (XGMI_LINK_GBS * 1e3) if P > 1 else 0.0
        ahead = nt.get_option("exchange_ahead") != 0
        extra_sync_us = (10.0 if ahead else sync_us) if P > 1 else 0.0   # (ahead: five small launches instead of a round trip)
        model_ms = dt * 1e3 + (extra_sync_us + t_halo_us + t_gather_us) * 1e-3
        rows.append(dict(ranks=P, panel_columns=n, measured_share_ms_per_step=dt * 1e3,
                         kernel_ms_per_step=acc["ms_numeric"] / max(1, acc["calls"]),
                         host_syncs_per_step_measured=syncs, halo_bytes_per_step=halo_bytes, halo_us=t_halo_us,
                         extent_allgather_us=t_gather_us, extra_host_round_trip_us=extra_sync_us,
                         modelled_ms_per_step=model_ms, modelled_iters_per_s=1e3 / model_ms))
        # ---- TRS4 (two products per iteration, the loop's matrices in slab form: a slab session on one rank, a session of
        # column panels on several -- psmatrix.cpp panel_slab_multiply): the one-rank loop at N / P measured; each of a panel
        # product's exchanges adds the halo of its left operand, the extent all-gather, the reduction of "every rank took its
        # panel" (4 doubles) and eight small launches (request, extents + scan, counts, plan, pack, layout), and the two
        # reductions of the traces ride on the loop's read-backs as on one rank
        p4 = nt.SolverParameters()
        p4.SetThreshold(thr)
        p4.SetConvergeDiff(1e-30)
        p4.SetMaxIterations(12)
        p4.SetMonitorConvergence(False)
        K = nt.Matrix_ps(n)
        for _ in range(2):   # (the second solve is the warm one)
            nt.DensityMatrixSolvers.TRS4(H, Ident, n / 2.0, K, p4)
        tr4 = nt.solver_trace()
        trs4_ms = tr4["loop_ms"] / max(1, tr4["iterations"])
        per_col4 = K.GetSize() / n
        halo4 = 2.0 * (per_col4 / 2.0) * per_col4 * 8.0
        t_prod_us = ((halo4 / 2.0) / (XGMI_LINK_GBS * 1e3) + t_gather_us + 16.0 + 5.0) if P > 1 else 0.0
        rows[-1].update(trs4_measured_share_ms_per_iteration=trs4_ms, trs4_halo_bytes_per_product=halo4,
                        trs4_exchange_us_per_product=t_prod_us, trs4_modelled_ms_per_iteration=trs4_ms + 2.0 * t_prod_us * 1e-3)
        del H, X, X2, Ident, pool, K
        # ---- the same share through the PANEL path, measured: a child process with a 1-rank RCCL communicator
        # (NTPOLY_AMD_FORCE_RCCL=1: every collective of a panel step is a real RCCL call, short-circuited by RCCL on one rank) times
        # bench.py at N / P -- the exchange's preparation, its small kernels and uploads, the reduction record are all in it; what a
        # real run adds on top is the halo's transfer and the collectives' own kernels
        if P > 1:
            import subprocess
            env = dict(os.environ, NTPOLY_AMD_FORCE_RCCL="1")
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", str(n), "--steps", str(args.steps), "--warmup",
                                str(args.warmup), "--blocks", "3", "--no-cpu-baseline", "--no-wrp-check"], env=env, cwd=ROOT,
                               stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if line:
                ms = json.loads(line[-1])["ms_per_step"]
                rows[-1].update(panel_step_ms_through_1rank_rccl=ms,
                                estimate_with_measured_machinery_ms=ms + (t_halo_us + t_gather_us) * 1e-3)
    rows = rows[1:]
    base = rows[0]["modelled_ms_per_step"]
    for r in rows:
        r["modelled_speedup"] = base / r["modelled_ms_per_step"]
        r["modelled_efficiency"] = r["modelled_speedup"] / r["ranks"]
        if "estimate_with_measured_machinery_ms" in r:
            r["estimate_with_measured_machinery_iters_per_s"] = 1e3 / r["estimate_with_measured_machinery_ms"]
            r["estimate_with_measured_machinery_efficiency"] = base / r["estimate_with_measured_machinery_ms"] / r["ranks"]
        r["trs4_modelled_speedup"] = rows[0]["trs4_modelled_ms_per_iteration"] / r["trs4_modelled_ms_per_iteration"]
        r["trs4_modelled_efficiency"] = r["trs4_modelled_speedup"] / r["ranks"]
    print(json.dumps({
        "what": "MODEL, not a measurement of several GPUs: one rank's share of a P-rank TRS2 step measured on ONE MI355X "
                "(banded operand of dimension N / P, every launch and read-back of the step included) + the preparation of the next "
                "step's exchange (five small launches; a host round trip with exchange_ahead = 0) + halo bytes / 153 GB/s per xGMI "
                "link + a ring all-gather of 8 B per column.  A RELABELLED operand costs the same per step on several ranks: the "
                "solver recovers the band once per solve and redistributes the operands (csrc/band_scope.cpp).  TRS4 rows: the one-rank "
                "loop at N / P (slab session) + per product the halo of the left operand, the extent all-gather, eight small launches "
                "and a 4-double reduction (sessions of column panels, option panel_sessions; tests/test_gpu_panel_sessions.py measures two "
                "host round trips per product, as on one rank).  estimate_with_measured_machinery_*: the panel step itself timed in a child "
                "process with a 1-rank RCCL communicator (its preparation kernels, uploads and reduction record included, RCCL's own kernels and "
                "the transfer not) + the modelled halo and all-gather times -- the more realistic of the two TRS2 estimates",
        "n": args.n, "halfband": h, "threshold": thr, "arithmetic": args.arithmetic,
        "host_round_trip_us": sync_us, "xgmi_link_GBps": XGMI_LINK_GBS, "table": rows}, indent=1))


if __name__ == "__main__":
    main()
