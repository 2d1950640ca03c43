"""CPU, world_size 2 over gloo: the column-panel decomposition of the distributed multiply.

The RCCL data path itself needs GPUs; what can be checked here is that the decomposition the
engine uses is *correct by construction*: panel ranges come from the engine's own C ABI
(ntpoly_amd_panel_range), each rank multiplies the gathered left operand with ITS column panel of
the right operand (computed here by the oracle, as the checker), panels are exchanged with the same
three-array all-gather the engine performs (offsets, indices, values), and the concatenation must be
bit-identical to the single-process product -- the property that makes the GPU-count-independent
result of the engine possible (DESIGN.md "Multi-GPU").  The exchange is the range-restricted one:
segment boundaries come from the engine's ntpoly_amd_halo_segment."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    try:
        import ctypes as C
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import ntpoly_amd as nt
        from gen import banded_triplets
        from oracle import oracle_py as O
        n, h, thr = 601, 9, 1e-6
        col, row, val = banded_triplets(n, h)
        a, b = C.c_int(), C.c_int()
        nt.lib.ntpoly_amd_panel_range(nt.capi.i(n), nt.capi.i(world), nt.capi.i(rank), C.byref(a), C.byref(b))
        c0, c1 = a.value, b.value
        # my panel of A and of B (columns [c0, c1)), as the engine stores them
        m = (col - 1 >= c0) & (col - 1 < c1)
        # --- range-restricted ("halo") exchange, as comm.cpp gather_needed does it: my B panel has rows
        # [kmin, kmax]; from every owner s I receive exactly the columns ntpoly_amd_halo_segment names
        kmin, kmax = int(row[m].min()) - 1, int(row[m].max()) - 1
        ranges = [None] * world
        dist.all_gather_object(ranges, (kmin, kmax))
        outbox = []
        for dst in range(world):  # what I owe rank dst
            sa, sb = C.c_int(), C.c_int()
            nt.lib.ntpoly_amd_halo_segment(nt.capi.i(n), nt.capi.i(world), nt.capi.i(rank), nt.capi.i(ranges[dst][0]),
                                           nt.capi.i(ranges[dst][1]), C.byref(sa), C.byref(sb))
            sel = (col - 1 >= sa.value) & (col - 1 < sb.value)
            assert np.all((col[sel] - 1 >= c0) & (col[sel] - 1 < c1))  # only columns I own
            outbox.append((col[sel], row[sel], val[sel]))
        inbox = [None] * world
        for src in range(world):  # all-to-all by scatter from every source
            recv = [None]
            dist.scatter_object_list(recv, outbox if rank == src else None, src=src)
            inbox[src] = recv[0]
        parts = [np.concatenate([inbox[s][k] for s in range(world)]) for k in range(3)]
        assert parts[0].min() - 1 >= kmin and parts[0].max() - 1 <= kmax
        assert len(parts[0]) < len(col)  # strictly less than the full gather for a banded operand
        A_full = O.Mat.from_triplets(n, n, parts[0], parts[1], parts[2])
        # B panel as an n x n matrix that is empty outside my columns: its product columns are my C panel
        B_mine = O.Mat.from_triplets(n, n, col[m], row[m], val[m])
        C_mine = O.ps_multiply(A_full, B_mine, None, 1.0, 0.0, thr)
        cc, cr, cv = C_mine.triplets()
        assert np.all((cc - 1 >= c0) & (cc - 1 < c1))
        # gather C panels on rank 0 and compare with the one-process product
        outs = [None] * world
        dist.gather_object((cc, cr, cv), outs if rank == 0 else None, dst=0)
        if rank == 0:
            full = O.Mat.from_triplets(n, n, col, row, val)
            ref = O.ps_multiply(full, full, None, 1.0, 0.0, thr).triplets()
            got = [np.concatenate([o[k] for o in outs]) for k in range(3)]
            ok = all(np.array_equal(g, r) for g, r in zip(got, ref))
            q.put(("ok" if ok else "mismatch", len(ref[0])))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put(("error", traceback.format_exc()))
        raise


@pytest.mark.timeout(300)
def test_column_panel_multiply_world2():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    status, info = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
    assert status == "ok", info
    assert info > 10000


def _worker_layout(rank, world, port, q):
    """the halo exchange of the FUSED panel steps (csrc/psmatrix.cpp slab_exchange_and_step) with the engine's own layout
    arithmetic: every rank's request and the send counts are gathered as the engine gathers them, the ENGINE
    (ntpoly_amd_panel_exchange_layout = panel_exchange_layout, the function the GPU path calls) says which of my columns go to
    whom at which offset of the send buffer and where every owner's segment lands in my receive buffer; the columns travel as
    dense runs (first .. last row, holes = 0: the slab form), packed and unpacked by exactly those numbers over gloo
    send / recv pairs; the product of the assembled halo operand with my panel (oracle) must be my panel of the one-process
    product, bit for bit."""
    try:
        import ctypes as C
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import ntpoly_amd as nt
        from gen import banded_triplets
        from oracle import oracle_py as O
        n, h, thr = 777, 11, 1e-6
        col, row, val = banded_triplets(n, h)
        keep = (np.random.default_rng(5).random(len(val)) > 0.2) | (col == row)      # (holes inside the runs)
        col, row, val = col[keep], row[keep], val[keep]
        a, b = C.c_int(), C.c_int()
        nt.lib.ntpoly_amd_panel_range(nt.capi.i(n), nt.capi.i(world), nt.capi.i(rank), C.byref(a), C.byref(b))
        c0, c1 = a.value, b.value
        m = (col - 1 >= c0) & (col - 1 < c1)
        # slab form of my panel: column j = the dense run first[j] .. last[j] (0-based rows), zeros = no entry
        first = np.full(c1 - c0, n, dtype=np.int64)
        last = np.full(c1 - c0, -1, dtype=np.int64)
        np.minimum.at(first, col[m] - 1 - c0, row[m] - 1)
        np.maximum.at(last, col[m] - 1 - c0, row[m] - 1)
        span = np.where(last >= first, last - first + 1, 0)
        pre = np.concatenate(([0], np.cumsum(span)))
        runs = np.zeros(pre[-1])
        runs[pre[col[m] - 1 - c0] + (row[m] - 1) - first[col[m] - 1 - c0]] = val[m]
        # what the step's all-gather carries: every rank's request (rows of its panel of B) and the extents of all columns
        kmin, kmax = int(row[m].min()) - 1, int(row[m].max()) - 1
        gathered = [None] * world
        dist.all_gather_object(gathered, (kmin, kmax, int(m.sum()), first, last))
        req = np.zeros(4 * world, dtype=np.int64)
        for s in range(world):
            req[4 * s], req[4 * s + 1], req[4 * s + 2] = gathered[s][0], gathered[s][1], gathered[s][2]
        # who sends how many doubles to whom (the engine: halo_counts_async on the gathered prefix sums)
        cnt = np.zeros(world * world, dtype=np.int64)
        seg = {}
        for s in range(world):
            sc0 = C.c_int(); sc1 = C.c_int()
            nt.lib.ntpoly_amd_panel_range(nt.capi.i(n), nt.capi.i(world), nt.capi.i(s), C.byref(sc0), C.byref(sc1))
            fs, ls = gathered[s][3], gathered[s][4]
            sp = np.where(ls >= fs, ls - fs + 1, 0)
            for qq in range(world):
                x, y = C.c_int(), C.c_int()
                nt.lib.ntpoly_amd_halo_segment(nt.capi.i(n), nt.capi.i(world), nt.capi.i(s), nt.capi.i(int(req[4 * qq])),
                                               nt.capi.i(int(req[4 * qq + 1])), C.byref(x), C.byref(y))
                seg[(s, qq)] = (x.value, y.value, sc0.value)
                if s != qq and y.value > x.value:
                    cnt[s * world + qq] = int(sp[x.value - sc0.value:y.value - sc0.value].sum())
        # ---- the engine's layout
        sa, sb, ra, rb = ((C.c_int * world)() for _ in range(4))
        soff, zoff = (C.c_longlong * (world + 1))(), (C.c_longlong * (world + 1))()
        nt.lib.ntpoly_amd_panel_exchange_layout(nt.capi.i(n), nt.capi.i(world), nt.capi.i(rank), req.ctypes.data_as(C.POINTER(C.c_longlong)),
                                                cnt.ctypes.data_as(C.POINTER(C.c_longlong)), sa, sb, soff, ra, rb, zoff)
        for qq in range(world):
            assert (sa[qq], sb[qq]) == seg[(rank, qq)][:2] and (ra[qq], rb[qq]) == seg[(qq, rank)][:2]
            assert soff[qq + 1] - soff[qq] == (0 if qq == rank else cnt[rank * world + qq])
            assert zoff[qq + 1] - zoff[qq] == (0 if qq == rank else cnt[qq * world + rank])
        # ---- pack, exchange, unpack by those numbers
        sendbuf = np.zeros(max(1, soff[world]))
        for qq in range(world):
            if qq != rank and sb[qq] > sa[qq]:
                sendbuf[soff[qq]:soff[qq + 1]] = runs[pre[sa[qq] - c0]:pre[sb[qq] - c0]]
        recvbuf = np.zeros(max(1, zoff[world]))
        ops = []
        for peer in range(world):
            if peer == rank:
                continue
            if soff[peer + 1] > soff[peer]:
                ops.append(dist.isend(torch.from_numpy(sendbuf[soff[peer]:soff[peer + 1]].copy()), dst=peer))
        for peer in range(world):
            if peer != rank and zoff[peer + 1] > zoff[peer]:
                t = torch.empty(int(zoff[peer + 1] - zoff[peer]), dtype=torch.float64)
                dist.recv(t, src=peer)
                recvbuf[zoff[peer]:zoff[peer + 1]] = t.numpy()
        for o in ops:
            o.wait()
        # the halo operand: columns kmin .. kmax of A from the received runs (and my own), as triplets
        acs, ars, avs = [], [], []
        for s in range(world):
            x, y, sc0 = ra[s], rb[s], seg[(s, rank)][2]
            if y <= x:
                continue
            fs, ls = gathered[s][3][x - sc0:y - sc0], gathered[s][4][x - sc0:y - sc0]
            sp = np.where(ls >= fs, ls - fs + 1, 0)
            src = runs[pre[x - c0]:pre[y - c0]] if s == rank else recvbuf[zoff[s]:zoff[s + 1]]
            assert len(src) == sp.sum()
            cc = np.repeat(np.arange(x, y), sp)
            rr = np.concatenate([np.arange(f, l + 1) for f, l in zip(fs, ls) if l >= f]) if sp.sum() else np.zeros(0, dtype=np.int64)
            nz = src != 0.0
            acs.append(cc[nz] + 1); ars.append(rr[nz] + 1); avs.append(src[nz])
        A_halo = O.Mat.from_triplets(n, n, np.concatenate(acs), np.concatenate(ars), np.concatenate(avs))
        B_mine = O.Mat.from_triplets(n, n, col[m], row[m], val[m])
        cc, cr, cv = O.ps_multiply(A_halo, B_mine, None, 1.0, 0.0, thr).triplets()
        outs = [None] * world
        dist.gather_object((cc, cr, cv), outs if rank == 0 else None, dst=0)
        if rank == 0:
            full = O.Mat.from_triplets(n, n, col, row, val)
            ref = O.ps_multiply(full, full, None, 1.0, 0.0, thr).triplets()
            got = [np.concatenate([o[k] for o in outs]) for k in range(3)]
            ok = all(np.array_equal(g, r) for g, r in zip(got, ref))
            q.put(("ok" if ok else "mismatch", len(ref[0])))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put(("error", traceback.format_exc()))
        raise


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_panel_exchange_layout_of_the_engine_over_gloo(world):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker_layout, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    status, info = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
    assert status == "ok", info
    assert info > 10000
