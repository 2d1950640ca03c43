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
