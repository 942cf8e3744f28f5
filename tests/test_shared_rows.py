"""Export(ADD) of shared-DOF rows between z-slab shards (mrhyde_amd/shared_rows.py), world_size 2, gloo on CPU.

Mirrors the reference's rank-count independence check (regression/thermal/2D_verification_mpi): the
2-shard result, after the shared-row exchange, equals the single-domain assembly row by row.
The per-shard assembly itself comes from the CPU oracle here (no GPU in this test); the exchange code
is device-agnostic torch and is the same object bench.py drives over RCCL.
"""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, dim, order, qdeg, nxy, nz_per, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import oracle_lib
    from mrhyde_amd.shared_rows import SlabExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ncell = (nxy, nxy, nz_per)
        lo, hi = [0.0, 0.0, float(rank)], [1.0, 1.0, float(rank + 1)]
        m = oracle_lib.mesh_structured(dim, order, ncell, lo, hi)
        D = order * nxy + 1
        P = D * D
        nrows = m["ndof"]
        # the global state restricted to this slab (shared plane holds identical values on both sides)
        rng = np.random.default_rng(99)
        u_glob = rng.uniform(-1, 1, P * (order * nz_per * world + 1))
        off = rank * (nrows - P)
        u = u_glob[off:off + nrows]
        fixed = m["boundary"].copy()  # physical boundary of the stacked domain only
        if rank > 0:
            fixed[:P].reshape(D, D)[1:-1, 1:-1] = 0
        if rank < world - 1:
            fixed[-P:].reshape(D, D)[1:-1, 1:-1] = 0
        out = oracle_lib.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, fixed=fixed,
                                          source=("sinprod", 2.0, [1.1, 0.7, 0.9]), diff=1.3)
        vals = torch.tensor(out["crs_vals"])
        res = torch.tensor(out["res"])
        ex = SlabExchange(out["rowptr"], out["colind"], P, nrows, rank, world, torch.device("cpu"))
        ex.export_add(res, vals)
        q.put((rank, out["rowptr"], out["colind"], vals.numpy(), res.numpy(),
               ex.remote_vals.numpy() if ex.has_upper else None,
               ex.remote_idx.numpy() if ex.has_upper else None, off, ex.bytes_on_wire()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("order,nxy,nz_per", [(1, 3, 2), (2, 2, 2)])
def test_two_slabs_equal_single_domain(oracle, order, nxy, nz_per):
    dim, qdeg, world = 3, 2 * order, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, dim, order, qdeg, nxy, nz_per, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=120)
        got[item[0]] = item
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    # single-domain reference
    mg = oracle.mesh_structured(dim, order, (nxy, nxy, nz_per * world), [0, 0, 0], [1, 1, world])
    D = order * nxy + 1
    P = D * D
    rng = np.random.default_rng(99)
    u_glob = rng.uniform(-1, 1, mg["ndof"])
    ref = oracle.assemble_thermal(dim, order, qdeg, mg["nodes"], mg["lids"], mg["offsets"], u_glob, fixed=mg["boundary"],
                                  source=("sinprod", 2.0, [1.1, 0.7, 0.9]), diff=1.3)
    Jg = sp.csr_matrix((ref["crs_vals"], ref["colind"], ref["rowptr"]), shape=(mg["ndof"],) * 2).toarray()
    scale = np.abs(Jg).max()

    for rank in range(world):
        _, rowptr, colind, vals, res, remote_vals, remote_idx, off, wire = got[rank]
        nrows = len(rowptr) - 1
        Jl = sp.csr_matrix((vals, colind, rowptr), shape=(nrows, nrows)).toarray()
        owned = np.arange(nrows) if rank == 0 else np.arange(P, nrows)  # bottom plane of rank>0 is ghost
        for r in owned:
            row_g = Jg[r + off]
            expect_local = row_g[off:off + nrows]
            assert np.abs(Jl[r] - expect_local).max() <= 1e-12 * scale, (rank, r)
            assert abs(res[r] - ref["res"][r + off]) <= 1e-12 * np.abs(ref["res"]).max()
        if rank < world - 1:
            # off-rank columns of the owned top-plane rows: entries of the upper slab's bottom rows whose
            # columns are NOT on the shared plane, in that slab's (row, col) order
            up_rowptr, up_colind = got[rank + 1][1], got[rank + 1][2]
            rows = np.repeat(np.arange(P), np.diff(up_rowptr[:P + 1]))
            cols = up_colind[:up_rowptr[P]]
            off_up = got[rank + 1][7]
            k = 0
            for i in np.flatnonzero(cols >= P):
                assert remote_idx[k] == i
                g = Jg[rows[i] + off_up, cols[i] + off_up]
                assert abs(remote_vals[k] - g) <= 1e-12 * scale
                k += 1
            assert k == len(remote_vals)
            assert wire == (up_rowptr[P] + P) * 8


def _worker_ns(rank, world, port, nxy, nz_per, q):
    """Same exchange for a multi-variable block (navierstokes Q2/Q1: ux, pr, uy, uz interleaved per node): the shared
    plane is still the first / last P rows of a slab, P = dofs of all variables on the plane."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import oracle_lib as orc
    from mrhyde_amd.shared_rows import SlabExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H = orc.HGRAD
        m = orc.mesh_multi(3, (nxy, nxy, nz_per), [H] * 4, [2, 1, 2, 2], lo=[0.0, 0.0, float(rank)],
                           hi=[1.0, 1.0, float(rank + 1)])
        nrows = m["ndof"]
        P = int(((m["side_mask"] >> 4) & 1).sum())  # dofs on the z- plane
        assert np.all(((m["side_mask"][:P] >> 4) & 1) == 1) and np.all(((m["side_mask"][-P:] >> 5) & 1) == 1)
        rng = np.random.default_rng(77)
        u_glob = rng.uniform(-1, 1, P + world * (nrows - P))
        off = rank * (nrows - P)
        u = u_glob[off:off + nrows]
        funcs = {"source ux": 0.3, "viscosity": 0.05, "density": 1.3}
        out = orc.assemble_block(m, orc.PHYS_NAVIERSTOKES, 4, u, funcs=funcs, params=[0, 0, 1])
        vals, res = torch.tensor(out["crs_vals"]), torch.tensor(out["res"])
        ex = SlabExchange(out["rowptr"], out["colind"], P, nrows, rank, world, torch.device("cpu"))
        ex.export_add(res, vals)
        q.put((rank, out["rowptr"], out["colind"], vals.numpy(), res.numpy(), off, P))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_slabs_navierstokes_block(oracle):
    nxy, nz_per, world = 2, 1, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ns, args=(r, world, port, nxy, nz_per, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=180)
        got[item[0]] = item
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    H = oracle.HGRAD
    mg = oracle.mesh_multi(3, (nxy, nxy, nz_per * world), [H] * 4, [2, 1, 2, 2], hi=[1.0, 1.0, float(world)])
    rng = np.random.default_rng(77)
    u_glob = rng.uniform(-1, 1, mg["ndof"])
    ref = oracle.assemble_block(mg, oracle.PHYS_NAVIERSTOKES, 4, u_glob,
                                funcs={"source ux": 0.3, "viscosity": 0.05, "density": 1.3}, params=[0, 0, 1])
    Jg = sp.csr_matrix((ref["crs_vals"], ref["colind"], ref["rowptr"]), shape=(mg["ndof"],) * 2).toarray()
    scale = np.abs(Jg).max()
    for rank in range(world):
        _, rowptr, colind, vals, res, off, P = got[rank]
        nrows = len(rowptr) - 1
        Jl = sp.csr_matrix((vals, colind, rowptr), shape=(nrows, nrows)).toarray()
        owned = np.arange(nrows) if rank == 0 else np.arange(P, nrows)
        for r in owned:
            assert np.abs(Jl[r] - Jg[r + off][off:off + nrows]).max() <= 1e-12 * scale, (rank, r)
            assert abs(res[r] - ref["res"][r + off]) <= 1e-12 * np.abs(ref["res"]).max()
