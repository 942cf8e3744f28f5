"""Export(ADD) of shared-DOF rows between z-slab shards (mrhyde_amd/shared_rows.py), world_size 2, gloo on CPU.

Mirrors the reference's rank-count independence check (regression/thermal/2D_verification_mpi): the
2-shard result, after the shared-row exchange, equals the single-domain assembly row by row.
The per-shard assembly itself comes from the CPU oracle here (no GPU in this test); the lists the exchange
walks (who sends which entries, where every received entry is added) are the ones the library's pack /
unpack kernels get on a GPU (tests/test_shared_rows_gpu.py runs the same comparison with GPU assembly and
the C-ABI kernels, two processes on one card).
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
        rv, ri = (None, None)
        if rank < world - 1:
            rv, ri = ex.remote_vals(rank + 1)
            rv = rv.numpy().copy()
        # bytes a rank sends: its bottom-plane rows (values + residual) when it has a lower neighbour
        q.put((rank, out["rowptr"], out["colind"], vals.numpy(), res.numpy(), rv, ri, off, ex.bytes_on_wire()))
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
            assert got[rank + 1][8] == (up_rowptr[P] + P) * 8


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


def _worker_porous(rank, world, port, nxy, nz, q):
    """porousMixed (HVOL p + HDIV u): the shared dofs are the z-faces between the slabs -- not a leading / trailing run
    of rows, so this is the explicit-list case; slabs of different thickness (nz[rank] layers)."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import oracle_lib as orc
    from mrhyde_amd.shared_rows import SharedRowExport
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        z0 = float(sum(nz[:rank]))
        m = orc.mesh_multi(3, (nxy, nxy, nz[rank]), [orc.HVOL, orc.HDIV], [0, 1], lo=[0.0, 0.0, z0], hi=[1.0, 1.0, z0 + nz[rank]])
        gid = _porous_gids(nxy, nz, rank)
        rng = np.random.default_rng(55)
        u_glob = rng.uniform(-1, 1, _porous_ndof(nxy, sum(nz)))
        u = u_glob[gid]
        out = orc.assemble_block(m, orc.PHYS_POROUS_MIXED, 2, u, funcs={"source": 0.7, "Kinv_xx": 1.3})
        vals, res = torch.tensor(out["crs_vals"]), torch.tensor(out["res"])
        ex = SharedRowExport(gid, out["rowptr"], out["colind"], rank, world, torch.device("cpu"))
        ex.export_add(res, vals)
        q.put((rank, gid, out["rowptr"], out["colind"], vals.numpy(), res.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _porous_ndof(nxy, nz):
    return nxy * nxy * nz + (nxy + 1) * nxy * nz * 2 + nxy * nxy * (nz + 1)


def _porous_gids(nxy, nz, rank):
    """Global ids (numbering of the single-domain mesh_multi: cells, x-faces, y-faces, z-faces) of a slab's dofs."""
    nzt, k0, nzl = sum(nz), sum(nz[:rank]), nz[rank]
    ne_g = nxy * nxy * nzt
    cells = np.arange(nxy * nxy * nzl) + nxy * nxy * k0
    fx = ne_g + np.arange((nxy + 1) * nxy * nzl) + (nxy + 1) * nxy * k0
    fy = ne_g + (nxy + 1) * nxy * nzt + np.arange(nxy * (nxy + 1) * nzl) + nxy * (nxy + 1) * k0
    fz = ne_g + 2 * (nxy + 1) * nxy * nzt + np.arange(nxy * nxy * (nzl + 1)) + nxy * nxy * k0
    return np.concatenate([cells, fx, fy, fz]).astype(np.int64)


def test_two_unequal_slabs_porous_mixed_explicit_lists(oracle):
    nxy, nz, world = 3, (2, 1), 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_porous, args=(r, world, port, nxy, nz, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=180)
        got[item[0]] = item
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mg = oracle.mesh_multi(3, (nxy, nxy, sum(nz)), [oracle.HVOL, oracle.HDIV], [0, 1], hi=[1.0, 1.0, float(sum(nz))])
    assert mg["ndof"] == _porous_ndof(nxy, sum(nz))
    u_glob = np.random.default_rng(55).uniform(-1, 1, mg["ndof"])
    ref = oracle.assemble_block(mg, oracle.PHYS_POROUS_MIXED, 2, u_glob, funcs={"source": 0.7, "Kinv_xx": 1.3})
    Jg = sp.csr_matrix((ref["crs_vals"], ref["colind"], ref["rowptr"]), shape=(mg["ndof"],) * 2).toarray()
    scale = np.abs(Jg).max()
    seen = np.zeros(mg["ndof"], bool)
    for rank in range(world):
        _, gid, rowptr, colind, vals, res = got[rank]
        n = len(gid)
        Jl = sp.csr_matrix((vals, colind, rowptr), shape=(n, n)).toarray()
        owned = ~np.isin(gid, got[0][1]) if rank > 0 else np.ones(n, bool)  # the lower slab owns the shared faces
        for r in np.flatnonzero(owned):
            assert np.abs(Jl[r] - Jg[gid[r]][gid]).max() <= 1e-12 * scale, (rank, r)
            assert abs(res[r] - ref["res"][gid[r]]) <= 1e-12 * np.abs(ref["res"]).max()
            seen[gid[r]] = True
    assert seen.all()


# ---- HDG trace rows shared between strips (config 5: the condensed trace system of SubGridDtN_Solver::updateFlux,
# ---- subgridDtN_solver.cpp:1542-1616, exported with ADD as linearAlgebraInterface.hpp:296-337 does) ----

def _trace_scatter(lids, nrows, S, g):
    """Plain COO accumulation of condensed element blocks into the trace system (what mha_scatter_plan_apply computes)."""
    n = lids.shape[1]
    rows = np.repeat(lids, n, axis=1).ravel()
    cols = np.tile(lids, (1, n)).ravel()
    J = sp.coo_matrix((S.ravel(), (rows, cols)), shape=(nrows, nrows)).tocsr()
    J.sort_indices()
    r = np.zeros(nrows)
    np.add.at(r, lids.ravel(), g.ravel())
    return J, r


def _worker_hdg(rank, world, port, ncx, ncy, residual_only, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    from mrhyde_amd.shared_rows import SharedRowExport, hdg_strip_gids, hdg_trace_lids
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lids, nrows = hdg_trace_lids(ncx, ncy[rank])
        gid = hdg_strip_gids(ncx, ncy, rank)
        E0 = ncx * sum(ncy[:rank])
        rng = np.random.default_rng(66)
        Sg = rng.uniform(-1, 1, (ncx * sum(ncy), 24, 24))  # one block per element of the WHOLE mesh: a strip takes its own
        gg = rng.uniform(-1, 1, (ncx * sum(ncy), 24))
        S, g = Sg[E0:E0 + lids.shape[0]], gg[E0:E0 + lids.shape[0]]
        J, r = _trace_scatter(lids, nrows, S, g)
        vals, res = torch.tensor(J.data.copy()), torch.tensor(r)
        ex = SharedRowExport(gid, J.indptr, J.indices, rank, world, torch.device("cpu"))
        ex.export_add(res, None if residual_only else vals)
        q.put((rank, gid, J.indptr.copy(), J.indices.copy(), vals.numpy(), res.numpy(), J.data.copy(), ex.bytes_on_wire()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("residual_only", [False, True])
def test_two_strips_hdg_trace_rows(residual_only):
    """Strips of 3 and 2 element rows: the trace rows of the horizontal edges between them are summed into the owner
    (the lower strip); every owned row equals the single-domain trace system.  residual_only: the exchange of
    assembleRes (no value array) moves and adds the residual halves only."""
    from mrhyde_amd.shared_rows import hdg_trace_lids
    ncx, ncy, world = 4, (3, 2), 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_hdg, args=(r, world, port, ncx, ncy, residual_only, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=180)
        got[item[0]] = item
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lg, ng = hdg_trace_lids(ncx, sum(ncy))
    rng = np.random.default_rng(66)
    Sg = rng.uniform(-1, 1, (ncx * sum(ncy), 24, 24))
    gg = rng.uniform(-1, 1, (ncx * sum(ncy), 24))
    Jg, rg = _trace_scatter(lg, ng, Sg, gg)
    Jg = Jg.toarray()
    seen = np.zeros(ng, bool)
    shared = np.intersect1d(got[0][1], got[1][1])
    assert len(shared) == ncx * 6  # one row of horizontal edges, 3 variables x 2 dofs each
    for rank in range(world):
        _, gid, rowptr, colind, vals, res, vals0, wire = got[rank]
        n = len(gid)
        Jl = sp.csr_matrix((vals, colind, rowptr), shape=(n, n)).toarray()
        owned = ~np.isin(gid, got[0][1]) if rank > 0 else np.ones(n, bool)
        for r in np.flatnonzero(owned):
            assert abs(res[r] - rg[gid[r]]) <= 1e-12 * np.abs(rg).max()
            if not residual_only:
                assert np.abs(Jl[r] - Jg[gid[r]][gid]).max() <= 1e-12 * np.abs(Jg).max(), (rank, r)
            seen[gid[r]] = True
        if residual_only:
            assert np.array_equal(vals, vals0)  # the value array is not touched
        assert wire == (0 if rank == 0 else 8 * (len(shared) + sum(rowptr[r + 1] - rowptr[r] for r in np.flatnonzero(np.isin(gid, shared)))))
    assert seen.all()
