"""Export(ADD) of shared-DOF rows with the GPU path end to end: two processes on one MI355X, each assembling its z-slab
with the library (row-owner kernels) and exchanging the shared plane through mha_export_pack / mha_export_unpack_add
(the C-ABI kernels); gloo moves the packed buffers (two ranks cannot share one card over RCCL).  The result equals the
single-domain oracle assembly row by row -- the reference's rank-count independence check
(regression/thermal/2D_verification_mpi) on the device path."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, order, nxy, nz_per, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    import torch
    import torch.distributed as dist
    import mrhyde_amd
    from mrhyde_amd.shared_rows import SlabExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dim, qdeg = 3, 2 * order
        dev = torch.device("cuda", 0)
        m = mrhyde_amd.mesh_structured(dim, order, (nxy, nxy, nz_per), [0.0, 0.0, float(rank)], [1.0, 1.0, float(rank + 1)])
        D = order * nxy + 1
        P, nrows = D * D, m["ndof"]
        u_glob = np.random.default_rng(99).uniform(-1, 1, P * (order * nz_per * world + 1))
        off = rank * (nrows - P)
        fixed = m["boundary"].copy()  # physical boundary of the stacked domain only
        if rank > 0:
            fixed[:P].reshape(D, D)[1:-1, 1:-1] = 0
        if rank < world - 1:
            fixed[-P:].reshape(D, D)[1:-1, 1:-1] = 0
        blk = mrhyde_amd.Block(dim, order, quadrature=qdeg)
        blk.set_mesh(m["nodes"], m["lids"], m["offsets"], nrows, fixed)
        blk.set_graph()
        blk.set_function("thermal source", ("sinprod", 2.0, [1.1, 0.7, 0.9]))
        blk.set_function("thermal diffusion", 1.3)
        rowptr, colind = blk.get_graph()
        u = torch.tensor(u_glob[off:off + nrows], device=dev)
        res = torch.zeros(nrows, dtype=torch.float64, device=dev)
        vals = torch.zeros(len(colind), dtype=torch.float64, device=dev)
        blk.assemble_jacres(u, res, vals, overwrite=True)
        assert blk.info("last_path") == mrhyde_amd.PATH_ROW_OWNER
        ex = SlabExchange(rowptr, colind, P, nrows, rank, world, dev)
        assert ex._plan is not None  # the library's plan, not the CPU stand-in
        ex.export_add(res, vals)
        torch.cuda.synchronize()
        q.put((rank, rowptr, colind, vals.cpu().numpy(), res.cpu().numpy(), off))
        dist.barrier()
        ex.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("order,nxy,nz_per", [(2, 4, 2)])
def test_two_slabs_on_one_gpu_equal_single_domain(oracle, order, nxy, nz_per):
    import torch.multiprocessing as mp
    dim, qdeg, world = 3, 2 * order, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, order, nxy, nz_per, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=300)
        got[item[0]] = item
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    mg = oracle.mesh_structured(dim, order, (nxy, nxy, nz_per * world), [0, 0, 0], [1, 1, world])
    D = order * nxy + 1
    P = D * D
    u_glob = np.random.default_rng(99).uniform(-1, 1, mg["ndof"])
    ref = oracle.assemble_thermal(dim, order, qdeg, mg["nodes"], mg["lids"], mg["offsets"], u_glob, fixed=mg["boundary"],
                                  source=("sinprod", 2.0, [1.1, 0.7, 0.9]), diff=1.3)
    Jg = sp.csr_matrix((ref["crs_vals"], ref["colind"], ref["rowptr"]), shape=(mg["ndof"],) * 2).toarray()
    scale = np.abs(Jg).max()
    for rank in range(world):
        _, rowptr, colind, vals, res, off = got[rank]
        nrows = len(rowptr) - 1
        Jl = sp.csr_matrix((vals, colind, rowptr), shape=(nrows, nrows)).toarray()
        owned = np.arange(nrows) if rank == 0 else np.arange(P, nrows)  # bottom plane of rank > 0 is ghost
        for r in owned:
            assert np.abs(Jl[r] - Jg[r + off][off:off + nrows]).max() <= 1e-12 * scale, (rank, r)
            assert abs(res[r] - ref["res"][r + off]) <= 1e-12 * np.abs(ref["res"]).max()


# ---- HDG trace rows (config 5): per-strip flux -> trace scatter on the device (mha_scatter_plan_apply), Export(ADD) of
# ---- the trace rows between the strips through the library's pack / unpack kernels, against the single-domain scatter ----

def _worker_hdg(rank, world, port, ncx, ncy, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    import torch
    import torch.distributed as dist
    import mrhyde_amd
    from mrhyde_amd.shared_rows import SharedRowExport, hdg_strip_gids, hdg_trace_lids
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        lids, nrows = hdg_trace_lids(ncx, ncy[rank])
        gid = hdg_strip_gids(ncx, ncy, rank)
        E0 = ncx * sum(ncy[:rank])
        rng = np.random.default_rng(66)
        Sg = rng.uniform(-1, 1, (ncx * sum(ncy), 24, 24))
        gg = rng.uniform(-1, 1, (ncx * sum(ncy), 24))
        S = torch.tensor(Sg[E0:E0 + lids.shape[0]], device=dev)
        g = torch.tensor(gg[E0:E0 + lids.shape[0]], device=dev)
        plan = mrhyde_amd.ScatterPlan(lids, nrows)
        rowptr, colind = plan.graph()
        vals = torch.zeros(plan.nnz, dtype=torch.float64, device=dev)
        res = torch.zeros(nrows, dtype=torch.float64, device=dev)
        plan.apply(S, g, res, vals, overwrite=True, stream=torch.cuda.current_stream().cuda_stream)
        ex = SharedRowExport(gid, rowptr, colind, rank, world, dev)
        assert ex._plan is not None
        ex.export_add(res, vals)
        res2 = res.clone()
        ex.export_add(res2, None)  # residual-only exchange on the device path: the value array stays as it is
        torch.cuda.synchronize()
        q.put((rank, gid, rowptr, colind, vals.cpu().numpy(), res.cpu().numpy()))
        dist.barrier()
        ex.close()
    finally:
        dist.destroy_process_group()


def test_two_strips_hdg_trace_rows_on_one_gpu():
    import torch.multiprocessing as mp
    from mrhyde_amd.shared_rows import hdg_trace_lids
    ncx, ncy, world = 6, (4, 3), 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_hdg, args=(r, world, port, ncx, ncy, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=300)
        got[item[0]] = item
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    lg, ng = hdg_trace_lids(ncx, sum(ncy))
    rng = np.random.default_rng(66)
    Sg = rng.uniform(-1, 1, (ncx * sum(ncy), 24, 24))
    gg = rng.uniform(-1, 1, (ncx * sum(ncy), 24))
    n = lg.shape[1]
    Jg = sp.coo_matrix((Sg.ravel(), (np.repeat(lg, n, axis=1).ravel(), np.tile(lg, (1, n)).ravel())), shape=(ng, ng)).toarray()
    rg = np.zeros(ng)
    np.add.at(rg, lg.ravel(), gg.ravel())
    seen = np.zeros(ng, bool)
    for rank in range(world):
        _, gid, rowptr, colind, vals, res = got[rank]
        nl = len(gid)
        Jl = sp.csr_matrix((vals, colind, rowptr), shape=(nl, nl)).toarray()
        owned = ~np.isin(gid, got[0][1]) if rank > 0 else np.ones(nl, bool)
        for r in np.flatnonzero(owned):
            assert np.abs(Jl[r] - Jg[gid[r]][gid]).max() <= 1e-12 * np.abs(Jg).max(), (rank, r)
            assert abs(res[r] - rg[gid[r]]) <= 1e-12 * np.abs(rg).max()
            seen[gid[r]] = True
    assert seen.all()
