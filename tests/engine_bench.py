"""Timing of the multi-variable paths at the sizes of BASELINE.json configs 2-5 (not a bench.py line: those configs are
parity cases; this script records where the implementation stands).  It lives under tests/ because it builds its
inputs with the oracle's mesh generator and can time the oracle as a CPU baseline -- test infrastructure, like the
parity tests.  Usage:
  python tests/engine_bench.py [thermal|porous|ns|hdg] [ncell] [full|gather|res|local] [cpu]"""
import sys
import time

import numpy as np
import torch

import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # oracle_lib
import mrhyde_amd  # noqa: E402
import oracle_lib as orc  # noqa: E402  (mesh generator only: test infrastructure building the input)


def run(kind, nc):
    dim = 3
    if kind == "thermal":
        types, orders, phys, qdeg, B = [orc.HGRAD], [2], "thermal", 4, 6564
    elif kind == "porous":
        types, orders, phys, qdeg, B = [orc.HVOL, orc.HDIV], [0, 1], "porousMixed", 2, 724
    else:
        types, orders, phys, qdeg, B = [orc.HGRAD] * 4, [2, 1, 2, 2], "navierstokes", 4, 65340
    t0 = time.time()
    m = orc.mesh_multi(dim, (nc,) * 3, types, orders)
    blk = mrhyde_amd.Block(dim, quadrature=qdeg, physics=phys, variables=list(zip(types, orders)))
    blk.set_mesh(m["nodes"], m["lids"], m["offsets"], m["ndof"])
    blk.set_orientation(m["orient"])
    blk.set_graph()
    rowptr, colind = blk.get_graph()
    print("setup %.1fs  elements %d  dofs %d  nnz %d" % (time.time() - t0, m["nelem"], m["ndof"], len(colind)), flush=True)
    rng = np.random.default_rng(5)
    u = torch.tensor(rng.uniform(-1, 1, m["ndof"]), device="cuda")
    res = torch.zeros(m["ndof"], dtype=torch.float64, device="cuda")
    vals = torch.zeros(len(colind), dtype=torch.float64, device="cuda")
    if kind == "ns":
        blk.set_function("viscosity", 1.0)
    blk.set_timing(True)
    mode = sys.argv[3] if len(sys.argv) > 3 else "full"
    if mode == "local":
        E, n = m["lids"].shape
        lJ = torch.zeros((E, n, n), dtype=torch.float64, device="cuda")
        lr = torch.zeros((E, n), dtype=torch.float64, device="cuda")
    for it in range(3):
        if mode == "full":
            blk.assemble_jacres(u, res, vals, overwrite=True, path=mrhyde_amd.PATH_POINT_ENGINE)
        elif mode == "gather":
            blk.assemble_jacres(u, res, vals, overwrite=True, path=mrhyde_amd.PATH_ROW_GATHER)
        elif mode == "res":
            blk.assemble_jacres(u, res, None, compute_jacobian=False, overwrite=True, path=mrhyde_amd.PATH_POINT_ENGINE)
        else:
            blk.compute_local_jacres(u, lJ, lr)
        torch.cuda.synchronize()
        ms = blk.last_kernel_ms()
        print(mode, "%s %d^3: %.3f ms  %.3e elements/s  %.1f GB/s algorithmic (%.1f%% of 8 TB/s)" %
              (kind, nc, ms, m["nelem"] / ms * 1e3, m["nelem"] * B / ms / 1e6, m["nelem"] * B / ms / 1e6 / 80.0), flush=True)


def run_hdg(nc):
    """Config 5 shape: shallowwaterHybridized on nc^2 quads, Q1 interior + HFACE-1 traces: volume element matrices
    (point engine, dense) + the HDG element blocks of all four sides (11 056 algorithmic bytes per element)."""
    H = orc.HGRAD
    m = orc.mesh_multi(2, (nc, nc), [H, H, H], [1, 1, 1])
    blk = mrhyde_amd.Block(2, quadrature=2, physics="shallowwaterHybridized", variables=[(0, 1)] * 3)
    blk.set_mesh(m["nodes"], m["lids"], m["offsets"], m["ndof"])
    blk.set_graph()
    rng = np.random.default_rng(6)
    E = m["nelem"]
    r2 = ((m["verts"] - 0.5) ** 2).sum(axis=1)
    u = 0.01 * rng.uniform(-1, 1, m["ndof"])
    hdofs = m["dof_var"] == 0
    u[hdofs] = 1.0 + 0.2 * rng.uniform(0, 1, hdofs.sum())
    lam = 0.01 * rng.uniform(-1, 1, (E, 3, 4, 2))
    lam[:, 0] = 1.0 + 0.2 * rng.uniform(0, 1, (E, 4, 2))
    ud, ld = torch.tensor(u, device="cuda"), torch.tensor(lam.reshape(E, 24), device="cuda")
    res = torch.zeros((E, 36), dtype=torch.float64, device="cuda")
    blocks = torch.zeros((E, 36, 36), dtype=torch.float64, device="cuda")
    lJ = torch.zeros((E, 12, 12), dtype=torch.float64, device="cuda")
    lr = torch.zeros((E, 12), dtype=torch.float64, device="cuda")
    blk.set_timing(True)
    B = 11056
    for it in range(3):
        blk.swhdg_element_blocks(ud, ld, res, blocks)
        torch.cuda.synchronize()
        t_side = blk.last_kernel_ms()
        blk.compute_local_jacres(ud, lJ, lr)
        torch.cuda.synchronize()
        t_vol = blk.last_kernel_ms()
        ms = t_side + t_vol
        print("hdg %d^2: sides %.3f ms + volume %.3f ms = %.3f ms  %.3e elements/s  %.1f GB/s algorithmic (%.1f%% of 8 TB/s)" %
              (nc, t_side, t_vol, ms, E / ms * 1e3, E * B / ms / 1e6, E * B / ms / 1e6 / 80.0), flush=True)
    # static condensation of the element blocks (interior block made non-singular by a mass-like shift, as a transient run has)
    off = torch.tensor(m["offsets"], device="cuda", dtype=torch.long)
    blocks[:, :12, :12] += lJ[:, off][:, :, off] + 10.0 * torch.eye(12, dtype=torch.float64, device="cuda")
    res[:, :12] += lr[:, off]
    for it in range(3):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        S, gv, du, ns = mrhyde_amd.batched_condense(12, 24, blocks, res)
        t1.record()
        torch.cuda.synchronize()
        print("hdg %d^2: condensation %.3f ms (incl. output allocation), singular %d" % (nc, t0.elapsed_time(t1), ns), flush=True)
    # flux -> trace scatter of the condensed blocks into the macro trace system (HFACE edge numbering of the nc x nc mesh)
    nvert = (nc + 1) * nc
    ii, jj = np.meshgrid(np.arange(nc), np.arange(nc), indexing="xy")
    ii, jj = ii.ravel(), jj.ravel()
    edges = np.stack([jj * (nc + 1) + ii, nvert + jj * nc + ii, jj * (nc + 1) + ii + 1, nvert + (jj + 1) * nc + ii], axis=1)
    lids = np.zeros((E, 24), np.int32)
    for v in range(3):
        for k in range(4):
            for f in range(2):
                lids[:, (v * 4 + k) * 2 + f] = (edges[:, k] * 3 + v) * 2 + f
    nrows_t = (nvert + nc * (nc + 1)) * 6
    plan = mrhyde_amd.ScatterPlan(lids, nrows_t)
    tv = torch.zeros(plan.nnz, dtype=torch.float64, device="cuda")
    tr_ = torch.zeros(nrows_t, dtype=torch.float64, device="cuda")
    for it in range(3):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        plan.apply(S, gv, tr_, tv, overwrite=True, stream=torch.cuda.current_stream().cuda_stream)
        t1.record()
        torch.cuda.synchronize()
        print("hdg %d^2: flux->trace scatter %.3f ms (%d trace rows, %d CRS entries)" % (nc, t0.elapsed_time(t1), nrows_t, plan.nnz), flush=True)


def cpu_baseline(kind, sample_nc):
    """The oracle's AD-array restatement (1 thread) on a bounded sample of the same workload."""
    dim = 3
    if kind == "thermal":
        types, orders, phys, qdeg = [orc.HGRAD], [2], orc.PHYS_THERMAL, 4
    elif kind == "porous":
        types, orders, phys, qdeg = [orc.HVOL, orc.HDIV], [0, 1], orc.PHYS_POROUS_MIXED, 2
    else:
        types, orders, phys, qdeg = [orc.HGRAD] * 4, [2, 1, 2, 2], orc.PHYS_NAVIERSTOKES, 4
    m = orc.mesh_multi(dim, (sample_nc,) * 3, types, orders)
    rowptr, colind = orc.build_graph(m["ndof"], m["lids"])
    u = np.random.default_rng(5).uniform(-1, 1, m["ndof"])
    t0 = time.time()
    orc.assemble_block(m, phys, qdeg, u, rowptr=rowptr, colind=colind)
    dt = time.time() - t0
    return {"value": m["nelem"] / dt, "unit": "elements/s", "cores": 1, "kind": "port", "seconds": dt,
            "sample": "%d^3 elements of the same block, oracle AD-array restatement, 1 thread" % sample_nc}


if __name__ == "__main__":
    if len(sys.argv) > 4 and sys.argv[4] == "cpu":
        import json
        print(json.dumps({"workload": sys.argv[1], "cpu_baseline": cpu_baseline(sys.argv[1], int(sys.argv[2]))}))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "hdg":
        run_hdg(int(sys.argv[2]) if len(sys.argv) > 2 else 256)
        sys.exit(0)
    run(sys.argv[1] if len(sys.argv) > 1 else "porous", int(sys.argv[2]) if len(sys.argv) > 2 else 32)
