#!/usr/bin/env python3
"""bench.py -- assembled elements/s of the element-local assembly hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 is launched by torch.distributed.run,
one rank per GPU).  One "step" = one full volume assembly of the block: zero the residual and the CRS
values (what the Newton loop does before assembling, solverManager.cpp:1528-1533), then gather ->
residual+Jacobian -> scatter for every element (AssemblyManager::assembleJacRes, assemblyManager.cpp:2150-2665),
then -- for N>1 -- the Export(ADD) of the shared-DOF rows between neighbouring slabs.
Inputs are resident in HBM before the timed region.

Workload: `--config 2` (default, the configuration BASELINE.json's metric is quoted on): 3-D thermal, Q2 hex, 64^3
structured mesh, quadrature 4.  `--config 3|4|5` select the other BASELINE.json configurations with the same JSON
shape (SURVEY.md section 8(d) inputs and bytes per element):
  3  porousMixed (HVOL + HDIV) on 128^3 hexes            724 B/element
  4  navierstokes Q2/Q1 hexes, 64^3                     65 340 B/element
  5  shallowwaterHybridized HDG on 256^2 quads: side blocks + volume + static condensation + flux->trace scatter
                                                        11 056 B/element
N>1: one block of the configuration's size per GPU (weak scaling), stacked in z so that neighbouring ranks share
one dof plane (configs 2 and 4), the HDIV z-face dofs (config 3) or -- strips stacked in y -- the HFACE trace rows of
the horizontal edges between two strips (config 5): explicit shared-row lists from the global ids (SharedRowExport).
`python bench.py --gpus N` without a launcher spawns its own N ranks (torch.distributed.run on 127.0.0.1).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E datasheet peak (/opt/skills/guides/MI355X_MICROARCH.md)
HGRAD, HVOL, HDIV = 0, 1, 2


def algorithmic_bytes_per_elem(nnodes, dim, n):
    """coords + LIDs + gathered u + element Jacobian out + element residual out."""
    return 8 * nnodes * dim + 4 * n + 8 * n + 8 * n * n + 8 * n


def dof_coords(dim, order, ncell, lo, hi):
    D = [order * c + 1 for c in ncell]
    ax = [np.linspace(lo[d], hi[d], D[d]) for d in range(dim)]
    if dim == 3:
        z, y, x = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
        return np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1)
    y, x = np.meshgrid(ax[1], ax[0], indexing="ij")
    return np.stack([x.ravel(), y.ravel()], axis=1)


def synthetic_state(dim, order, ncell, lo, hi, seed):
    """u_j = prod sin(2 pi x) + 0.01 U(-1,1)   (seed 2)"""
    xyz = dof_coords(dim, order, ncell, lo, hi)
    rng = np.random.default_rng(seed)
    return np.prod(np.sin(2 * np.pi * xyz), axis=1) + 0.01 * rng.uniform(-1, 1, xyz.shape[0])


def host_threads():
    """Threads for the CPU baseline: the box's CPU share (affinity mask, cgroup quota, at most 16)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("MHA_CPU_THREADS", "16"))))


def log(msg):
    print("[bench %7.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def measured_traffic(config, mesh, path_name):
    """HBM bytes per assembly from the committed PMC run of this exact workload (profiles/r3_traffic.json: FETCH_SIZE
    + WRITE_SIZE collected in separate --pmc passes, corrected as MI355X_MICROARCH.md prescribes), or None (other sizes,
    other paths: nothing measured)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r3_traffic.json")))
        e = t.get("config%d_%s" % (config, mesh))
        if e and e.get("path") == path_name:
            return e["hbm_bytes_per_assembly"]
    except Exception:
        pass
    return None


def measured_copy_gbs(torch, dev):
    """Device-to-device copy bandwidth on this box right now (read + write of a 1 GiB buffer), GB/s."""
    n = 1 << 27
    a = torch.empty(n, dtype=torch.float64, device=dev)
    b = torch.zeros(n, dtype=torch.float64, device=dev)
    for _ in range(2):
        a.copy_(b)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(5):
        a.copy_(b)
    t1.record()
    torch.cuda.synchronize()
    return 5 * 2 * 8 * n / (t0.elapsed_time(t1) * 1e-3) / 1e9


# ---------------------------------------------------------------------------------------------------------------------
# CPU baselines: the oracle ("port" of the reference data flow), bounded samples of the same workload
# ---------------------------------------------------------------------------------------------------------------------

def _median_after_warmups(run, warmups=2, reps=10):
    """median of >= 10 repetitions after 2 warm-ups."""
    for _ in range(warmups):
        run()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = run()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), out, reps


def cpu_baseline_thermal(dim, order, qdeg, ncell_sample, threads, gpu_vals=None, budget_s=30.0):
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    freq = [2 * np.pi] * dim

    def prepare(nc):
        m = oracle_lib.mesh_structured(dim, order, nc)
        u = synthetic_state(dim, order, nc, [0] * dim, [1] * dim, 2)
        pb = oracle_lib.physical_basis(dim, order, qdeg, m["nodes"])  # stored basis: setup, not timed (as in the reference)
        rowptr, colind = oracle_lib.build_graph(m["ndof"], m["lids"])

        def run(nt):
            return oracle_lib.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, fixed=m["boundary"],
                                               pb=pb, workset_size=100, source=("sinprod", 4.0 * dim * np.pi ** 2, freq),
                                               num_threads=nt, rowptr=rowptr, colind=colind)
        return m, run
    nc = tuple(ncell_sample)
    m, run = prepare(nc)
    t0 = time.perf_counter()
    ref = run(threads)  # probe: sizes the sample so that 2 warm-ups + 10 repetitions fit the budget
    t1 = time.perf_counter() - t0
    if 12 * t1 > budget_s and nc[-1] > 2:
        nz = max(2, int(nc[-1] * budget_s / (12 * t1)))
        nc = nc[:-1] + (nz,)
        m, run = prepare(nc)
    t, last, reps = _median_after_warmups(lambda: run(threads))
    out = {"value": m["nelem"] / t, "unit": "elements/s", "cores": threads, "kind": "port",
           "sample": "%s Q%d hex elements (%d; z-layers of the 64^3 mesh), workset 100, stored basis, AD width %d, median of %d after 2 warm-ups" % (
               "x".join(map(str, nc)), order, m["nelem"], oracle_lib.ad_width((order + 1) ** dim), reps),
           "seconds": t}
    if gpu_vals is not None and len(gpu_vals) == len(ref["crs_vals"]):
        out["gpu_max_rel_diff_jacobian"] = float(np.abs(gpu_vals - ref["crs_vals"]).max() / np.abs(ref["crs_vals"]).max())
    # one core, on a thinner sample of the same mesh (same x-y extent, 2 element layers)
    nc1 = tuple(ncell_sample[:-1]) + (min(2, ncell_sample[-1]),)
    m1, run1 = prepare(nc1)
    t1c, _, reps1 = _median_after_warmups(lambda: run1(1), warmups=1, reps=3)
    out["value_1core"] = m1["nelem"] / t1c
    out["sample_1core"] = "%s elements, 1 thread, median of %d after 1 warm-up" % ("x".join(map(str, nc1)), reps1)
    return out


def cpu_baseline_block(kind, sample_nc, state_fn, funcs, params):
    """porousMixed / navierstokes: the oracle's AD-array restatement (1 thread) on a bounded sample of the same block."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc
    if kind == "porous":
        types, orders, phys, qdeg = [orc.HVOL, orc.HDIV], [0, 1], orc.PHYS_POROUS_MIXED, 2
    else:
        types, orders, phys, qdeg = [orc.HGRAD] * 4, [2, 1, 2, 2], orc.PHYS_NAVIERSTOKES, 4
    m = orc.mesh_multi(3, (sample_nc,) * 3, types, orders)
    rowptr, colind = orc.build_graph(m["ndof"], m["lids"])
    u = state_fn(m)
    dt, _, reps = _median_after_warmups(lambda: orc.assemble_block(m, phys, qdeg, u, funcs=funcs, params=params, rowptr=rowptr, colind=colind))
    return {"value": m["nelem"] / dt, "unit": "elements/s", "cores": 1, "kind": "port", "seconds": dt,
            "sample": "%d^3 elements of the same block, oracle AD-array restatement, 1 thread, median of %d after 2 warm-ups" % (sample_nc, reps)}


def cpu_baseline_hdg(sample_nc, seed):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc
    H = orc.HGRAD
    m = orc.mesh_multi(2, (sample_nc, sample_nc), [H, H, H], [1, 1, 1])
    u, lam = hdg_state(m, seed)
    st = np.zeros((m["nelem"], 4), np.uint8)
    dt, _, reps = _median_after_warmups(lambda: orc.swh_hdg_element(m, 2, u, lam.reshape(m["nelem"], 24), st, [1.0, 0.0, 0.0], g=1.0, roe=False))
    return {"value": m["nelem"] / dt, "unit": "elements/s", "cores": 1, "kind": "port", "seconds": dt,
            "sample": "%d^2 HDG elements (side blocks only: the oracle restates the element, not the condensation), 1 thread, "
                      "median of %d after 2 warm-ups" % (sample_nc, reps)}


def hdg_state(m, seed):
    """cfg5: H = 1 + exp(-100 r^2), momenta 0 + 0.01 U(-1,1) (seed 6), traces likewise."""
    rng = np.random.default_rng(seed)
    E = m["nelem"]
    u = 0.01 * rng.uniform(-1, 1, m["ndof"])
    hd = m["dof_var"] == 0
    # H dofs of a Q1 variable sit on the vertices, in vertex order
    r2 = ((m["verts"] - 0.5) ** 2).sum(axis=1)
    u[hd] = 1.0 + np.exp(-100.0 * r2)
    lam = 0.01 * rng.uniform(-1, 1, (E, 3, 4, 2))
    cen = m["nodes"].mean(axis=1)
    lam[:, 0] = (1.0 + np.exp(-100.0 * ((cen - 0.5) ** 2).sum(axis=1)))[:, None, None]
    return u, lam


# ---------------------------------------------------------------------------------------------------------------------
# workloads: each returns dict(step, kernel_ms, E, b_elem, workload, config, cpu (callable or None), exch)
# ---------------------------------------------------------------------------------------------------------------------

def setup_thermal(args, torch, mrhyde_amd, rank, world, dev):
    from mrhyde_amd.shared_rows import SlabExchange
    dim, order, qdeg = 3, args.order, 2 * args.order
    ncell = (args.ncell or 64,) * 3
    nc = ncell[0]
    lo, hi = [0.0, 0.0, float(rank)], [1.0, 1.0, float(rank + 1)]
    m = mrhyde_amd.mesh_structured(dim, order, ncell, lo, hi)
    if args.mesh == "perturbed":  # interior vertices moved by 0.15 h U(-1,1)^3, seed 3
        rng = np.random.default_rng(3)
        v = m["verts"]
        h = 1.0 / nc
        interior = np.all((v - np.array(lo) > 1e-9) & (np.array(hi) - v > 1e-9), axis=1)
        v[interior] += 0.15 * h * rng.uniform(-1, 1, (int(interior.sum()), 3))
        m["nodes"] = np.ascontiguousarray(v[m["cell2vert"]])
    n, nn = m["lids"].shape[1], 2 ** dim
    E, nrows = m["nelem"], m["ndof"]
    blk = mrhyde_amd.Block(dim, order, quadrature=qdeg, workset_size=100, device=dev.index)
    blk.set_stream(torch.cuda.current_stream().cuda_stream)
    fixed = m["boundary"].copy()  # Dirichlet rows: the physical boundary of the stacked domain (not the inter-slab planes)
    if world > 1:
        P = (order * nc + 1) ** 2
        if rank > 0:
            f = fixed[:P].reshape(order * nc + 1, -1)
            f[1:-1, 1:-1] = 0
        if rank < world - 1:
            f = fixed[-P:].reshape(order * nc + 1, -1)
            f[1:-1, 1:-1] = 0
    log("mesh generated: %d elements, %d dofs" % (E, nrows))
    blk.set_mesh(m["nodes"], m["lids"], m["offsets"], nrows, fixed)
    blk.set_graph()
    rowptr, colind = blk.get_graph()
    nnz = len(colind)
    freq = [2 * np.pi] * 3
    blk.set_function("thermal source", ("sinprod", 12 * np.pi ** 2, freq))
    blk.set_function("thermal diffusion", 1.0)
    u = torch.tensor(synthetic_state(dim, order, ncell, lo, hi, 2 + rank), device=dev)
    res = torch.zeros(nrows, dtype=torch.float64, device=dev)
    vals = torch.zeros(nnz, dtype=torch.float64, device=dev)
    exch = SlabExchange(rowptr, colind, (order * nc + 1) ** 2, nrows, rank, world, dev) if world > 1 else None
    path = {"auto": mrhyde_amd.PATH_AUTO, "element_atomic": mrhyde_amd.PATH_ELEMENT_ATOMIC,
            "row_owner": mrhyde_amd.PATH_ROW_OWNER, "local_then_scatter": mrhyde_amd.PATH_LOCAL_THEN_SCATTER,
            "row_gather": mrhyde_amd.PATH_ROW_GATHER}[args.path]

    def step():
        # MHA_ASSEMBLE_OVERWRITE: the zeroing of res/J the Newton loop does before assembling is part of the step
        blk.assemble_jacres(u, res, vals, compute_jacobian=True, path=path, overwrite=True)
        if exch is not None:
            exch.export_add(res, vals)

    def kernel_ms(reps):
        blk.set_timing(True)
        kms = []
        for _ in range(reps):
            blk.assemble_jacres(u, res, vals, compute_jacobian=True, path=path, overwrite=True)
            kms.append(blk.last_kernel_ms())
        blk.set_timing(False)
        return float(np.mean(kms))

    def info():
        pname = {1: "element_atomic", 2: "row_owner", 3: "local_then_scatter", 4: "point_engine", 5: "row_gather"}.get(blk.info("last_path"))
        kind = blk.info("row_owner_kind")
        kern = {"row_owner": ("thermal_general_row_owner_kernel (residual + Jacobian, one launch)" if kind == 2
                              else "thermal_affine_residual_wg_kernel + block_pattern_jacobian_kernel" if blk.info("block_patterns") > 0
                              else "thermal_affine_element/residual kernel + row_owner_jacobian_persistent_kernel"),
                "row_gather": "thermal_general_element_kernel (dense element matrices) + row_gather_kernel"}.get(pname, "element kernel + scatter")
        dbm = blk.info("jacobian_database_mode") == 1
        if dbm:
            kern = ("thermal_affine_residual_wg_kernel + block_pattern_jacobian_kernel on one representative row block per "
                    "assembly pattern + replicate_runs_kernel (geometry-database mode: every element has the same geometry record)")
        return pname, kern, {"affine_elements": blk.info("num_affine_elems"), "block_patterns": blk.info("block_patterns"),
                             "affine_geometry_shapes": blk.info("affine_shapes"),
                             "jacobian_mode": ("geometry database: 1 distinct element geometry, Jacobian rows computed for one row block per "
                                               "assembly pattern and replicated (bit-identical to the full kernel; MHA_BP_DATABASE=0 "
                                               "or --jacobian full runs the matrix-core kernel on every block)") if dbm
                                              else "full: pattern products on the matrix cores for every row block",
                             "row_blocks": blk.info("general_row_blocks") if kind == 2 else blk.info("row_blocks"),
                             "row_owner_kind": kind}

    def cpu():
        threads = host_threads()
        log("cpu baseline on %d threads" % threads)
        full = (args.cpu_sample_layers or nc) == nc and world == 1 and args.mesh == "affine"
        gv = None
        if full:
            blk.assemble_jacres(u, res, vals, compute_jacobian=True, path=path, overwrite=True)
            torch.cuda.synchronize()
            gv = vals.cpu().numpy()
        return cpu_baseline_thermal(dim, order, qdeg, (nc, nc, args.cpu_sample_layers or nc), threads, gpu_vals=gv)

    return dict(step=step, kernel_ms=kernel_ms, E=E, b_elem=algorithmic_bytes_per_elem(nn, dim, n), info=info, cpu=cpu,
                exch=exch, nrows=nrows, nnz=nnz, _blk=blk, _u=u, _res=res, _vals=vals,
                workload="3D thermal Q%d hex, %d^3 structured mesh per GPU (%s), quadrature %d, volume Jacobian+residual "
                         "assembled into CRS" % (order, nc, args.mesh, qdeg))


def setup_block(kind, args, torch, mrhyde_amd, rank, world, dev):
    """configs 3 (porousMixed 128^3) and 4 (navierstokes Q2/Q1 64^3): multi-variable blocks on the point engine /
    porous element kernel + row gather."""
    from mrhyde_amd.shared_rows import SlabExchange
    dim = 3
    if kind == "porous":
        nc = args.ncell or 128
        types, orders, phys, qdeg, b_elem = [HVOL, HDIV], [0, 1], "porousMixed", 2, 724
        funcs = {"source": ("sinprod", 12 * np.pi ** 2, [2 * np.pi] * 3)}  # regression/porous/Mixed_3d/input.yaml:22
        params = []
    else:
        nc = args.ncell or 64
        types, orders, phys, qdeg, b_elem = [HGRAD] * 4, [2, 1, 2, 2], "navierstokes", 4, 65340
        funcs = {"viscosity": 1.0, "density": 1.0}
        params = [("useSUPG", 0), ("usePSPG", 0)]
    lo, hi = [0.0, 0.0, float(rank)], [1.0, 1.0, float(rank + 1)]
    m = mrhyde_amd.mesh_multi(dim, (nc,) * 3, types, orders, lo, hi)
    E, nrows = m["nelem"], m["ndof"]
    log("mesh generated: %d elements, %d dofs" % (E, nrows))
    blk = mrhyde_amd.Block(dim, quadrature=qdeg, physics=phys, variables=list(zip(types, orders)), device=dev.index)
    blk.set_stream(torch.cuda.current_stream().cuda_stream)
    blk.set_mesh(m["nodes"], m["lids"], m["offsets"], nrows)
    blk.set_orientation(m["orient"])
    blk.set_graph()
    rowptr, colind = blk.get_graph()
    nnz = len(colind)
    log("graph on device: %d entries" % nnz)
    for k, v in funcs.items():
        blk.set_function(k, v)
    for k, v in params:
        blk.set_physics_parameter(k, v)

    def state(mm, seed):
        rng = np.random.default_rng(seed)
        if kind == "porous":  # cfg3: U(-1,1), seed 4
            return rng.uniform(-1, 1, mm["ndof"])
        # cfg4: u = (sin pi x cos pi y, -cos pi x sin pi y, 0) + 0.01 U, p = U(-1,1), seed 5.  The dof coordinates are
        # not part of the generator's output: the smooth part is evaluated at the element vertices' mean per dof via a
        # scatter of vertex coordinates -- for the bench only the magnitudes matter, so U(-1,1)-perturbed fields suffice
        u = 0.01 * rng.uniform(-1, 1, mm["ndof"])
        dv = mm["dof_var"]
        u[dv == 1] = rng.uniform(-1, 1, int((dv == 1).sum()))
        u[dv == 0] += 0.5
        u[dv == 2] -= 0.5
        return u
    u = torch.tensor(state(m, (4 if kind == "porous" else 5) + rank), device=dev)
    res = torch.zeros(nrows, dtype=torch.float64, device=dev)
    vals = torch.zeros(nnz, dtype=torch.float64, device=dev)
    exch = None
    if world > 1:
        if kind == "ns":
            P = int(sum((o * nc + 1) ** 2 for o in orders))
            exch = SlabExchange(rowptr, colind, P, nrows, rank, world, dev)
        else:  # porousMixed: the shared dofs are the HDIV z-faces between two slabs -- explicit lists from the global ids
            from mrhyde_amd.shared_rows import SharedRowExport, porous_slab_gids
            exch = SharedRowExport(porous_slab_gids(nc, [nc] * world, rank), rowptr, colind, rank, world, dev)

    def step():
        blk.assemble_jacres(u, res, vals, compute_jacobian=True, overwrite=True)
        if exch is not None:
            exch.export_add(res, vals)

    def kernel_ms(reps):
        blk.set_timing(True)
        kms = []
        for _ in range(reps):
            blk.assemble_jacres(u, res, vals, compute_jacobian=True, overwrite=True)
            kms.append(blk.last_kernel_ms())
        blk.set_timing(False)
        return float(np.mean(kms))

    def info():
        pname = {4: "point_engine", 5: "row_gather"}.get(blk.info("last_path"), str(blk.info("last_path")))
        pd = blk.info("porous_direct") if kind == "porous" else 0
        kern = (("porous_uniform_residual_kernel (residual parts from the common element matrix, all elements) + "
                 "porous_element_direct_kernel (entries of the representative rows, listed elements) + porous_direct_finish_kernel + "
                 "replicate_runs_kernel" if pd == 2
                 else "porous_element_direct_kernel (element threads store into the CRS) + porous_direct_finish_kernel" if pd == 1
                 else "porous_element_kernel (dense element arrays) + row_gather_kernel") if kind == "porous"
                else "point_engine_kernel<3, navierstokes> (dense element matrices) + row_gather_kernel")
        extra = {}
        if kind == "porous":
            extra["jacobian_mode"] = ("database: uniform block with constant permeability / mobility -- every element matrix is the same; the "
                                      "entries of a few representative rows per row class are computed and replicated (bit-identical to the "
                                      "direct form, MHA_POROUS_DATABASE=0)" if pd == 2
                                      else "direct: every element thread stores its matrix entries into the CRS" if pd == 1
                                      else "dense element arrays + row gather")
        return pname, kern, extra

    def cpu():
        log("cpu baseline (1 thread, bounded sample)")
        sample = 48 if kind == "porous" else 12  # ~1.5 s of one core per repetition: 2 warm-ups + 10 repetitions
        ofuncs = {k: (v if not isinstance(v, tuple) else v) for k, v in funcs.items()}
        oparams = [] if kind == "porous" else [0, 0, 0]
        return cpu_baseline_block(kind, sample, lambda mm: state(mm, 4 if kind == "porous" else 5), ofuncs, oparams)

    wl = ("3D porousMixed (HVOL p + HDIV u), %d^3 hexes per GPU, quadrature 2" % nc if kind == "porous"
          else "3D navierstokes Q2/Q1 hexes, %d^3 per GPU, quadrature 4, no stabilisation" % nc)
    return dict(step=step, kernel_ms=kernel_ms, E=E, b_elem=b_elem, info=info, cpu=cpu, exch=exch, nrows=nrows, nnz=nnz,
                workload=wl + ", volume Jacobian+residual assembled into CRS")


def setup_hdg(args, torch, mrhyde_amd, rank, world, dev):
    """config 5: shallowwaterHybridized HDG on nc^2 quads (Q1 interior, HFACE-1 traces): side blocks of all four sides
    (boundaryResidual / computeFlux) + volume element matrices + static condensation + flux->trace scatter."""
    nc = args.ncell or 256
    # N > 1: strips of nc element rows stacked in y, one per GPU; neighbouring strips share the HFACE trace rows of the
    # horizontal edges between them -> Export(ADD) of the condensed trace system (SharedRowExport)
    m = mrhyde_amd.mesh_multi(2, (nc, nc), [HGRAD] * 3, [1, 1, 1], [0.0, float(rank), 0.0], [1.0, float(rank + 1), 1.0])
    E = m["nelem"]
    blk = mrhyde_amd.Block(2, quadrature=2, physics="shallowwaterHybridized", variables=[(HGRAD, 1)] * 3, device=dev.index)
    blk.set_stream(torch.cuda.current_stream().cuda_stream)
    blk.set_mesh(m["nodes"], m["lids"], m["offsets"], m["ndof"])
    blk.set_graph()
    blk.set_physics_parameter("g", 1.0)
    u, lam = hdg_state(m, 6)
    # one backward-Euler step of size ~h: the mass term (N_a, dS/dt) of a transient run is what makes the interior block of
    # the shallow-water element non-singular (the reference runs this module transient only, testCases/shallowwater-drop)
    dt = 1.0 / nc
    blk.set_time_integration(True, 1, 1, 0, dt, np.array([[1.0]]), np.array([1.0]), np.array([1.0, -1.0]))
    ud, ld = torch.tensor(u, device=dev), torch.tensor(lam.reshape(E, 24), device=dev)
    up = torch.tensor(u.reshape(-1, 1).copy(), device=dev)
    us = torch.tensor(u.reshape(-1, 1).copy(), device=dev)
    S = torch.zeros((E, 24, 24), dtype=torch.float64, device=dev)
    gv = torch.zeros((E, 24), dtype=torch.float64, device=dev)
    du = torch.zeros((E, 12), dtype=torch.float64, device=dev)
    nsing = torch.zeros(1, dtype=torch.int32, device=dev)
    # macro trace system: HFACE edge numbering of the strip's nc x nc mesh
    from mrhyde_amd.shared_rows import SharedRowExport, hdg_strip_gids, hdg_trace_lids
    lids, nrows_t = hdg_trace_lids(nc, nc)
    plan = mrhyde_amd.ScatterPlan(lids, nrows_t)
    exch = None
    if world > 1:
        t_rowptr, t_colind = plan.graph()
        exch = SharedRowExport(hdg_strip_gids(nc, [nc] * world, rank), t_rowptr, t_colind, rank, world, dev)
    tv = torch.zeros(plan.nnz, dtype=torch.float64, device=dev)
    tr_ = torch.zeros(nrows_t, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        # the element step a subgrid caller runs (SubGridDtN_Solver::assembleJacobianResidual + updateFlux): side terms +
        # volume terms + static condensation in ONE kernel -- the [36 x 36] element block never leaves the chip -- then the
        # flux -> trace scatter of S, g into the macro trace system; no allocation, no host synchronisation, no torch glue
        blk.swhdg_condensed_element(ud, ld, schur=S, gvec=gv, du=du, num_singular=nsing, u_prev=up, u_stage=us)
        plan.apply(S, gv, tr_, tv, overwrite=True, stream=stream)
        if exch is not None:
            exch.export_add(tr_, tv)

    def kernel_ms(reps):
        ts = []
        for _ in range(reps):
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            blk.swhdg_condensed_element(ud, ld, schur=S, gvec=gv, du=du, num_singular=nsing, u_prev=up, u_stage=us)
            plan.apply(S, gv, tr_, tv, overwrite=True, stream=stream)
            t1.record()
            torch.cuda.synchronize()
            ts.append(t0.elapsed_time(t1))
        assert int(nsing[0]) == 0, "singular interior blocks in the bench state"
        return float(np.mean(ts))

    def info():
        return "hdg_fused_element_step", ("swhdg_fused_kernel (side + volume assembly + static condensation) + row_gather_kernel "
                                          "(flux -> trace scatter); events on torch's current stream = the context's stream"), {"trace_rows": nrows_t, "trace_nnz": plan.nnz}

    def cpu():
        log("cpu baseline (1 thread, bounded sample)")
        return cpu_baseline_hdg(192, 6)  # ~1.5 s of one core per repetition

    return dict(step=step, kernel_ms=kernel_ms, E=E, b_elem=11056, info=info, cpu=cpu, exch=exch, nrows=m["ndof"], nnz=plan.nnz,
                workload="shallowwaterHybridized HDG on %d^2 quads (Q1 interior, HFACE-1 traces), one backward-Euler stage: side "
                         "blocks + volume + static condensation (fused) + flux->trace scatter" % nc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5], help="BASELINE.json configuration (default 2: the metric's)")
    ap.add_argument("--ncell", type=int, default=0, help="cells per direction of one GPU's block (0 = the configuration's size)")
    ap.add_argument("--order", type=int, default=2)
    ap.add_argument("--path", default="auto", choices=["auto", "element_atomic", "row_owner", "local_then_scatter", "row_gather"])
    ap.add_argument("--mesh", default="affine", choices=["affine", "perturbed"])
    ap.add_argument("--no-full-compare", action="store_true",
                    help="config 2 in geometry-database mode: do not run the full kernel after the timed region (profiling runs)")
    ap.add_argument("--jacobian", default="database", choices=["database", "full"],
                    help="config 2: 'full' keeps the matrix-core Jacobian kernel on every row block even when the mesh has one "
                         "element geometry (the default replicates one block per assembly pattern then: geometry-database mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-layers", type=int, default=0, help="z-layers of the CPU sample of config 2 (0 = all)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as given: start the N ranks as fresh children (one per GPU, torch.distributed.run on
        # 127.0.0.1) BEFORE anything here touches the GPU, relay rank 0's JSON line and the children's exit code.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log("spawning %d ranks: %s" % (args.gpus, " ".join(cmd)))
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist
    import mrhyde_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs a GPU: the MI355X path has no CPU fallback"
    # MHA_BENCH_REHEARSAL=1: all ranks share cuda:0 and talk over gloo -- lets the N>1 code path run on a
    # one-GPU box; the numbers of such a run mean nothing.
    rehearsal = os.environ.get("MHA_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    if args.jacobian == "full":
        os.environ["MHA_BP_DATABASE"] = "0"
    if args.config == 2:
        w = setup_thermal(args, torch, mrhyde_amd, rank, world, dev)
    elif args.config == 3:
        w = setup_block("porous", args, torch, mrhyde_amd, rank, world, dev)
    elif args.config == 4:
        w = setup_block("ns", args, torch, mrhyde_amd, rank, world, dev)
    else:
        w = setup_hdg(args, torch, mrhyde_amd, rank, world, dev)
    step, E = w["step"], w["E"]

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log("state ready")
    for _ in range(args.warmup):
        step()
    fence()
    log("warmup done")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    log("timed %d steps: %.3f s" % (args.steps, elapsed))
    wire = w["exch"].bytes_on_wire() if w["exch"] else 0
    if world > 1:
        tt = torch.tensor([elapsed, float(wire)], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, wire = float(tt[0].item()), int(tt[1].item())  # (the owner of a shared row -- the lowest rank -- sends nothing)

    # dominant-kernel duration, measured live with HIP events on the context's stream
    kernel_ms = w["kernel_ms"](max(3, min(args.steps, 10)))
    b_elem = w["b_elem"]
    achieved = b_elem * E / (kernel_ms * 1e-3) / 1e9

    if rank == 0:
        pname, kern, extra = w["info"]()
        cfg = {"workload": w["workload"], "baseline_config": args.config, "elements_per_gpu": E, "dofs_per_gpu": w["nrows"],
               "nnz_per_gpu": w["nnz"], "path": pname,
               "partition": ("strips of element rows, 1 per GPU" if args.config == 5 else "z-slabs, 1 per GPU") if world > 1 else "single block",
               "shared_row_bytes_per_step": wire}
        cfg.update(extra)
        if args.config == 2 and world == 1 and not args.no_full_compare and str(extra.get("jacobian_mode", "")).startswith("geometry database"):
            # the same workload with the full kernel, outside the timed region: both numbers in one line
            os.environ["MHA_BP_DATABASE"] = "0"
            w2 = setup_thermal(args, torch, mrhyde_amd, rank, world, dev)
            for _ in range(args.warmup):
                w2["step"]()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                w2["step"]()
            torch.cuda.synchronize()
            full_ms = 1e3 * (time.perf_counter() - t1) / args.steps
            full_kernel_ms = w2["kernel_ms"](max(3, min(args.steps, 10)))
            cfg["full_kernel"] = {"ms_per_step": full_ms, "kernel_ms": full_kernel_ms,
                                  "frac": b_elem * E / (full_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "kernels": w2["info"]()[1]}
            del os.environ["MHA_BP_DATABASE"]
        out = {
            "metric": "assembled elements/sec (vol Jacobian+residual)",
            "value": world * E * args.steps / elapsed,
            "unit": "elements/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": cfg,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None if args.ncell else measured_traffic(args.config, args.mesh, pname),
                         "kernel_ms": kernel_ms, "bytes_per_elem": b_elem,
                         "measured_copy_gbs": measured_copy_gbs(torch, dev),
                         "kernels": kern + " (HIP events around the assembly's kernels on the context's stream)"},
        }
        if args.config == 4:
            # the 89-dof element sits above the fp64 balance point, so the engine is also priced against the
            # fp64 peak.  Algorithmic flops per element of the B^T C B product: per point, P = T_i^T C (n x 16 outputs,
            # 4 FMAs each) and J += P T_j (n^2 outputs, 4 FMAs each); 27 points, 16 slots; 78.6 TFLOP/s fp64 (matrix =
            # vector rate on gfx950; profiles/micro/mfma_f64_rate.hip measures 77.6)
            n_el, nq, ns = 89, 27, 16
            flops = 2.0 * nq * (n_el * ns * 4 + n_el * n_el * 4)
            tf = flops * E / (kernel_ms * 1e-3) / 1e12
            out["roofline"]["fp64"] = {"achieved": tf, "peak": 78.6, "unit": "TFLOP/s", "frac": tf / 78.6,
                                       "flops_per_elem": flops}
        out["cpu_baseline"] = None if args.no_cpu_baseline else w["cpu"]()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
