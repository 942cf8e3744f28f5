#!/usr/bin/env python3
"""bench.py -- assembled elements/s of the thermal volume Jacobian+residual on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 is launched by torch.distributed.run,
one rank per GPU).  One "step" = one full volume assembly of the block: zero the residual and the CRS
values (what the Newton loop does before assembling, solverManager.cpp:1528-1533), then gather ->
residual+Jacobian -> scatter for every element (AssemblyManager::assembleJacRes, assemblyManager.cpp:2150-2665),
then -- for N>1 -- the Export(ADD) of the shared-DOF rows between neighbouring slabs.
Inputs are resident in HBM before the timed region.  Workload at N=1: BASELINE.json configs[1]
(3-D thermal, Q2 hex, 64^3 structured mesh, quadrature 4).  N>1: one 64^3 slab per GPU (weak scaling),
the slabs stacked in z so that neighbouring ranks share one dof plane.

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


def measured_traffic(args, blk):
    """HBM bytes per assembly from the committed PMC run of this exact workload (profiles/r1_traffic.json:
    FETCH_SIZE x2 + WRITE_SIZE, the gfx950 corrections of MI355X_MICROARCH.md), or None."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r1_traffic.json")))
        if (t["ncell"], t["order"], t["mesh"]) == (args.ncell, args.order, args.mesh) and blk.info("last_path") == t["path"]:
            return t["hbm_bytes_per_assembly"]
    except Exception:
        pass
    return None


def cpu_baseline(dim, order, qdeg, ncell_sample, threads, target_s=12.0, max_reps=8):
    """Oracle ("port" of the reference data flow) timed on the host cores on a bounded sample."""
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    m = oracle_lib.mesh_structured(dim, order, ncell_sample)
    u = synthetic_state(dim, order, ncell_sample, [0] * dim, [1] * dim, 2)
    pb = oracle_lib.physical_basis(dim, order, qdeg, m["nodes"])  # stored basis: setup, not timed (as in the reference)
    rowptr, colind = oracle_lib.build_graph(m["ndof"], m["lids"])
    freq = [2 * np.pi] * dim
    ts = []
    reps = 1
    while len(ts) < reps:
        t0 = time.perf_counter()
        oracle_lib.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, fixed=m["boundary"],
                                    pb=pb, workset_size=100, source=("sinprod", 4.0 * dim * np.pi ** 2, freq),
                                    num_threads=threads, rowptr=rowptr, colind=colind)
        ts.append(time.perf_counter() - t0)
        if len(ts) == 1:  # size the repetition count so that about target_s of CPU work is timed
            reps = int(min(max_reps, max(2, np.ceil(target_s / max(ts[0], 1e-3)))))
    t = float(np.median(ts))
    return {"value": m["nelem"] / t, "unit": "elements/s", "cores": threads, "kind": "port",
            "sample": "%s Q%d hex elements (%d), workset 100, stored basis, AD width %d, median of %d" % (
                "x".join(map(str, ncell_sample)), order, m["nelem"], oracle_lib.ad_width((order + 1) ** dim), reps),
            "seconds": t}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ncell", type=int, default=64, help="cells per direction of one GPU's block")
    ap.add_argument("--order", type=int, default=2)
    ap.add_argument("--path", default="auto", choices=["auto", "element_atomic", "row_owner", "local_then_scatter", "row_gather"])
    ap.add_argument("--mesh", default="affine", choices=["affine", "perturbed"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-layers", type=int, default=0, help="z-layers of the CPU sample (0 = all)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import mrhyde_amd
    from mrhyde_amd.shared_rows import SlabExchange

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "WORLD_SIZE (%d) != --gpus (%d): launch with torch.distributed.run" % (world, args.gpus)
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

    dim, order, qdeg = 3, args.order, 2 * args.order
    ncell = (args.ncell,) * 3
    lo, hi = [0.0, 0.0, float(rank)], [1.0, 1.0, float(rank + 1)]
    m = mrhyde_amd.mesh_structured(dim, order, ncell, lo, hi)
    if args.mesh == "perturbed":  # interior vertices moved by 0.15 h U(-1,1)^3, seed 3
        rng = np.random.default_rng(3)
        v = m["verts"]
        h = 1.0 / args.ncell
        interior = np.all((v - np.array(lo) > 1e-9) & (np.array(hi) - v > 1e-9), axis=1)
        v[interior] += 0.15 * h * rng.uniform(-1, 1, (int(interior.sum()), 3))
        m["nodes"] = np.ascontiguousarray(v[m["cell2vert"]])
    n, nn = m["lids"].shape[1], 2 ** dim
    E, nrows = m["nelem"], m["ndof"]

    blk = mrhyde_amd.Block(dim, order, quadrature=qdeg, workset_size=100, device=local_rank)
    blk.set_stream(torch.cuda.current_stream().cuda_stream)
    # Dirichlet rows: the physical boundary of the stacked domain (not the inter-slab planes)
    fixed = m["boundary"].copy()
    if world > 1:
        P = (order * args.ncell + 1) ** 2
        if rank > 0:
            f = fixed[:P].reshape(order * args.ncell + 1, -1)
            f[1:-1, 1:-1] = 0
        if rank < world - 1:
            f = fixed[-P:].reshape(order * args.ncell + 1, -1)
            f[1:-1, 1:-1] = 0
    log("mesh generated: %d elements, %d dofs" % (E, nrows))
    blk.set_mesh(m["nodes"], m["lids"], m["offsets"], nrows, fixed)
    blk.set_graph()
    log("mesh + graph on device")
    rowptr, colind = blk.get_graph()
    nnz = len(colind)
    freq = [2 * np.pi] * 3
    blk.set_function("thermal source", ("sinprod", 12 * np.pi ** 2, freq))
    blk.set_function("thermal diffusion", 1.0)
    u = torch.tensor(synthetic_state(dim, order, ncell, lo, hi, 2 + rank), device=dev)
    res = torch.zeros(nrows, dtype=torch.float64, device=dev)
    vals = torch.zeros(nnz, dtype=torch.float64, device=dev)
    exch = SlabExchange(rowptr, colind, (order * args.ncell + 1) ** 2, nrows, rank, world, dev) if world > 1 else None
    path = {"auto": mrhyde_amd.PATH_AUTO, "element_atomic": mrhyde_amd.PATH_ELEMENT_ATOMIC,
            "row_owner": mrhyde_amd.PATH_ROW_OWNER, "local_then_scatter": mrhyde_amd.PATH_LOCAL_THEN_SCATTER,
            "row_gather": mrhyde_amd.PATH_ROW_GATHER}[args.path]

    def step():
        # MHA_ASSEMBLE_OVERWRITE: the zeroing of res/J the Newton loop does before assembling is part of the step
        blk.assemble_jacres(u, res, vals, compute_jacobian=True, path=path, overwrite=True)
        if exch is not None:
            exch.export_add(res, vals)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log("state ready, nnz = %d" % nnz)
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
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # dominant-kernel duration, measured live with HIP events on the context's stream
    blk.set_timing(True)
    kms = []
    for _ in range(max(3, min(args.steps, 10))):
        blk.assemble_jacres(u, res, vals, compute_jacobian=True, path=path, overwrite=True)
        kms.append(blk.last_kernel_ms())
    blk.set_timing(False)
    kernel_ms = float(np.mean(kms))
    b_elem = algorithmic_bytes_per_elem(nn, dim, n)
    achieved = b_elem * E / (kernel_ms * 1e-3) / 1e9

    if rank == 0:
        out = {
            "metric": "assembled elements/sec (vol Jacobian+residual)",
            "value": world * E * args.steps / elapsed,
            "unit": "elements/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "3D thermal Q%d hex, %d^3 structured mesh per GPU (%s), quadrature %d, "
                                   "volume Jacobian+residual assembled into CRS" % (order, args.ncell, args.mesh, qdeg),
                       "elements_per_gpu": E, "dofs_per_gpu": nrows, "nnz_per_gpu": nnz,
                       "path": {1: "element_atomic", 2: "row_owner", 3: "local_then_scatter", 4: "point_engine",
                                5: "row_gather"}.get(blk.info("last_path")),
                       "affine_elements": blk.info("num_affine_elems"), "row_blocks": blk.info("row_blocks"),
                       "row_owner_lds_bytes": blk.info("row_owner_lds_bytes"),
                       "partition": "z-slabs, 1 per GPU" if world > 1 else "single block",
                       "shared_row_bytes_per_step": exch.bytes_on_wire() if exch else 0},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args, blk),
                         "kernel_ms": kernel_ms, "bytes_per_elem": b_elem,
                         "kernels": {2: "thermal_affine_element_kernel + row_owner_jacobian_persistent_kernel",
                                     5: "thermal_general_element_kernel (dense element matrices) + row_gather_kernel"
                                     }.get(blk.info("last_path"), "element kernel + scatter") +
                                    " (HIP events around the assembly's kernels on the context's stream)"},
        }
        if not args.no_cpu_baseline:
            threads = host_threads()
            log("cpu baseline on %d threads" % threads)
            out["cpu_baseline"] = cpu_baseline(dim, order, qdeg, (args.ncell, args.ncell, args.cpu_sample_layers or args.ncell), threads)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
