"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bar (BASELINE.json north_star): Jacobian entries within 1e-12 relative, integer maps bit-exact.
"Relative" is measured against the largest magnitude in the same array (element matrix block /
global CRS / residual vector), which is how an entry that is zero by cancellation is judged.
"""
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-12
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference")


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    scale = max(np.abs(b).max(), 1e-300)
    return np.abs(a - b).max() / scale


def crs_err(a, ref, factor=1.0):
    """Per-entry relative error of a CRS value array: max_ij |a_ij - ref_ij| / max(|ref_ij|, 1e-3 max_j |ref_ij|) -- the
    north star's "Jacobian entries within 1e-12 relative" with a cancellation floor of a thousandth of the row's largest
    entry (entries that are sums of cancelling element contributions carry the rounding of their terms).  The
    array-relative measure rel_err is asserted beside it."""
    a, b = np.asarray(a), factor * np.asarray(ref["crs_vals"])
    rowptr = np.asarray(ref["rowptr"])
    rows = np.repeat(np.arange(len(rowptr) - 1), np.diff(rowptr))
    rowmax = np.zeros(len(rowptr) - 1)
    np.maximum.at(rowmax, rows, np.abs(b))
    den = np.maximum(np.abs(b), 1e-3 * rowmax[rows])
    den = np.maximum(den, 1e-300)
    return max(float((np.abs(a - b) / den).max()), rel_err(a, b))


def perturbed(oracle, dim, order, ncell, seed, amp=0.15):
    m = oracle.mesh_structured(dim, order, ncell)
    rng = np.random.default_rng(seed)
    v = m["verts"].copy()
    interior = np.all((v > 1e-12) & (v < 1 - 1e-12), axis=1)
    h = 1.0 / np.asarray(ncell, dtype=np.float64)
    v[interior] += amp * h * rng.uniform(-1, 1, size=(interior.sum(), dim))
    m["verts"] = v
    m["nodes"] = np.ascontiguousarray(v[m["cell2vert"]])
    return m


def make_block(m, dim, order, qdeg, workset=100, fixed=None, graph=None):
    import mrhyde_amd
    blk = mrhyde_amd.Block(dim, order, quadrature=qdeg, workset_size=workset)
    blk.set_mesh(m["nodes"], m["lids"], m["offsets"], m["ndof"], fixed)
    if graph is None:
        blk.set_graph()
    else:
        blk.set_graph(*graph)
    return blk


CASES = [  # dim, order, qdeg, ncell
    (2, 1, 2, (7, 5)),
    (2, 2, 4, (4, 3)),
    (2, 4, 8, (3, 2)),
    (3, 1, 2, (4, 3, 2)),
    (3, 2, 4, (3, 3, 2)),
]


@pytest.mark.parametrize("dim,order,qdeg,ncell", CASES)
@pytest.mark.parametrize("path", ["element_atomic", "local_then_scatter"])
def test_jacres_matches_oracle(oracle, dim, order, qdeg, ncell, path):
    torch = _torch()
    import mrhyde_amd
    m = perturbed(oracle, dim, order, ncell, seed=3)
    rng = np.random.default_rng(11)
    u = rng.uniform(-1, 1, m["ndof"])
    fixed = m["boundary"]
    freq = [2 * np.pi] * dim
    amp = 4.0 * dim * np.pi ** 2
    ref = oracle.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, fixed=fixed,
                                  workset_size=7, source=("sinprod", amp, freq), diff=1.7, want_local=True)
    blk = make_block(m, dim, order, qdeg, workset=7, fixed=fixed, graph=(ref["rowptr"], ref["colind"]))
    blk.set_function("thermal source", ("sinprod", amp, freq))
    blk.set_function("thermal diffusion", 1.7)
    ud = torch.tensor(u, device="cuda")
    res = torch.zeros(m["ndof"], dtype=torch.float64, device="cuda")
    vals = torch.zeros(len(ref["colind"]), dtype=torch.float64, device="cuda")
    p = {"element_atomic": mrhyde_amd.PATH_ELEMENT_ATOMIC, "local_then_scatter": mrhyde_amd.PATH_LOCAL_THEN_SCATTER}[path]
    blk.assemble_jacres(ud, res, vals, path=p)
    torch.cuda.synchronize()
    assert crs_err(vals.cpu().numpy(), ref) < RTOL
    assert rel_err(res.cpu().numpy(), ref["res"]) < RTOL
    # fixed rows untouched by the scatter, then unit diagonal
    fr = np.flatnonzero(fixed)
    v = vals.cpu().numpy()
    for r in fr[:50]:
        assert np.all(v[ref["rowptr"][r]:ref["rowptr"][r + 1]] == 0.0)
    blk.apply_dbc_diag(vals)
    torch.cuda.synchronize()
    oracle.apply_dbc_diag(fixed, ref["rowptr"], ref["colind"], ref["crs_vals"])
    assert crs_err(vals.cpu().numpy(), ref) < RTOL


@pytest.mark.parametrize("dim,order,qdeg,ncell", CASES)
def test_local_jacres_matches_oracle(oracle, dim, order, qdeg, ncell):
    """updateJac / updateRes convention: element by element, entry by entry."""
    torch = _torch()
    m = perturbed(oracle, dim, order, ncell, seed=5)
    rng = np.random.default_rng(12)
    u = rng.uniform(-1, 1, m["ndof"])
    E, n = m["lids"].shape
    nq = oracle.ref_sizes(dim, order, qdeg)[1]
    src = rng.uniform(-2, 2, (E, nq))
    kap = rng.uniform(0.5, 2.0, (E, nq))
    ref = oracle.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, source=("array", src),
                                  diff_ip=kap, want_local=True, want_crs=False, want_res=False)
    blk = make_block(m, dim, order, qdeg)
    blk.set_function("thermal source", torch.tensor(src, device="cuda"))
    blk.set_function("thermal diffusion", torch.tensor(kap, device="cuda"))
    ud = torch.tensor(u, device="cuda")
    lJ = torch.zeros(E, n, n, dtype=torch.float64, device="cuda")
    lr = torch.zeros(E, n, dtype=torch.float64, device="cuda")
    blk.compute_local_jacres(ud, lJ, lr)
    torch.cuda.synchronize()
    J, r = lJ.cpu().numpy(), lr.cpu().numpy()
    for e in range(E):
        assert rel_err(J[e], ref["local_J"][e]) < RTOL, e
    assert rel_err(r, ref["local_res"]) < RTOL
    # residual-only (ScalarT) path leaves the Jacobian alone: assembleRes, assemblyManager.cpp:2946-3151
    lJ.zero_(); lr.zero_()
    blk.compute_local_jacres(ud, None, lr, compute_jacobian=False)
    torch.cuda.synchronize()
    assert rel_err(lr.cpu().numpy(), ref["local_res"]) < RTOL


@pytest.mark.parametrize("dim,order,qdeg,ncell,stage", [(2, 1, 2, (6, 5), 0), (3, 2, 4, (2, 3, 2), 1), (3, 1, 2, (3, 3, 3), 2)])
def test_transient_seeding_matches_oracle(oracle, dim, order, qdeg, ncell, stage):
    """computeSolnTransientSeeded (workset.cpp:589-623): 3-stage DIRK tableau, BDF-2 weights."""
    torch = _torch()
    m = perturbed(oracle, dim, order, ncell, seed=7)
    rng = np.random.default_rng(13)
    nd = m["ndof"]
    u = rng.uniform(-1, 1, nd)
    nsteps, nstages = 2, 3
    A = np.array([[0.4358665215, 0, 0], [0.2820667392, 0.4358665215, 0], [1.208496649, -0.644363171, 0.4358665215]])
    b = np.array([1.208496649, -0.644363171, 0.4358665215])
    bdf = np.array([1.5, -2.0, 0.5])
    tr = dict(u_prev=rng.uniform(-1, 1, (nd, nsteps)), u_stage=rng.uniform(-1, 1, (nd, nstages)), stage=stage,
              butcher_A=A, butcher_b=b, bdf=bdf, dt=0.013)
    ref = oracle.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, transient=tr, rho=1.3,
                                  cp=0.7, diff=0.9, source=("const", 0.4))
    blk = make_block(m, dim, order, qdeg, graph=(ref["rowptr"], ref["colind"]))
    blk.set_function("thermal source", 0.4)
    blk.set_function("thermal diffusion", 0.9)
    blk.set_function("density", 1.3)
    blk.set_function("specific heat", 0.7)
    blk.set_time_integration(True, nsteps, nstages, stage, 0.013, A, b, bdf)
    t = lambda a: torch.tensor(a, device="cuda")
    res = torch.zeros(nd, dtype=torch.float64, device="cuda")
    vals = torch.zeros(len(ref["colind"]), dtype=torch.float64, device="cuda")
    blk.assemble_jacres(t(u), res, vals, u_prev=t(tr["u_prev"]), u_stage=t(tr["u_stage"]))
    torch.cuda.synchronize()
    assert crs_err(vals.cpu().numpy(), ref) < RTOL
    assert rel_err(res.cpu().numpy(), ref["res"]) < RTOL


@pytest.mark.parametrize("dim,order,qdeg,ncell", [(2, 1, 2, (5, 4)), (3, 2, 4, (2, 2, 3)), (3, 1, 2, (3, 2, 2))])
def test_workset_views_match_oracle(oracle, dim, order, qdeg, ncell):
    """basis / basis_grad / wts / x,y,z views of every workset (ragged last workset included)."""
    _torch()
    m = perturbed(oracle, dim, order, ncell, seed=9)
    pb = oracle.physical_basis(dim, order, qdeg, m["nodes"])
    ws = 7
    blk = make_block(m, dim, order, qdeg, workset=ws)
    E = m["nelem"]
    assert blk.num_worksets() == (E + ws - 1) // ws
    for w in range(blk.num_worksets()):
        blk.workset_update(w)
        e0, e1 = w * ws, min(E, (w + 1) * ws)
        b = blk.workset_view_numpy("basis")
        assert b.shape == (e1 - e0, m["lids"].shape[1], pb["basis"].shape[2], 1)
        assert rel_err(b[..., 0], pb["basis"][e0:e1]) < RTOL
        assert rel_err(blk.workset_view_numpy("basis_grad"), pb["basis_grad"][e0:e1]) < RTOL
        assert rel_err(blk.workset_view_numpy("wts"), pb["wts"][e0:e1]) < RTOL
        for k, name in enumerate("xyz"[:dim]):
            assert rel_err(blk.workset_view_numpy(name), pb["ip"][e0:e1, :, k]) < RTOL
        assert np.array_equal(blk.workset_view_numpy("LIDs"), m["lids"][e0:e1])  # integer map bit-exact
    assert np.array_equal(blk.workset_view_numpy("offsets")[0], m["offsets"])
    import mrhyde_amd
    with pytest.raises(mrhyde_amd.MhaError) as ei:
        blk.workset_view("grad(q)[x]")
    assert ei.value.code == 4


def test_hgrad_gold_through_device_views():
    """The reference's own basis gold (regression/discretization/HGRAD) read back from the device views."""
    _torch()
    import mrhyde_amd
    txt = open(os.path.join(GOLD, "discretization_HGRAD.gold")).read()
    vals = re.findall(r"^dof (\d+), point (\d+): ([-0-9.e]+)$", txt, flags=re.M)
    grads = re.findall(r"^dof (\d+), point (\d+) grad: \(([-0-9.e,]+)\)$", txt, flags=re.M)
    for dim, nv in ((2, vals[:16]), (3, vals[16:])):
        m = mrhyde_amd.mesh_structured(dim, 1, (1,) * dim)
        blk = make_block(m, dim, 1, 2)
        blk.workset_update(0)
        b = blk.workset_view_numpy("basis")
        g = blk.workset_view_numpy("basis_grad")
        for dof, pt, v in nv:
            assert float("%.6g" % b[0, int(dof), int(pt), 0]) == float(v)
        for dof, pt, tup in grads:
            comp = [float(x) for x in tup.split(",")]
            if len(comp) != dim:
                continue
            for d in range(dim):
                assert float("%.6g" % g[0, int(dof), int(pt), d]) == comp[d]


def test_gather_bit_exact(oracle):
    torch = _torch()
    m = oracle.mesh_structured(3, 2, (3, 2, 2))
    blk = make_block(m, 3, 2, 4)
    u = np.random.default_rng(1).uniform(-1, 1, m["ndof"])
    out = torch.zeros(m["lids"].shape, dtype=torch.float64, device="cuda")
    blk.gather(torch.tensor(u, device="cuda"), out)
    torch.cuda.synchronize()
    # data(e,dof) = u[LIDs(e, offsets(dof))]  (assemblyManager.cpp:3633-3641)
    assert np.array_equal(out.cpu().numpy(), u[m["lids"][:, m["offsets"]]])


def test_thermal_gold_end_to_end(oracle):
    """regression/thermal/2D_verification: assemble on the GPU, solve on the host, L2 error = 0.00102776."""
    torch = _torch()
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    import mrhyde_amd
    dim, order, qdeg = 2, 1, 2
    m = mrhyde_amd.mesh_structured(dim, order, (40, 40))
    blk = make_block(m, dim, order, qdeg, workset=100, fixed=m["boundary"])
    freq = [2 * np.pi] * 2
    blk.set_function("thermal source", ("sinprod", 8 * np.pi ** 2, freq))
    rowptr, colind = blk.get_graph()
    u = torch.zeros(m["ndof"], dtype=torch.float64, device="cuda")
    res = torch.zeros_like(u)
    vals = torch.zeros(len(colind), dtype=torch.float64, device="cuda")
    blk.assemble_jacres(u, res, vals)
    blk.apply_dbc_diag(vals)
    torch.cuda.synchronize()
    J = sp.csr_matrix((vals.cpu().numpy(), colind, rowptr), shape=(m["ndof"],) * 2)
    du = spla.spsolve(J.tocsc(), res.cpu().numpy())
    pb = oracle.physical_basis(dim, order, qdeg, m["nodes"])
    err = oracle.l2_error_sinprod(dim, order, qdeg, m["lids"], m["offsets"], pb, du, freq)
    gold = float(re.search(r"for e = ([-0-9.e]+)", open(os.path.join(GOLD, "thermal_2D_verification.gold")).read()).group(1))
    assert "%.6g" % err == "%.6g" % gold


def affine_mesh(oracle, dim, order, ncell, shear=True):
    """Structured mesh pushed through an affine map: every element is a (non-rectangular) parallelepiped."""
    m = oracle.mesh_structured(dim, order, ncell)
    if shear:
        A = np.eye(dim) + 0.25 * np.array([[0.3, 0.8, -0.4], [-0.5, 0.2, 0.6], [0.7, -0.3, 0.1]])[:dim, :dim]
        m["verts"] = m["verts"] @ A.T + 0.3
        m["nodes"] = np.ascontiguousarray(m["verts"][m["cell2vert"]])
    return m


@pytest.mark.parametrize("adjoint,lump", [(True, False), (False, True), (True, True)])
def test_scatter_options_adjoint_and_lumped_mass(oracle, adjoint, lump):
    """The scatter options of AssemblyManager::scatter (assemblyManager.cpp:4124-4133), as the reference has them:
    isAdjoint_ -> every column of a row gets res(elem,row).fastAccessDx(row); lump_mass_ -> cols[col] = rowIndex (all of
    a row's element contributions land on its diagonal).  Reference: the ORACLE's element Jacobians scattered with that
    rule in numpy; fixed rows skipped."""
    torch = _torch()
    import mrhyde_amd
    dim, order, qdeg, ncell = 3, 2, 4, (3, 2, 2)
    m = perturbed(oracle, dim, order, ncell, seed=21)
    rng = np.random.default_rng(8)
    nd = m["ndof"]
    u = rng.uniform(-1, 1, nd)
    fixed = m["boundary"]
    # transient: lumping a pure stiffness matrix gives row sums of zero (roundoff only); the mass term is what gets lumped
    nsteps, nstages, stage = 1, 1, 0
    A, b, bdf = np.array([[1.0]]), np.array([1.0]), np.array([1.0, -1.0])
    tr = dict(u_prev=rng.uniform(-1, 1, (nd, nsteps)), u_stage=rng.uniform(-1, 1, (nd, nstages)), stage=stage,
              butcher_A=A, butcher_b=b, bdf=bdf, dt=0.05)
    ref = oracle.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, fixed=fixed, transient=tr,
                                  rho=1.3, cp=0.7, source=("const", 0.5), diff=1.1, want_local=True)
    rowptr, colind = ref["rowptr"], ref["colind"]
    expect = np.zeros(len(colind))
    n = m["lids"].shape[1]
    for e, L in enumerate(m["lids"]):
        Je = ref["local_J"][e]                      # [n][n] in LID-position order (updateJac convention)
        for i in range(n):
            r = L[i]
            if fixed[r]:
                continue
            lo, hi = rowptr[r], rowptr[r + 1]
            for j in range(n):
                v = Je[i, i] if adjoint else Je[i, j]
                c = r if lump else L[j]
                expect[lo + np.searchsorted(colind[lo:hi], c)] += v
    blk = make_block(m, dim, order, qdeg, fixed=fixed, graph=(rowptr, colind))
    blk.set_function("thermal source", 0.5)
    blk.set_function("thermal diffusion", 1.1)
    blk.set_function("density", 1.3)
    blk.set_function("specific heat", 0.7)
    blk.set_time_integration(True, nsteps, nstages, stage, 0.05, A, b, bdf)
    t = lambda a: torch.tensor(a, device="cuda")
    res = torch.zeros(nd, dtype=torch.float64, device="cuda")
    vals = torch.full((len(colind),), 9.0, dtype=torch.float64, device="cuda")
    blk.assemble_jacres(t(u), res, vals, u_prev=t(tr["u_prev"]), u_stage=t(tr["u_stage"]), overwrite=True, adjoint=adjoint,
                        lump_mass=lump)
    torch.cuda.synchronize()
    assert blk.info("last_path") == mrhyde_amd.PATH_ROW_GATHER
    assert rel_err(vals.cpu().numpy(), expect) < RTOL
    assert rel_err(res.cpu().numpy(), ref["res"]) < RTOL      # the residual does not depend on the options
    with pytest.raises(mrhyde_amd.MhaError):
        blk.assemble_jacres(t(u), res, vals, u_prev=t(tr["u_prev"]), u_stage=t(tr["u_stage"]),
                            path=mrhyde_amd.PATH_ELEMENT_ATOMIC, adjoint=True)


def test_caller_graph_is_validated(oracle):
    """A caller-supplied CRS graph that lacks an element coupling (or is unsorted) is an input error at mha_set_graph /
    mha_scatter_plan_create: the device slot maps are built by a column search and a missing column has no slot."""
    _torch()
    import mrhyde_amd
    dim, order, qdeg, ncell = 2, 1, 2, (3, 3)
    m = oracle.mesh_structured(dim, order, ncell)
    rowptr, colind = oracle.build_graph(m["ndof"], m["lids"])
    blk = mrhyde_amd.Block(dim, order, quadrature=qdeg)
    blk.set_mesh(m["nodes"], m["lids"], m["offsets"], m["ndof"])
    # drop the last column of row 5
    keep = np.ones(len(colind), bool)
    keep[rowptr[6] - 1] = False
    rp = rowptr.copy()
    rp[6:] -= 1
    with pytest.raises(mrhyde_amd.MhaError, match="misses a coupling"):
        blk.set_graph(rp, colind[keep])
    # unsorted row
    ci = colind.copy()
    ci[rowptr[5]], ci[rowptr[5] + 1] = ci[rowptr[5] + 1], ci[rowptr[5]]
    with pytest.raises(mrhyde_amd.MhaError, match="ascending"):
        blk.set_graph(rowptr, ci)
    with pytest.raises(mrhyde_amd.MhaError, match="misses a coupling"):
        mrhyde_amd.ScatterPlan(m["lids"], m["ndof"], rp, colind[keep])
    blk.set_graph(rowptr, colind)  # the complete graph is accepted


def test_baseline_kernel_knob_on_the_row_gather_path(oracle, monkeypatch):
    """MHA_BASELINE_ELEMENT_KERNEL=1 (the documented cross-check knob) on a perturbed mesh with AUTO: the baseline element
    kernel accumulates into the row gather's scratch, which must therefore be zeroed every call -- two assemblies in a
    row, both equal to the oracle (the scratch used to keep the previous call's contents)."""
    torch = _torch()
    import mrhyde_amd
    monkeypatch.setenv("MHA_BASELINE_ELEMENT_KERNEL", "1")
    dim, order, qdeg, ncell = 3, 2, 4, (3, 3, 2)
    m = perturbed(oracle, dim, order, ncell, seed=12)
    u = np.random.default_rng(6).uniform(-1, 1, m["ndof"])
    fixed = m["boundary"]
    ref = oracle.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, fixed=fixed, source=("const", 1.5))
    blk = make_block(m, dim, order, qdeg, fixed=fixed, graph=(ref["rowptr"], ref["colind"]))
    blk.set_function("thermal source", 1.5)
    res = torch.full((m["ndof"],), 3.0, dtype=torch.float64, device="cuda")
    vals = torch.full((len(ref["colind"]),), 3.0, dtype=torch.float64, device="cuda")
    for _ in range(2):
        blk.assemble_jacres(torch.tensor(u, device="cuda"), res, vals, overwrite=True, path=mrhyde_amd.PATH_ROW_GATHER)
        torch.cuda.synchronize()
        assert blk.info("last_path") == mrhyde_amd.PATH_ROW_GATHER
        assert crs_err(vals.cpu().numpy(), ref) < RTOL
        assert rel_err(res.cpu().numpy(), ref["res"]) < RTOL


RO_CASES = [  # dim, order, qdeg, ncell
    (2, 1, 2, (9, 7)),
    (2, 2, 4, (6, 5)),
    (2, 4, 8, (5, 4)),
    (3, 1, 2, (5, 4, 3)),
    (3, 2, 4, (5, 4, 6)),
]


def _select_k1(monkeypatch, k1):
    """K1 forms (thermal_affine_residual.hip / thermal_row_owner.hip): wg = workgroup-merged (the default), morton = the
    same with the elements grouped along a Morton curve, thread = one thread per element with global atomics, lanes =
    32 lanes per element."""
    if k1 in ("wg", "morton"):
        monkeypatch.delenv("MHA_K1", raising=False)
        monkeypatch.setenv("MHA_K1_ORDER", "morton" if k1 == "morton" else "natural")
    else:
        monkeypatch.setenv("MHA_K1", k1)


@pytest.mark.parametrize("dim,order,qdeg,ncell", RO_CASES)
@pytest.mark.parametrize("mode", ["accumulate", "overwrite"])
@pytest.mark.parametrize("k2,k1,shear", [("pattern", "wg", True), ("pattern", "wg", False), ("pattern", "morton", True),
                                          ("pattern", "morton", False), ("pattern", "thread", True),
                                          ("blocks", "lanes", True), ("blocks", "wg", False)])
def test_row_owner_matches_oracle(oracle, monkeypatch, dim, order, qdeg, ncell, mode, k2, k1, shear):
    """Fused row-owner kernel on affine meshes (sheared: general source evaluation; axis-aligned: the separable one) vs
    the oracle: CRS values, residual, fixed rows.
    k2 = pattern (the default): the Jacobian rows as block-pattern GEMMs on the matrix cores (block_pattern.hip);
    k2 = blocks: the LDS-accumulator row-block kernel (the fallback for meshes whose blocks do not group)."""
    torch = _torch()
    import mrhyde_amd
    monkeypatch.setenv("MHA_K2", k2)
    _select_k1(monkeypatch, k1)
    m = affine_mesh(oracle, dim, order, ncell, shear=shear)
    rng = np.random.default_rng(21)
    u = rng.uniform(-1, 1, m["ndof"])
    fixed = m["boundary"]
    freq = [1.3, 0.7, 2.1][:dim]
    ref = oracle.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, fixed=fixed,
                                  source=("sinprod", 3.0, freq), diff=1.7)
    blk = make_block(m, dim, order, qdeg, fixed=fixed, graph=(ref["rowptr"], ref["colind"]))
    blk.set_function("thermal source", ("sinprod", 3.0, freq))
    blk.set_function("thermal diffusion", 1.7)
    ud = torch.tensor(u, device="cuda")
    if mode == "overwrite":  # garbage in, every entry must be overwritten (fixed rows: zeros)
        res = torch.full((m["ndof"],), 7.0, dtype=torch.float64, device="cuda")
        vals = torch.full((len(ref["colind"]),), -3.0, dtype=torch.float64, device="cuda")
    else:
        res = torch.zeros(m["ndof"], dtype=torch.float64, device="cuda")
        vals = torch.zeros(len(ref["colind"]), dtype=torch.float64, device="cuda")
    blk.assemble_jacres(ud, res, vals, path=mrhyde_amd.PATH_ROW_OWNER, overwrite=(mode == "overwrite"))
    torch.cuda.synchronize()
    assert blk.info("num_affine_elems") == m["nelem"] and blk.info("last_path") == mrhyde_amd.PATH_ROW_OWNER
    assert (blk.info("block_patterns") > 0) == (k2 == "pattern")
    assert crs_err(vals.cpu().numpy(), ref) < RTOL
    assert rel_err(res.cpu().numpy(), ref["res"]) < RTOL
    v = vals.cpu().numpy()
    for r in np.flatnonzero(fixed)[:40]:
        assert np.all(v[ref["rowptr"][r]:ref["rowptr"][r + 1]] == 0.0)
    # accumulate semantics: a second call doubles everything
    if mode == "accumulate":
        blk.assemble_jacres(ud, res, vals, path=mrhyde_amd.PATH_ROW_OWNER)
        torch.cuda.synchronize()
        assert crs_err(vals.cpu().numpy(), ref, 2.0) < RTOL
        assert rel_err(res.cpu().numpy(), 2 * ref["res"]) < RTOL
    # residual-only pass (assembleRes): Jacobian untouched
    keep = vals.clone()
    res.zero_()
    blk.assemble_jacres(ud, res, vals, compute_jacobian=False, path=mrhyde_amd.PATH_ROW_OWNER)
    torch.cuda.synchronize()
    assert torch.equal(keep, vals)
    assert rel_err(res.cpu().numpy(), ref["res"]) < RTOL


@pytest.mark.parametrize("k2,k1", [("blocks", "lanes"), ("pattern", "thread"), ("pattern", "wg"), ("pattern", "morton")])
def test_row_owner_transient_and_source_array(oracle, monkeypatch, k2, k1):
    torch = _torch()
    import mrhyde_amd
    monkeypatch.setenv("MHA_K2", k2)
    _select_k1(monkeypatch, k1)
    dim, order, qdeg, ncell = 3, 2, 4, (3, 4, 3)
    m = affine_mesh(oracle, dim, order, ncell)
    rng = np.random.default_rng(23)
    nd = m["ndof"]
    u = rng.uniform(-1, 1, nd)
    E = m["nelem"]
    nq = oracle.ref_sizes(dim, order, qdeg)[1]
    src = rng.uniform(-2, 2, (E, nq))
    nsteps, nstages, stage = 2, 2, 1
    A = np.array([[0.2928932188, 0.0], [0.7071067812, 0.2928932188]])
    b = np.array([0.7071067812, 0.2928932188])
    bdf = np.array([1.5, -2.0, 0.5])
    tr = dict(u_prev=rng.uniform(-1, 1, (nd, nsteps)), u_stage=rng.uniform(-1, 1, (nd, nstages)), stage=stage,
              butcher_A=A, butcher_b=b, bdf=bdf, dt=0.02)
    ref = oracle.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, transient=tr, rho=1.3,
                                  cp=0.7, diff=0.9, source=("array", src), fixed=m["boundary"])
    blk = make_block(m, dim, order, qdeg, fixed=m["boundary"], graph=(ref["rowptr"], ref["colind"]))
    blk.set_function("thermal source", torch.tensor(src, device="cuda"))
    blk.set_function("thermal diffusion", 0.9)
    blk.set_function("density", 1.3)
    blk.set_function("specific heat", 0.7)
    blk.set_time_integration(True, nsteps, nstages, stage, 0.02, A, b, bdf)
    t = lambda a: torch.tensor(a, device="cuda")
    res = torch.zeros(nd, dtype=torch.float64, device="cuda")
    vals = torch.zeros(len(ref["colind"]), dtype=torch.float64, device="cuda")
    blk.assemble_jacres(t(u), res, vals, u_prev=t(tr["u_prev"]), u_stage=t(tr["u_stage"]), path=mrhyde_amd.PATH_ROW_OWNER)
    torch.cuda.synchronize()
    assert crs_err(vals.cpu().numpy(), ref) < RTOL
    assert rel_err(res.cpu().numpy(), ref["res"]) < RTOL


@pytest.mark.parametrize("order_env", ["natural", "morton", None])
@pytest.mark.parametrize("transient", [False, True])
def test_merged_residual_kernel_on_a_shuffled_numbering(oracle, monkeypatch, order_env, transient):
    """The workgroup-merged K1 takes 256 elements per workgroup through a host-made plan (distinct rows + 16-bit
    positions).  Here: 3 workgroups (the last one partial), the elements renumbered at random (a group's rows then
    spread over the whole mesh; with MHA_K1_ORDER unset the plan may regroup them along a Morton curve), axis-aligned
    elements with the closed-form source (the separable evaluation), steady and transient."""
    torch = _torch()
    import mrhyde_amd
    monkeypatch.delenv("MHA_K1", raising=False)
    if order_env:
        monkeypatch.setenv("MHA_K1_ORDER", order_env)
    else:
        monkeypatch.delenv("MHA_K1_ORDER", raising=False)
    dim, order, qdeg, ncell = 3, 2, 4, (9, 8, 9)
    m = affine_mesh(oracle, dim, order, ncell, shear=False)
    rng = np.random.default_rng(77)
    perm = rng.permutation(m["nelem"])
    m["nodes"] = np.ascontiguousarray(m["nodes"][perm])
    m["lids"] = np.ascontiguousarray(m["lids"][perm])
    nd = m["ndof"]
    u = rng.uniform(-1, 1, nd)
    fixed = m["boundary"]
    freq = [1.3, 0.7, 2.1]
    kw = {}
    if transient:
        nsteps, nstages, stage = 2, 2, 1
        A = np.array([[0.2928932188, 0.0], [0.7071067812, 0.2928932188]])
        bb = np.array([0.7071067812, 0.2928932188])
        bdf = np.array([1.5, -2.0, 0.5])
        tr = dict(u_prev=rng.uniform(-1, 1, (nd, nsteps)), u_stage=rng.uniform(-1, 1, (nd, nstages)), stage=stage,
                  butcher_A=A, butcher_b=bb, bdf=bdf, dt=0.02)
        kw = dict(transient=tr, rho=1.3, cp=0.7)
    ref = oracle.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, fixed=fixed,
                                  source=("sinprod", 3.0, freq), diff=1.7, **kw)
    blk = make_block(m, dim, order, qdeg, fixed=fixed, graph=(ref["rowptr"], ref["colind"]))
    blk.set_function("thermal source", ("sinprod", 3.0, freq))
    blk.set_function("thermal diffusion", 1.7)
    t = lambda a: torch.tensor(a, device="cuda")
    res = torch.full((nd,), 7.0, dtype=torch.float64, device="cuda")
    vals = torch.zeros(len(ref["colind"]), dtype=torch.float64, device="cuda")
    if transient:
        blk.set_function("density", 1.3)
        blk.set_function("specific heat", 0.7)
        blk.set_time_integration(True, nsteps, nstages, stage, 0.02, A, bb, bdf)
        blk.assemble_jacres(t(u), res, vals, u_prev=t(tr["u_prev"]), u_stage=t(tr["u_stage"]),
                            path=mrhyde_amd.PATH_ROW_OWNER, overwrite=True)
    else:
        blk.assemble_jacres(t(u), res, vals, path=mrhyde_amd.PATH_ROW_OWNER, overwrite=True)
    torch.cuda.synchronize()
    assert blk.info("num_affine_elems") == m["nelem"]
    assert rel_err(res.cpu().numpy(), ref["res"]) < RTOL
    assert crs_err(vals.cpu().numpy(), ref) < RTOL


def test_auto_path_selection(oracle):
    """AUTO = the row-owner family wherever it exists: the fused affine kernels on affine meshes with constant
    coefficients (kind 1), the general-element row-owner kernel otherwise (kind 2); both correct, the general one also
    with accumulate semantics, fixed rows and a residual-only pass; the element-matrix + row-gather path stays
    selectable and agrees."""
    torch = _torch()
    import mrhyde_amd
    dim, order, qdeg, ncell = 3, 2, 4, (3, 3, 3)
    for mesh_fn, kind in ((lambda: affine_mesh(oracle, dim, order, ncell), 1),
                          (lambda: perturbed(oracle, dim, order, ncell, seed=2), 2)):
        m = mesh_fn()
        u = np.random.default_rng(5).uniform(-1, 1, m["ndof"])
        ref = oracle.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, source=("const", 1.0))
        blk = make_block(m, dim, order, qdeg, graph=(ref["rowptr"], ref["colind"]))
        blk.set_function("thermal source", 1.0)
        res = torch.zeros(m["ndof"], dtype=torch.float64, device="cuda")
        vals = torch.zeros(len(ref["colind"]), dtype=torch.float64, device="cuda")
        blk.assemble_jacres(torch.tensor(u, device="cuda"), res, vals, overwrite=True)
        torch.cuda.synchronize()
        assert blk.info("last_path") == mrhyde_amd.PATH_ROW_OWNER and blk.info("row_owner_kind") == kind
        assert crs_err(vals.cpu().numpy(), ref) < RTOL
        assert rel_err(res.cpu().numpy(), ref["res"]) < RTOL
        if kind == 2:
            assert blk.info("general_row_blocks") > 0
            # accumulate on top of the first result, then a residual-only overwrite
            blk.assemble_jacres(torch.tensor(u, device="cuda"), res, vals)
            torch.cuda.synchronize()
            assert crs_err(vals.cpu().numpy(), ref, 2.0) < RTOL and rel_err(res.cpu().numpy(), 2 * ref["res"]) < RTOL
            blk.assemble_jacres(torch.tensor(u, device="cuda"), res, None, compute_jacobian=False, overwrite=True)
            torch.cuda.synchronize()
            assert rel_err(res.cpu().numpy(), ref["res"]) < RTOL
            assert crs_err(vals.cpu().numpy(), ref, 2.0) < RTOL      # the matrix was left alone
            # the explicit row-owner request lands on the same kernel; the dense path is still there
            for path in (mrhyde_amd.PATH_ROW_OWNER, mrhyde_amd.PATH_ROW_GATHER):
                res.fill_(4.0)
                vals.fill_(4.0)
                blk.assemble_jacres(torch.tensor(u, device="cuda"), res, vals, overwrite=True, path=path)
                torch.cuda.synchronize()
                assert blk.info("last_path") == path
                assert crs_err(vals.cpu().numpy(), ref) < RTOL and rel_err(res.cpu().numpy(), ref["res"]) < RTOL
            # fixed rows: skipped by the owner, zeroed by the overwrite
            fixed = m["boundary"]
            reff = oracle.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, fixed=fixed,
                                           source=("const", 1.0))
            blk2 = make_block(m, dim, order, qdeg, fixed=fixed, graph=(ref["rowptr"], ref["colind"]))
            blk2.set_function("thermal source", 1.0)
            res.fill_(5.0)
            vals.fill_(5.0)
            blk2.assemble_jacres(torch.tensor(u, device="cuda"), res, vals, overwrite=True)
            torch.cuda.synchronize()
            assert blk2.info("row_owner_kind") == 2
            assert crs_err(vals.cpu().numpy(), reff) < RTOL and rel_err(res.cpu().numpy(), reff["res"]) < RTOL
            # ... and left untouched when accumulating
            blk2.assemble_jacres(torch.tensor(u, device="cuda"), res, vals)
            torch.cuda.synchronize()
            assert crs_err(vals.cpu().numpy(), reff, 2.0) < RTOL and rel_err(res.cpu().numpy(), 2 * reff["res"]) < RTOL


GENERAL_RO_CASES = [  # dim, order, qdeg, ncell: every instantiation of kernels/thermal_general_row_owner.hip
    (2, 1, 2, (7, 6)), (2, 2, 4, (6, 5)), (2, 3, 6, (4, 3)), (2, 4, 8, (4, 3)), (3, 1, 2, (4, 3, 3)), (3, 2, 4, (4, 3, 2)),
]


@pytest.mark.parametrize("dim,order,qdeg,ncell", GENERAL_RO_CASES)
@pytest.mark.parametrize("mode", ["steady-sinprod", "transient-arrays", "expression"])
def test_general_row_owner_matches_oracle(oracle, dim, order, qdeg, ncell, mode):
    """The general-element row-owner kernel against the oracle: perturbed (non-affine) meshes, coefficient functions
    given as constants / closed forms / per-point arrays / deck expressions, steady and transient (3-stage DIRK, BDF-2),
    fixed rows, overwrite on garbage.  Residual and Jacobian come out of ONE launch with no atomics on global memory."""
    torch = _torch()
    import mrhyde_amd
    m = perturbed(oracle, dim, order, ncell, seed=31)
    rng = np.random.default_rng(17)
    nd = m["ndof"]
    u = rng.uniform(-1, 1, nd)
    fixed = m["boundary"]
    pb = oracle.physical_basis(dim, order, qdeg, m["nodes"])
    E, nq = pb["wts"].shape
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")
    kw, okw, tr = {}, {}, None
    funcs = {}
    if mode == "steady-sinprod":
        fr = [2.0, 1.0, 1.5][:dim]
        okw = dict(source=("sinprod", 3.0, fr), diff=1.7)
        funcs = {"thermal source": ("sinprod", 3.0, fr), "thermal diffusion": 1.7}
    elif mode == "transient-arrays":
        A = np.array([[0.4358665215, 0, 0], [0.2820667392, 0.4358665215, 0], [1.208496649, -0.644363171, 0.4358665215]])
        bb = np.array([1.208496649, -0.644363171, 0.4358665215])
        tr = dict(u_prev=rng.uniform(-1, 1, (nd, 2)), u_stage=rng.uniform(-1, 1, (nd, 3)), stage=2, butcher_A=A,
                  butcher_b=bb, bdf=np.array([1.5, -2.0, 0.5]), dt=0.013)
        kap, src = rng.uniform(0.5, 2.0, (E, nq)), rng.uniform(-1, 1, (E, nq))
        okw = dict(source=("array", src), diff_ip=kap, rho=1.3, cp=0.7, transient=tr)
        funcs = {"thermal source": t(src), "thermal diffusion": t(kap), "density": 1.3, "specific heat": 0.7}
        kw = dict(u_prev=t(tr["u_prev"]), u_stage=t(tr["u_stage"]))
    else:
        expr = "1.0+0.5*x*y" if dim == 2 else "1.0+0.5*x*y+0.25*z"
        kap = 1.0 + 0.5 * pb["ip"][..., 0] * pb["ip"][..., 1] + (0.25 * pb["ip"][..., 2] if dim == 3 else 0.0)
        okw = dict(source=("const", 0.8), diff_ip=kap)
        funcs = {"thermal source": 0.8, "thermal diffusion": expr}
    ref = oracle.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, fixed=fixed, pb=pb, **okw)
    blk = make_block(m, dim, order, qdeg, fixed=fixed, graph=(ref["rowptr"], ref["colind"]))
    for k, v in funcs.items():
        blk.set_function(k, v)
    if tr is not None:
        blk.set_time_integration(True, 2, 3, 2, tr["dt"], tr["butcher_A"], tr["butcher_b"], tr["bdf"])
    res = torch.full((nd,), 7.0, dtype=torch.float64, device="cuda")
    vals = torch.full((len(ref["colind"]),), -3.0, dtype=torch.float64, device="cuda")
    blk.assemble_jacres(t(u), res, vals, overwrite=True, path=mrhyde_amd.PATH_ROW_OWNER, **kw)
    torch.cuda.synchronize()
    assert blk.info("row_owner_kind") == 2
    assert crs_err(vals.cpu().numpy(), ref) < RTOL
    assert rel_err(res.cpu().numpy(), ref["res"]) < RTOL
    # run twice: bit-identical up to the order of LDS adds within one entry -- in practice identical to roundoff
    res2, vals2 = torch.zeros_like(res), torch.zeros_like(vals)
    blk.assemble_jacres(t(u), res2, vals2, overwrite=True, path=mrhyde_amd.PATH_ROW_OWNER, **kw)
    torch.cuda.synchronize()
    assert rel_err(vals2.cpu().numpy(), vals.cpu().numpy()) < 1e-14 and rel_err(res2.cpu().numpy(), res.cpu().numpy()) < 1e-14


def test_error_behaviour_on_device():
    torch = _torch()
    import mrhyde_amd
    m = mrhyde_amd.mesh_structured(2, 1, (3, 3))
    blk = mrhyde_amd.Block(2, 1)
    u = torch.zeros(m["ndof"], dtype=torch.float64, device="cuda")
    with pytest.raises(mrhyde_amd.MhaError) as ei:
        blk.assemble_jacres(u, u.clone(), u.clone())
    assert ei.value.code == 2  # no mesh yet
    blk.set_mesh(m["nodes"], m["lids"], m["offsets"], m["ndof"])
    with pytest.raises(mrhyde_amd.MhaError) as ei:
        blk.assemble_jacres(u, u.clone(), u.clone())
    assert ei.value.code == 2  # no graph yet
    bad = m["lids"].copy()
    bad[0, 0] = m["ndof"]
    with pytest.raises(mrhyde_amd.MhaError) as ei:
        blk.set_mesh(m["nodes"], bad, m["offsets"], m["ndof"])
    assert ei.value.code == 1


def test_thermal_transient_gold_end_to_end(oracle):
    """regression/thermal/2D_verification_transient with the GPU assembling every Newton step: all 20 printed
    L2 errors (backward Euler, BDF-1).  Pins the transient seeding of the HIP path against the reference's gold."""
    torch = _torch()
    import scipy.sparse as sp
    import mrhyde_amd
    from test_oracle_golden import _transient_golds, run_transient_bwe
    A, b, bdf = np.array([[1.0]]), np.array([1.0]), np.array([1.0, -1.0])
    state = {}

    def assemble(m, pb, u, u_prev, t, dt):
        if "blk" not in state:
            blk = make_block(m, 2, 1, 2, fixed=m["boundary"])
            state["blk"], state["graph"] = blk, blk.get_graph()
            state["res"] = torch.zeros(m["ndof"], dtype=torch.float64, device="cuda")
            state["vals"] = torch.zeros(len(state["graph"][1]), dtype=torch.float64, device="cuda")
        blk = state["blk"]
        rowptr, colind = state["graph"]
        amp = 8 * np.pi ** 2 * np.sin(2 * np.pi * t) + 2 * np.pi * np.cos(2 * np.pi * t)
        blk.set_function("thermal source", ("sinprod", amp, [2 * np.pi] * 2))
        blk.set_time_integration(True, 1, 1, 0, dt, A, b, bdf)
        tt = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")
        blk.assemble_jacres(tt(u), state["res"], state["vals"], u_prev=tt(u_prev[:, None]), u_stage=tt(u[:, None]),
                            overwrite=True)
        blk.apply_dbc_diag(state["vals"])
        torch.cuda.synchronize()
        J = sp.csr_matrix((state["vals"].cpu().numpy(), colind, rowptr), shape=(m["ndof"],) * 2)
        return J, state["res"].cpu().numpy()

    errs = run_transient_bwe(assemble, oracle)
    assert state["blk"].info("last_path") == mrhyde_amd.PATH_ROW_OWNER
    for (t, e), (tg, eg) in zip(errs, _transient_golds()):
        if tg > 0.0:
            assert "%.6g" % e == "%.6g" % eg, (t, e, eg)


def _deck(name):
    import yaml
    return yaml.safe_load(open(os.path.join(GOLD, name)))["ANONYMOUS"]


def test_deck_strings_through_the_device_interpreter(oracle):
    """MHA_FUNC_EXPRESSION: the `Functions:` strings of the mirrored decks run verbatim on the device.
    (1) 2D_verification: 'thermal source' string -> gold L2 error; (2) expression source/diffusion on a perturbed mesh
    against the oracle's independent evaluator, with the time variable t."""
    torch = _torch()
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    import mrhyde_amd
    src = _deck("thermal_2D_verification.input.yaml")["Functions"]["thermal source"]
    assert "sin(2*pi*x)" in src
    m = oracle.mesh_structured(2, 1, (40, 40))
    pb = oracle.physical_basis(2, 1, 2, m["nodes"])
    blk = make_block(m, 2, 1, 2, fixed=m["boundary"])
    blk.set_function("thermal source", src)
    rowptr, colind = blk.get_graph()
    u = np.zeros(m["ndof"])
    res = torch.empty(m["ndof"], dtype=torch.float64, device="cuda")
    vals = torch.empty(len(colind), dtype=torch.float64, device="cuda")
    blk.assemble_jacres(torch.tensor(u, device="cuda"), res, vals, overwrite=True)
    blk.apply_dbc_diag(vals)
    torch.cuda.synchronize()
    J = sp.csr_matrix((vals.cpu().numpy(), colind, rowptr), shape=(m["ndof"],) * 2)
    u = spla.spsolve(J.tocsc(), res.cpu().numpy())
    err = oracle.l2_error_sinprod(2, 1, 2, m["lids"], m["offsets"], pb, u, [2 * np.pi] * 2)
    assert "%.6g" % err == "0.00102776"
    # (2) time-dependent expression + variable diffusion on a perturbed Q2 hex mesh, general path, vs the oracle
    mm = oracle.mesh_multi(3, (3, 2, 2), [oracle.HGRAD], [2])
    rng = np.random.default_rng(81)
    v = mm["verts"].copy()
    inner = np.all((v > 1e-12) & (v < 1 - 1e-12), axis=1)
    v[inner] += 0.04 * rng.uniform(-1, 1, (inner.sum(), 3))
    mm["verts"], mm["nodes"] = v, np.ascontiguousarray(v[mm["cell2vert"]])
    uu = rng.uniform(-1, 1, mm["ndof"])
    fsrc = "(8*pi^2*sin(2*pi*t) + 2*pi*cos(2*pi*t))*sin(2*pi*x)*sin(2*pi*y)*exp(-z)"
    fdif = "1.5 + 0.5*cos(pi*x)*(y<0.5) - z/4"
    t = 0.37
    ref = oracle.assemble_block(mm, oracle.PHYS_THERMAL, 4, uu,
                                funcs={"thermal source": ("expr", fsrc, t), "thermal diffusion": ("expr", fdif, t)})
    b2 = make_block(mm, 3, 2, 4, graph=(ref["rowptr"], ref["colind"]))
    b2.set_function("thermal source", fsrc)
    b2.set_function("thermal diffusion", fdif)
    b2.set_time(t)
    for path in (mrhyde_amd.PATH_AUTO, mrhyde_amd.PATH_ELEMENT_ATOMIC, mrhyde_amd.PATH_POINT_ENGINE):
        r2 = torch.zeros(mm["ndof"], dtype=torch.float64, device="cuda")
        v2 = torch.zeros(len(ref["colind"]), dtype=torch.float64, device="cuda")
        b2.assemble_jacres(torch.tensor(uu, device="cuda"), r2, v2, overwrite=True, path=path)
        torch.cuda.synchronize()
        assert crs_err(v2.cpu().numpy(), ref) < RTOL and rel_err(r2.cpu().numpy(), ref["res"]) < RTOL
    # an identifier nothing defines (not a coordinate, a solution field of the block or a function of the deck) is refused
    # when the functions are next evaluated -- the strings may name functions that are defined later
    b2.set_function("thermal source", "2*e + grud")
    with pytest.raises(mrhyde_amd.MhaError) as e:
        b2.assemble_jacres(torch.tensor(uu, device="cuda"), r2, v2, overwrite=True)
    assert e.value.code == 1 and "grud" in str(e.value)


@pytest.mark.parametrize("dim,order,ncell", [(2, 2, (4, 3)), (3, 1, (3, 3, 2)), (3, 2, (2, 2, 2))])
def test_nonlinear_diffusion_deck_string(oracle, dim, order, ncell):
    """Deck strings that read the solution fields: 'thermal diffusion' = "1+e*e" (a nonlinear diffusion), a source that
    depends on e and on grad(e), and a function that names another function of the deck.  The reference evaluates them
    with FunctionManager<AD> (functionManager.cpp:95-860), so res.dx picks up d kappa / d e; here the point engine
    interprets the program on Dual numbers.  Residual and Jacobian against the numpy restatement
    (oracle_lib.assemble_thermal_fields) to 1e-12, per entry relative to the row's largest entry."""
    torch = _torch()
    import mrhyde_amd
    qdeg = 2 * order
    m = oracle.mesh_structured(dim, order, ncell)
    rng = np.random.default_rng(71)
    v = m["verts"].copy()
    inner = np.all((v > 1e-12) & (v < 1 - 1e-12), axis=1)
    v[inner] += 0.04 * rng.uniform(-1, 1, (int(inner.sum()), dim))
    m["verts"], m["nodes"] = v, np.ascontiguousarray(v[m["cell2vert"]])
    u = rng.uniform(-1, 1, m["ndof"])
    fixed = m["boundary"]
    rowptr, colind = oracle.build_graph(m["ndof"], m["lids"])
    funcs = {"thermal source": "2*sin(pi*x)*y + 0.1*e", "thermal diffusion": "kappa0 + 0.2*grad(e)[x]^2"}
    ref = oracle.assemble_thermal_fields(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, funcs, fixed=fixed,
                                         rowptr=rowptr, colind=colind, functions={"kappa0": "1+e*e"})
    blk = make_block(m, dim, order, qdeg, fixed=fixed, graph=(rowptr, colind))
    blk.set_function("thermal diffusion", funcs["thermal diffusion"])   # names kappa0 before it is defined
    blk.set_function("kappa0", "1+e*e")
    blk.set_function("thermal source", funcs["thermal source"])
    ud = torch.tensor(u, device="cuda")
    for path in (mrhyde_amd.PATH_AUTO, mrhyde_amd.PATH_POINT_ENGINE):
        res = torch.zeros(m["ndof"], dtype=torch.float64, device="cuda")
        vals = torch.zeros(len(colind), dtype=torch.float64, device="cuda")
        blk.assemble_jacres(ud, res, vals, overwrite=(path == mrhyde_amd.PATH_AUTO), path=path)
        torch.cuda.synchronize()
        assert blk.info("last_path") in (mrhyde_amd.PATH_ROW_GATHER, mrhyde_amd.PATH_POINT_ENGINE)
        gv, gr = vals.cpu().numpy(), res.cpu().numpy()
        assert rel_err(gr, ref["res"]) < RTOL
        rows = np.repeat(np.arange(m["ndof"]), np.diff(rowptr))
        rowmax = np.zeros(m["ndof"])
        np.maximum.at(rowmax, rows, np.abs(ref["crs_vals"]))
        tol = RTOL * np.maximum(np.abs(ref["crs_vals"]), 1e-3 * rowmax[rows]) + 1e-300
        assert np.all(np.abs(gv - ref["crs_vals"]) <= tol), float((np.abs(gv - ref["crs_vals"]) / tol).max())
    # the row-owner kernels assume coefficients that do not depend on the solution: refused, not silently wrong
    with pytest.raises(mrhyde_amd.MhaError):
        blk.assemble_jacres(ud, res, vals, path=mrhyde_amd.PATH_ROW_OWNER)


def test_mixed_bcs_deck_strings(oracle):
    """2D_mixed_bcs with the deck's own Neumann strings (normals-free cos/sin products) evaluated on the device."""
    torch = _torch()
    import scipy.sparse as sp
    import mrhyde_amd
    from test_oracle_golden import _gold_l2, solve_mixed_bcs
    deck = _deck("thermal_2D_mixed_bcs.input.yaml")
    neu = deck["Physics"]["Neumann conditions"]["e"]
    state = {}

    def assemble(m, pb, u, groups):
        if "blk" not in state:
            blk = make_block(m, 2, 1, 2, fixed=m["fixed"])
            blk.set_function("thermal source", deck["Functions"]["thermal source"])
            for name, (be, bs, _) in zip(("top", "bottom"), groups):
                blk.add_boundary_group(name, mrhyde_amd.BC_NEUMANN, be, bs)
                blk.set_function("Neumann e " + name, neu[name])
            state["blk"], state["graph"] = blk, blk.get_graph()
        blk = state["blk"]
        rowptr, colind = state["graph"]
        ud = torch.tensor(u, device="cuda")
        res = torch.empty(m["ndof"], dtype=torch.float64, device="cuda")
        vals = torch.empty(len(colind), dtype=torch.float64, device="cuda")
        blk.assemble_jacres(ud, res, vals, overwrite=True)
        blk.assemble_boundary(ud, res, vals)
        blk.apply_dbc_diag(vals)
        torch.cuda.synchronize()
        return sp.csr_matrix((vals.cpu().numpy(), colind, rowptr), shape=(m["ndof"],) * 2), res.cpu().numpy()

    err = solve_mixed_bcs(assemble, oracle)
    assert "%.6g" % err == "%.6g" % _gold_l2("thermal_2D_mixed_bcs.gold") == "0.00102733"


@pytest.mark.parametrize("transient", [False, True])
def test_deterministic_mode_is_bit_reproducible(oracle, transient):
    """MHA_ASSEMBLE_DETERMINISTIC: two assemblies of the same state give bit-identical residual and Jacobian (fixed-order
    register sums for the CRS rows, pair-ordered sums by the row's owner for the residual: no atomics anywhere), and the
    result is the oracle's to 1e-12.  A second, different state in between guards against a cached result.  Where the
    combination of kernels does not exist (non-affine elements) the flag is refused."""
    torch = _torch()
    import mrhyde_amd
    dim, order, qdeg, ncell = 3, 2, 4, (10, 9, 7)
    m = affine_mesh(oracle, dim, order, ncell)
    rng = np.random.default_rng(41)
    nd = m["ndof"]
    u = rng.uniform(-1, 1, nd)
    fixed = m["boundary"]
    tr, kw, okw = None, {}, {}
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")
    if transient:
        A, bb, bdf = np.array([[0.5, 0.0], [0.3, 0.7]]), np.array([0.4, 0.6]), np.array([1.5, -2.0, 0.5])
        tr = dict(u_prev=rng.uniform(-1, 1, (nd, 2)), u_stage=rng.uniform(-1, 1, (nd, 2)), stage=1, butcher_A=A, butcher_b=bb,
                  bdf=bdf, dt=0.05)
        okw = dict(transient=tr, rho=1.3, cp=0.7)
        kw = dict(u_prev=t(tr["u_prev"]), u_stage=t(tr["u_stage"]))
    ref = oracle.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, fixed=fixed, diff=1.7,
                                  source=("sinprod", 3.0, [2.0, 1.0, 1.5]), **okw)
    blk = make_block(m, dim, order, qdeg, fixed=fixed, graph=(ref["rowptr"], ref["colind"]))
    blk.set_function("thermal source", ("sinprod", 3.0, [2.0, 1.0, 1.5]))
    blk.set_function("thermal diffusion", 1.7)
    if transient:
        blk.set_function("density", 1.3)
        blk.set_function("specific heat", 0.7)
        blk.set_time_integration(True, 2, 2, 1, tr["dt"], tr["butcher_A"], tr["butcher_b"], tr["bdf"])
    runs = []
    for k in range(3):
        uu = u if k != 1 else rng.uniform(-1, 1, nd)
        res = torch.full((nd,), float(k), dtype=torch.float64, device="cuda")
        vals = torch.full((len(ref["colind"]),), -float(k), dtype=torch.float64, device="cuda")
        blk.assemble_jacres(t(uu), res, vals, overwrite=True, deterministic=True, **kw)
        torch.cuda.synchronize()
        runs.append((res.clone(), vals.clone()))
    assert blk.info("last_path") == mrhyde_amd.PATH_ROW_OWNER and blk.info("row_owner_kind") == 1
    assert torch.equal(runs[0][0], runs[2][0]) and torch.equal(runs[0][1], runs[2][1])
    assert not torch.equal(runs[0][0], runs[1][0])
    assert crs_err(runs[0][1].cpu().numpy(), ref) < RTOL
    assert rel_err(runs[0][0].cpu().numpy(), ref["res"]) < RTOL
    # residual-only, accumulate on top
    res = runs[0][0].clone()
    blk.assemble_jacres(t(u), res, None, compute_jacobian=False, deterministic=True, **kw)
    torch.cuda.synchronize()
    assert rel_err(res.cpu().numpy(), 2 * ref["res"]) < RTOL
    # the default mode agrees to rounding
    res2, vals2 = torch.zeros_like(res), torch.zeros_like(runs[0][1])
    blk.assemble_jacres(t(u), res2, vals2, overwrite=True, **kw)
    torch.cuda.synchronize()
    assert rel_err(res2.cpu().numpy(), runs[0][0].cpu().numpy()) < 1e-13
    mp = perturbed(oracle, dim, order, (3, 3, 3), seed=2)
    blk2 = make_block(mp, dim, order, qdeg)
    with pytest.raises(mrhyde_amd.MhaError):
        blk2.assemble_jacres(t(np.zeros(mp["ndof"])), torch.zeros(mp["ndof"], dtype=torch.float64, device="cuda"),
                             torch.zeros(blk2.get_graph()[1].shape[0], dtype=torch.float64, device="cuda"), deterministic=True)


@pytest.mark.parametrize("dim,order,qdeg,ncell", [(3, 2, 4, (8, 8, 4)), (3, 1, 2, (16, 8, 8)), (2, 2, 4, (16, 16))])
@pytest.mark.parametrize("transient", [False, True])
def test_geometry_database_mode_is_bit_identical(oracle, monkeypatch, dim, order, qdeg, ncell, transient):
    """One geometry shape in the block (a uniform mesh whose spacing is exactly representable): the Jacobian kernel runs
    on one representative row block per pattern and the representative's runs of CRS rows are replicated
    (`jacobian_database_mode`); MHA_BP_DATABASE=0 runs the full kernel.  Same inputs through the same instructions:
    the two must agree BIT FOR BIT, and with the oracle to the usual tolerance; fixed rows, overwrite on garbage."""
    torch = _torch()
    import mrhyde_amd
    monkeypatch.delenv("MHA_K1", raising=False)
    monkeypatch.delenv("MHA_K2", raising=False)
    m = affine_mesh(oracle, dim, order, ncell, shear=False)
    rng = np.random.default_rng(5)
    nd = m["ndof"]
    u = rng.uniform(-1, 1, nd)
    fixed = m["boundary"]
    freq = [1.3, 0.7, 2.1][:dim]
    kw, okw = {}, {}
    if transient:
        nsteps, nstages, stage = 2, 2, 1
        A = np.array([[0.2928932188, 0.0], [0.7071067812, 0.2928932188]])
        bb = np.array([0.7071067812, 0.2928932188])
        bdf = np.array([1.5, -2.0, 0.5])
        tr = dict(u_prev=rng.uniform(-1, 1, (nd, nsteps)), u_stage=rng.uniform(-1, 1, (nd, nstages)), stage=stage,
                  butcher_A=A, butcher_b=bb, bdf=bdf, dt=0.02)
        okw = dict(transient=tr, rho=1.3, cp=0.7)
    ref = oracle.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, fixed=fixed,
                                  source=("sinprod", 3.0, freq), diff=1.7, **okw)
    got = {}
    for db in (True, False):
        if db:
            monkeypatch.delenv("MHA_BP_DATABASE", raising=False)
        else:
            monkeypatch.setenv("MHA_BP_DATABASE", "0")
        blk = make_block(m, dim, order, qdeg, fixed=fixed, graph=(ref["rowptr"], ref["colind"]))
        blk.set_function("thermal source", ("sinprod", 3.0, freq))
        blk.set_function("thermal diffusion", 1.7)
        t = lambda a: torch.tensor(a, device="cuda")
        if transient:
            blk.set_function("density", 1.3)
            blk.set_function("specific heat", 0.7)
            blk.set_time_integration(True, nsteps, nstages, stage, 0.02, A, bb, bdf)
            kw = dict(u_prev=t(tr["u_prev"]), u_stage=t(tr["u_stage"]))
        res = torch.full((nd,), 7.0, dtype=torch.float64, device="cuda")
        vals = torch.full((len(ref["colind"]),), -3.0, dtype=torch.float64, device="cuda")
        blk.assemble_jacres(t(u), res, vals, path=mrhyde_amd.PATH_ROW_OWNER, overwrite=True, **kw)
        torch.cuda.synchronize()
        assert blk.info("block_patterns") > 0 and blk.info("affine_shapes") == 1
        assert blk.info("jacobian_database_mode") == (1 if db else 0)
        got[db] = vals.cpu().numpy().copy()
        assert crs_err(got[db], ref) < RTOL and rel_err(res.cpu().numpy(), ref["res"]) < RTOL
        if db:  # accumulating assemblies take the full kernel (a copy cannot add)
            blk.assemble_jacres(t(u), res, vals, path=mrhyde_amd.PATH_ROW_OWNER, **kw)
            torch.cuda.synchronize()
            assert blk.info("jacobian_database_mode") == 0
            assert crs_err(vals.cpu().numpy(), ref, 2.0) < RTOL
            # a value array that does not start on a 128-byte line: the full kernel again, same bits
            big = torch.full((len(ref["colind"]) + 1,), -3.0, dtype=torch.float64, device="cuda")
            blk.assemble_jacres(t(u), res, big[1:], path=mrhyde_amd.PATH_ROW_OWNER, overwrite=True, **kw)
            torch.cuda.synchronize()
            assert blk.info("jacobian_database_mode") == 0 and np.array_equal(big[1:].cpu().numpy(), got[True])
    assert np.array_equal(got[True], got[False])
