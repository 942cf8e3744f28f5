"""GPU parity tests of the multi-variable point engine: porousMixed (HVOL + HDIV) and
navierstokes (Q2/Q1, SUPG/PSPG) through the C ABI against the CPU oracle's AD-array restatement, thermal through the same
engine as a cross-check, and the reference golds (porous Mixed_3d, navierstokes channel) end to end with GPU assembly."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu
RTOL = 1e-12


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def crs_err(a, ref, factor=1.0):
    """Per-entry relative error of a CRS value array with a cancellation floor of a thousandth of the row's largest
    entry (tests/test_thermal_gpu.py: crs_err), beside the array-relative measure."""
    a, b = np.asarray(a), factor * np.asarray(ref["crs_vals"])
    rowptr = np.asarray(ref["rowptr"])
    rows = np.repeat(np.arange(len(rowptr) - 1), np.diff(rowptr))
    rowmax = np.zeros(len(rowptr) - 1)
    np.maximum.at(rowmax, rows, np.abs(b))
    den = np.maximum(np.maximum(np.abs(b), 1e-3 * rowmax[rows]), 1e-300)
    return max(float((np.abs(a - b) / den).max()), rel_err(a, b))


def warp(m):
    """Smooth warp of every vertex: non-affine elements, non-constant Jacobians (exercises the Piola transforms)."""
    v = m["verts"].copy()
    dim = v.shape[1]
    w = v.copy()
    w[:, 0] += 0.06 * np.sin(1.3 * v[:, 1] + 0.4) + (0.04 * v[:, 2] ** 2 if dim == 3 else 0.0)
    w[:, 1] += 0.05 * np.cos(1.1 * v[:, 0]) * (1 + 0.5 * v[:, 1])
    if dim == 3:
        w[:, 2] += 0.05 * v[:, 0] * v[:, 1] + 0.03 * np.sin(2.0 * v[:, 2])
    m["verts"] = w
    m["nodes"] = np.ascontiguousarray(w[m["cell2vert"]])
    return m


def make_block(m, physics, qdeg, fixed=None, graph=None):
    import mrhyde_amd
    blk = mrhyde_amd.Block(m["dim"], quadrature=qdeg, physics=physics,
                           variables=list(zip(m["types"].tolist(), m["orders"].tolist())))
    blk.set_mesh(m["nodes"], m["lids"], m["offsets"], m["ndof"], fixed)
    blk.set_orientation(m["orient"])
    blk.set_graph(*graph) if graph is not None else blk.set_graph()
    return blk


def transient_state(rng, ndof, u):
    A, b, bdf = np.array([[0.5, 0.0], [0.3, 0.7]]), np.array([0.4, 0.6]), np.array([1.5, -2.0, 0.5])
    return dict(u_prev=rng.uniform(-1, 1, (ndof, 2)), u_stage=rng.uniform(-1, 1, (ndof, 2)), stage=1, butcher_A=A,
                butcher_b=b, bdf=bdf, dt=0.05)


def run_gpu(blk, m, u, tr, nnz, local=False):
    torch = _torch()
    kw = {}
    if tr is not None:
        blk.set_time_integration(True, 2, 2, 1, tr["dt"], tr["butcher_A"], tr["butcher_b"], tr["bdf"])
        kw = dict(u_prev=torch.tensor(tr["u_prev"], device="cuda"), u_stage=torch.tensor(tr["u_stage"], device="cuda"))
    ud = torch.tensor(u, device="cuda")
    res = torch.zeros(m["ndof"], dtype=torch.float64, device="cuda")
    vals = torch.zeros(nnz, dtype=torch.float64, device="cuda")
    import mrhyde_amd
    blk.assemble_jacres(ud, res, vals, path=mrhyde_amd.PATH_POINT_ENGINE, **kw)
    out = dict(res=res.cpu().numpy(), crs_vals=vals.cpu().numpy())
    if local:
        E, n = m["lids"].shape
        lJ = torch.zeros((E, n, n), dtype=torch.float64, device="cuda")
        lr = torch.zeros((E, n), dtype=torch.float64, device="cuda")
        blk.compute_local_jacres(ud, lJ, lr, **kw)
        out["local_J"], out["local_res"] = lJ.cpu().numpy(), lr.cpu().numpy()
        # the two-step path (updateJac -> scatterJac), workset by workset, must give the same global arrays
        res2, vals2 = torch.zeros_like(res), torch.zeros_like(vals)
        blk.assemble_jacres(ud, res2, vals2, path=mrhyde_amd.PATH_LOCAL_THEN_SCATTER, **kw)
        out["res2"], out["crs_vals2"] = res2.cpu().numpy(), vals2.cpu().numpy()
        # row-gather path (no global atomics): overwrite semantics on garbage, then accumulate on top
        res3, vals3 = torch.full_like(res, 7.0), torch.full_like(vals, -3.0)
        blk.assemble_jacres(ud, res3, vals3, path=mrhyde_amd.PATH_ROW_GATHER, overwrite=True, **kw)
        r3, v3 = res3.cpu().numpy().copy(), vals3.cpu().numpy().copy()
        blk.assemble_jacres(ud, res3, vals3, path=mrhyde_amd.PATH_ROW_GATHER, **kw)
        assert rel_err(res3.cpu().numpy(), 2 * r3) < 1e-14 and rel_err(vals3.cpu().numpy(), 2 * v3) < 1e-14
        assert rel_err(r3, out["res"]) < RTOL and rel_err(v3, out["crs_vals"]) < RTOL
        # residual-only pass leaves the matrix alone
        blk.assemble_jacres(ud, res3, vals3, path=mrhyde_amd.PATH_ROW_GATHER, compute_jacobian=False, overwrite=True, **kw)
        assert rel_err(res3.cpu().numpy(), r3) < 1e-14 and rel_err(vals3.cpu().numpy(), 2 * v3) < 1e-14
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("dim,order,qdeg,ncell", [(2, 1, 2, (5, 4)), (2, 4, 8, (3, 2)), (3, 2, 4, (3, 2, 2))])
@pytest.mark.parametrize("transient", [False, True])
def test_thermal_through_point_engine(oracle, dim, order, qdeg, ncell, transient):
    rng = np.random.default_rng(31)
    m = warp(oracle.mesh_multi(dim, ncell, [oracle.HGRAD], [order]))
    u = rng.uniform(-1, 1, m["ndof"])
    fixed = (m["side_mask"] != 0).astype(np.uint8)
    tr = transient_state(rng, m["ndof"], u) if transient else None
    funcs = {"thermal source": ("sinprod", 3.0, [2.0, 1.0, 1.5][:dim]), "thermal diffusion": 1.7, "density": 0.8,
             "specific heat": 1.4}
    ref = oracle.assemble_block(m, oracle.PHYS_THERMAL, qdeg, u, funcs=funcs, fixed=fixed, transient=tr, want_local=True)
    blk = make_block(m, "thermal", qdeg, fixed=fixed, graph=(ref["rowptr"], ref["colind"]))
    for k, v in funcs.items():
        blk.set_function(k, v)
    out = run_gpu(blk, m, u, tr, len(ref["colind"]), local=True)
    for k in ("crs_vals", "res", "local_J", "local_res"):
        assert rel_err(out[k], ref[k]) < RTOL, k
    assert crs_err(out["crs_vals2"], ref) < RTOL and rel_err(out["res2"], ref["res"]) < RTOL


@pytest.mark.parametrize("dim,ncell", [(2, (5, 4)), (3, (3, 2, 3))])
def test_porous_mixed_matches_oracle(oracle, dim, ncell):
    rng = np.random.default_rng(32)
    m = warp(oracle.mesh_multi(dim, ncell, [oracle.HVOL, oracle.HDIV], [0, 1]))
    # flip a few orientation signs consistently per global face dof: the kernel must take the caller's signs verbatim
    flip = rng.uniform(size=m["ndof"]) < 0.3
    u0 = m["varptr"][1]
    for e in range(m["nelem"]):
        for f in range(2 * dim):
            if flip[m["lids"][e, m["offsets"][u0 + f]]]:
                m["orient"][e, u0 + f] *= -1
    u = rng.uniform(-1, 1, m["ndof"])
    E, nq = m["nelem"], oracle.ref_sizes(dim, 1, 2)[1]
    src = rng.uniform(-2, 2, (E, nq))
    funcs = {"source": ("array", src), "Kinv_xx": 1.3, "Kinv_yy": 0.7, "Kinv_zz": 2.1, "total_mobility": 1.9}
    ref = oracle.assemble_block(m, oracle.PHYS_POROUS_MIXED, 2, u, funcs=funcs, want_local=True)
    torch = _torch()
    blk = make_block(m, "porousMixed", 2, graph=(ref["rowptr"], ref["colind"]))
    keep = torch.tensor(src, device="cuda")
    for k, v in funcs.items():
        blk.set_function(k, keep if k == "source" else v)
    out = run_gpu(blk, m, u, None, len(ref["colind"]), local=True)
    for k in ("crs_vals", "res", "local_J", "local_res"):
        assert rel_err(out[k], ref[k]) < RTOL, k
    assert crs_err(out["crs_vals2"], ref) < RTOL and rel_err(out["res2"], ref["res"]) < RTOL


@pytest.mark.parametrize("dim,ncell", [(2, (6, 5)), (3, (4, 3, 5))])
@pytest.mark.parametrize("transient", [False, True])
def test_porous_direct_form_equals_row_gather_and_oracle(oracle, monkeypatch, dim, ncell, transient):
    """porousMixed behind MHA_PATH_ROW_GATHER / AUTO: by default the element threads store their matrix entries straight
    into the CRS (two elements share one dof: only the diagonal of a face row has two contributors) and a finishing
    pass sums the residual and diagonal parts per row; MHA_POROUS_DIRECT=0 keeps dense element arrays + the row gather.
    Both against the oracle and each other: fixed rows, overwrite on garbage, accumulate, residual only."""
    torch = _torch()
    import mrhyde_amd
    rng = np.random.default_rng(61)
    m = warp(oracle.mesh_multi(dim, ncell, [oracle.HVOL, oracle.HDIV], [0, 1]))
    u = rng.uniform(-1, 1, m["ndof"])
    fixed = (rng.uniform(size=m["ndof"]) < 0.1).astype(np.uint8)
    funcs = {"source": ("sinprod", 2.0, [1.1, 0.7, 1.9][:dim]), "Kinv_xx": 1.3, "Kinv_yy": 0.7, "Kinv_zz": 2.1,
             "total_mobility": 1.9}
    tr = transient_state(rng, m["ndof"], u) if transient else None
    ref = oracle.assemble_block(m, oracle.PHYS_POROUS_MIXED, 2, u, funcs=funcs, transient=tr, fixed=fixed)
    results = {}
    for direct in (True, False):
        if direct:
            monkeypatch.delenv("MHA_POROUS_DIRECT", raising=False)
        else:
            monkeypatch.setenv("MHA_POROUS_DIRECT", "0")
        blk = make_block(m, "porousMixed", 2, fixed=fixed, graph=(ref["rowptr"], ref["colind"]))
        for k, v in funcs.items():
            blk.set_function(k, v)
        kw = {}
        if tr is not None:
            blk.set_time_integration(True, 2, 2, 1, tr["dt"], tr["butcher_A"], tr["butcher_b"], tr["bdf"])
            kw = dict(u_prev=torch.tensor(tr["u_prev"], device="cuda"), u_stage=torch.tensor(tr["u_stage"], device="cuda"))
        ud = torch.tensor(u, device="cuda")
        res = torch.full((m["ndof"],), 7.0, dtype=torch.float64, device="cuda")
        vals = torch.full((len(ref["colind"]),), -3.0, dtype=torch.float64, device="cuda")
        blk.assemble_jacres(ud, res, vals, overwrite=True, **kw)  # AUTO
        torch.cuda.synchronize()
        assert blk.info("last_path") == mrhyde_amd.PATH_ROW_GATHER and blk.info("porous_direct") == (1 if direct else 0)
        r1, v1 = res.cpu().numpy().copy(), vals.cpu().numpy().copy()
        assert rel_err(r1, ref["res"]) < RTOL and crs_err(v1, ref) < RTOL
        for r in np.flatnonzero(fixed)[:30]:
            assert r1[r] == 0.0 and np.all(v1[ref["rowptr"][r]:ref["rowptr"][r + 1]] == 0.0)
        blk.assemble_jacres(ud, res, vals, **kw)  # accumulate on top
        assert rel_err(res.cpu().numpy(), 2 * r1) < 1e-14 and rel_err(vals.cpu().numpy(), 2 * v1) < 1e-14
        blk.assemble_jacres(ud, res, vals, compute_jacobian=False, overwrite=True, **kw)  # residual only
        assert rel_err(res.cpu().numpy(), r1) < 1e-14 and rel_err(vals.cpu().numpy(), 2 * v1) < 1e-14
        results[direct] = (r1, v1)
    assert rel_err(results[True][0], results[False][0]) < 1e-13 and rel_err(results[True][1], results[False][1]) < 1e-13


@pytest.mark.parametrize("dim,ncell", [(3, (64, 2, 4)), (2, (128, 4))])
@pytest.mark.parametrize("transient", [False, True])
def test_porous_database_mode_is_bit_identical(oracle, monkeypatch, dim, ncell, transient):
    """porousMixed on a UNIFORM block with constant permeability / mobility: every element matrix is the same, rows of
    the same class (incident local dofs + column slots) are equal; the direct kernel stores the entries of a few
    representative rows per class and replicate_runs_kernel fills the rest (`porous_direct` = 2).  Bit for bit the
    matrix of the plain direct form (MHA_POROUS_DATABASE=0: `porous_direct` = 1), and the oracle's to tolerance; the
    residual likewise; fixed rows, overwrite on garbage.  (Long x-lines: a class needs a run of >= 2 (ceil(128 / len) + 2)
    consecutive rows to have representatives.)"""
    torch = _torch()
    import mrhyde_amd
    rng = np.random.default_rng(67)
    m = oracle.mesh_multi(dim, ncell, [oracle.HVOL, oracle.HDIV], [0, 1])  # unit-spaced, not warped: exactly uniform
    u = rng.uniform(-1, 1, m["ndof"])
    fixed = (rng.uniform(size=m["ndof"]) < 0.02).astype(np.uint8)
    funcs = {"source": ("sinprod", 2.0, [0.11, 0.7, 1.9][:dim]), "Kinv_xx": 1.3, "Kinv_yy": 0.7, "Kinv_zz": 2.1,
             "total_mobility": 1.9}
    tr = transient_state(rng, m["ndof"], u) if transient else None
    ref = oracle.assemble_block(m, oracle.PHYS_POROUS_MIXED, 2, u, funcs=funcs, transient=tr, fixed=fixed)
    got = {}
    for db in (True, False):
        if db:
            monkeypatch.delenv("MHA_POROUS_DATABASE", raising=False)
        else:
            monkeypatch.setenv("MHA_POROUS_DATABASE", "0")
        blk = make_block(m, "porousMixed", 2, fixed=fixed, graph=(ref["rowptr"], ref["colind"]))
        for k, v in funcs.items():
            blk.set_function(k, v)
        kw = {}
        if tr is not None:
            blk.set_time_integration(True, 2, 2, 1, tr["dt"], tr["butcher_A"], tr["butcher_b"], tr["bdf"])
            kw = dict(u_prev=torch.tensor(tr["u_prev"], device="cuda"), u_stage=torch.tensor(tr["u_stage"], device="cuda"))
        ud = torch.tensor(u, device="cuda")
        res = torch.full((m["ndof"],), 7.0, dtype=torch.float64, device="cuda")
        vals = torch.full((len(ref["colind"]),), -3.0, dtype=torch.float64, device="cuda")
        blk.assemble_jacres(ud, res, vals, overwrite=True, **kw)
        torch.cuda.synchronize()
        assert blk.info("porous_direct") == (2 if db else 1)
        got[db] = (res.cpu().numpy().copy(), vals.cpu().numpy().copy())
        assert rel_err(got[db][0], ref["res"]) < RTOL and crs_err(got[db][1], ref) < RTOL
        for r in np.flatnonzero(fixed)[:30]:
            assert got[db][0][r] == 0.0 and np.all(got[db][1][ref["rowptr"][r]:ref["rowptr"][r + 1]] == 0.0)
        if db:  # an accumulating assembly cannot replicate: plain direct form
            blk.assemble_jacres(ud, res, vals, **kw)
            torch.cuda.synchronize()
            assert blk.info("porous_direct") == 1
            assert rel_err(vals.cpu().numpy(), 2 * got[db][1]) < 1e-14
    # the matrix: bit for bit; the residual comes from the lean (residual-only) build of the same loop in database mode,
    # whose arithmetic the compiler schedules differently: equal to round-off
    assert np.array_equal(got[True][1], got[False][1]) and rel_err(got[True][0], got[False][0]) < 1e-14


@pytest.mark.parametrize("dim,ncell,orders", [(2, (4, 3), (1, 1)), (2, (3, 2), (2, 1)), (3, (2, 2, 2), (2, 1)),
                                             (3, (2, 3, 2), (1, 1))])
@pytest.mark.parametrize("mode", ["plain", "supg+pspg transient", "fix_uz"])
def test_navierstokes_matches_oracle(oracle, dim, ncell, orders, mode):
    rng = np.random.default_rng(33)
    H = oracle.HGRAD
    m = warp(oracle.mesh_multi(dim, ncell, [H] * (dim + 1), [orders[0], orders[1]] + [orders[0]] * (dim - 1)))
    u = rng.uniform(-1, 1, m["ndof"])
    fixed = (((m["side_mask"] & 0b1100) != 0) & (m["dof_var"] != 1)).astype(np.uint8)
    stab = mode.startswith("supg")
    tr = transient_state(rng, m["ndof"], u) if stab else None
    params = [1, 1, 0] if stab else ([0, 0, 1] if mode == "fix_uz" else [0, 0, 0])
    funcs = {"source ux": 0.3, "source uy": ("sinprod", 1.0, [1.0, 2.0, 0.5][:dim]), "source uz": -0.2,
             "viscosity": 0.05, "density": 1.3}
    qdeg = 2 * orders[0]
    ref = oracle.assemble_block(m, oracle.PHYS_NAVIERSTOKES, qdeg, u, funcs=funcs, params=params, fixed=fixed,
                                transient=tr, want_local=True)
    blk = make_block(m, "navierstokes", qdeg, fixed=fixed, graph=(ref["rowptr"], ref["colind"]))
    for k, v in funcs.items():
        blk.set_function(k, v)
    for name, val in zip(("useSUPG", "usePSPG", "fix_uz_offsets"), params):
        blk.set_physics_parameter(name, val)
    out = run_gpu(blk, m, u, tr, len(ref["colind"]), local=True)
    for k in ("crs_vals", "res", "local_J", "local_res"):
        assert rel_err(out[k], ref[k]) < RTOL, k
    assert crs_err(out["crs_vals2"], ref) < RTOL and rel_err(out["res2"], ref["res"]) < RTOL
    if dim == 3 and mode == "plain":  # the reference's uz-offset quirk: uz rows stay empty
        assert np.all(out["res"][m["dof_var"] == 3] == 0.0)


def test_navierstokes_q_streamed_engine_on_a_ragged_block(oracle, monkeypatch):
    """The 89-dof navierstokes element through the row-gather path = the q-streamed form of the point engine's
    Jacobian phase (owner waves, P blocks two points ahead, residual rows on the spare wave, element matrix put
    together in LDS).  295 elements on 256 persistent workgroups: ranges of two elements, a last range of one (its
    second pass recomputes with outputs off), workgroups without elements.  Against the oracle and, to round-off,
    against the panel-by-panel form of the same kernel (MHA_ENGINE_STOP=32), transient with SUPG + PSPG."""
    torch = _torch()
    import mrhyde_amd
    rng = np.random.default_rng(35)
    H = oracle.HGRAD
    m = warp(oracle.mesh_multi(3, (59, 5, 1), [H] * 4, [2, 1, 2, 2]))
    assert m["lids"].shape == (295, 89)
    u = rng.uniform(-1, 1, m["ndof"])
    tr = transient_state(rng, m["ndof"], u)
    funcs = {"source ux": 0.3, "source uy": ("sinprod", 1.0, [1.0, 2.0, 0.5]), "source uz": -0.2, "viscosity": 0.05,
             "density": 1.3}
    params = [1, 1, 1]
    ref = oracle.assemble_block(m, oracle.PHYS_NAVIERSTOKES, 4, u, funcs=funcs, params=params, transient=tr)
    blk = make_block(m, "navierstokes", 4, graph=(ref["rowptr"], ref["colind"]))
    for k, v in funcs.items():
        blk.set_function(k, v)
    for name, val in zip(("useSUPG", "usePSPG", "fix_uz_offsets"), params):
        blk.set_physics_parameter(name, val)
    blk.set_time_integration(True, 2, 2, 1, tr["dt"], tr["butcher_A"], tr["butcher_b"], tr["bdf"])
    kw = dict(u_prev=torch.tensor(tr["u_prev"], device="cuda"), u_stage=torch.tensor(tr["u_stage"], device="cuda"))
    ud = torch.tensor(u, device="cuda")
    got = {}
    for form in ("q_streamed", "panels"):
        if form == "panels":
            monkeypatch.setenv("MHA_ENGINE_STOP", "32")
        res = torch.full((m["ndof"],), 7.0, dtype=torch.float64, device="cuda")
        vals = torch.full((len(ref["colind"]),), -3.0, dtype=torch.float64, device="cuda")
        blk.assemble_jacres(ud, res, vals, path=mrhyde_amd.PATH_ROW_GATHER, overwrite=True, **kw)
        torch.cuda.synchronize()
        got[form] = (res.cpu().numpy(), vals.cpu().numpy())
        assert rel_err(got[form][0], ref["res"]) < RTOL and crs_err(got[form][1], ref) < RTOL, form
    assert rel_err(got["q_streamed"][0], got["panels"][0]) < 1e-13 and rel_err(got["q_streamed"][1], got["panels"][1]) < 1e-13


def test_porous_mixed_3d_gold_end_to_end(oracle):
    """regression/porous/Mixed_3d with the GPU assembling: p and u L2 errors print as the reference's gold."""
    torch = _torch()
    import scipy.sparse.linalg as spla
    from test_oracle_multi import fmt, gold_errors
    dim, ncell, qdeg = 3, (8, 8, 8), 2
    m = oracle.mesh_multi(dim, ncell, [oracle.HVOL, oracle.HDIV], [0, 1])
    blk = make_block(m, "porousMixed", qdeg)
    blk.set_function("source", ("sinprod", 12 * np.pi ** 2, [2 * np.pi] * 3))
    rowptr, colind = blk.get_graph()
    out = run_gpu(blk, m, np.zeros(m["ndof"]), None, len(colind))
    J = sp.csr_matrix((out["crs_vals"], colind, rowptr), shape=(m["ndof"],) * 2)
    u = spla.spsolve(J.tocsc(), out["res"])
    pb = oracle.physical_basis_var(dim, oracle.HVOL, 0, qdeg, m["nodes"])
    ub = oracle.physical_basis_var(dim, oracle.HDIV, 1, qdeg, m["nodes"], m["orient"][:, m["varptr"][1]:])
    x, w = pb["ip"], pb["wts"]
    s, c = np.sin(2 * np.pi * x), np.cos(2 * np.pi * x)
    ep = np.sqrt(np.sum((u[m["lids"][:, m["offsets"][0]]][:, None] - np.prod(s, axis=-1)) ** 2 * w))
    uh = np.einsum("ef,efqd->eqd", u[m["lids"][:, m["offsets"][m["varptr"][1]:]]], ub["basis"])
    eu = 0.0
    for d in range(dim):
        others = [k for k in range(dim) if k != d]
        eu += np.sum((uh[..., d] + 2 * np.pi * c[..., d] * np.prod(s[..., others], axis=-1)) ** 2 * w)
    g = gold_errors("porous_Mixed_3d.gold")
    assert fmt(ep) == fmt(g["p"]) and fmt(np.sqrt(eu)) == fmt(g["u"])


def test_navierstokes_channel_gold_end_to_end(oracle):
    """regression/navierstokes/channel (Q1/Q1 + PSPG, Newton) with the GPU assembling every iteration."""
    torch = _torch()
    from test_oracle_multi import fmt, gold_errors, solve_ns_channel
    state = {}

    def assemble(m, u):
        if "blk" not in state:
            blk = make_block(m, "navierstokes", 2, fixed=m["fixed"])
            blk.set_function("source ux", 1.0)
            blk.set_physics_parameter("usePSPG", 1)
            state["blk"], state["graph"] = blk, blk.get_graph()
        blk = state["blk"]
        rowptr, colind = state["graph"]
        ud = torch.tensor(u, device="cuda")
        res = torch.empty(m["ndof"], dtype=torch.float64, device="cuda")
        vals = torch.empty(len(colind), dtype=torch.float64, device="cuda")
        blk.assemble_jacres(ud, res, vals, overwrite=True)
        blk.apply_dbc_diag(vals)
        torch.cuda.synchronize()
        return sp.csr_matrix((vals.cpu().numpy(), colind, rowptr), shape=(m["ndof"],) * 2), res.cpu().numpy()

    errs = solve_ns_channel(assemble, oracle)
    g = gold_errors("navierstokes_channel.gold")
    for k in ("ux", "pr", "uy"):
        assert fmt(errs[k]) == fmt(g[k]), (k, errs[k], g[k])


@pytest.mark.parametrize("dim,ncell", [(2, (5, 4)), (3, (3, 2, 3))])
def test_porous_boundary_matches_oracle(oracle, dim, ncell):
    """porousMixed::boundaryResidual (weak Dirichlet p) on curved sides, closed-form and per-ip data."""
    torch = _torch()
    import mrhyde_amd
    rng = np.random.default_rng(34)
    m = warp(oracle.mesh_multi(dim, ncell, [oracle.HVOL, oracle.HDIV], [0, 1]))
    u = rng.uniform(-1, 1, m["ndof"])
    rowptr, colind = oracle.build_graph(m["ndof"], m["lids"])
    nqs = oracle.side_sizes(dim, 2)[1]
    names = ["left", "right", "bottom", "top"] + (["back", "front"] if dim == 3 else [])
    blk = make_block(m, "porousMixed", 2, graph=(rowptr, colind))
    res_ref, vals_ref = np.zeros(m["ndof"]), np.zeros(rowptr[-1])
    keep = []
    for i, name in enumerate(names):
        be, bs = oracle.boundary_sides(dim, ncell, name)
        data = ("array", rng.uniform(-2, 2, (len(be), nqs))) if i % 2 else ("sinprod", 1.7, [1.0, 2.0, 0.5][:dim])
        oracle.assemble_block_boundary(m, oracle.PHYS_POROUS_MIXED, 2, u, be, bs, 1, data, rowptr=rowptr, colind=colind,
                                       crs_vals=vals_ref, res=res_ref)
        blk.add_boundary_group(name, mrhyde_amd.BC_WEAK_DIRICHLET, be, bs)
        if data[0] == "array":
            t = torch.tensor(data[1], device="cuda")
            keep.append(t)
            blk.set_function("Dirichlet p " + name, t)
        else:
            blk.set_function("Dirichlet p " + name, data)
    res = torch.zeros(m["ndof"], dtype=torch.float64, device="cuda")
    vals = torch.zeros(len(colind), dtype=torch.float64, device="cuda")
    blk.assemble_boundary(torch.tensor(u, device="cuda"), res, vals)
    torch.cuda.synchronize()
    assert np.abs(res_ref).max() > 0 and np.all(vals_ref == 0.0)
    assert rel_err(res.cpu().numpy(), res_ref) < RTOL
    assert np.all(vals.cpu().numpy() == 0.0)


def test_porous_mixed_2d_gold_end_to_end(oracle):
    """regression/porous/Mixed: p = 1 on all four sides through the boundary-group path, GPU assembly."""
    torch = _torch()
    import mrhyde_amd
    import scipy.sparse.linalg as spla
    from test_oracle_multi import fmt, gold_errors
    dim, ncell, qdeg = 2, (8, 8), 2
    m = oracle.mesh_multi(dim, ncell, [oracle.HVOL, oracle.HDIV], [0, 1])
    blk = make_block(m, "porousMixed", qdeg)
    blk.set_function("source", ("sinprod", 8 * np.pi ** 2, [2 * np.pi] * 2))
    for name in ("left", "right", "bottom", "top"):
        be, bs = oracle.boundary_sides(dim, ncell, name)
        blk.add_boundary_group(name, mrhyde_amd.BC_WEAK_DIRICHLET, be, bs)
        blk.set_function("Dirichlet p " + name, 1.0)
    rowptr, colind = blk.get_graph()
    ud = torch.zeros(m["ndof"], dtype=torch.float64, device="cuda")
    res = torch.empty(m["ndof"], dtype=torch.float64, device="cuda")
    vals = torch.empty(len(colind), dtype=torch.float64, device="cuda")
    blk.assemble_jacres(ud, res, vals, overwrite=True)
    blk.assemble_boundary(ud, res, vals)
    torch.cuda.synchronize()
    J = sp.csr_matrix((vals.cpu().numpy(), colind, rowptr), shape=(m["ndof"],) * 2)
    u = spla.spsolve(J.tocsc(), res.cpu().numpy())
    pb = oracle.physical_basis_var(dim, oracle.HVOL, 0, qdeg, m["nodes"])
    ub = oracle.physical_basis_var(dim, oracle.HDIV, 1, qdeg, m["nodes"], m["orient"][:, m["varptr"][1]:])
    x, w = pb["ip"], pb["wts"]
    s, c = np.sin(2 * np.pi * x), np.cos(2 * np.pi * x)
    ep = np.sqrt(np.sum((u[m["lids"][:, m["offsets"][0]]][:, None] - 1.0 - np.prod(s, axis=-1)) ** 2 * w))
    uh = np.einsum("ef,efqd->eqd", u[m["lids"][:, m["offsets"][m["varptr"][1]:]]], ub["basis"])
    eu = np.sum((uh[..., 0] + 2 * np.pi * c[..., 0] * s[..., 1]) ** 2 * w) + \
        np.sum((uh[..., 1] + 2 * np.pi * s[..., 0] * c[..., 1]) ** 2 * w)
    g = gold_errors("porous_Mixed.gold")
    assert fmt(ep) == fmt(g["p"]) and fmt(np.sqrt(eu)) == fmt(g["u"])


@pytest.mark.parametrize("physics", ["thermal", "porousMixed", "navierstokes", "shallowwaterHybridized"])
def test_single_element_and_ragged_groups(oracle, physics):
    """Edge cases of the engine's work split: a one-element block (workgroups with idle element slots), element counts
    that are not a multiple of the elements-per-workgroup, a residual-only pass, NULL orientation."""
    torch = _torch()
    import mrhyde_amd
    H, V, D = oracle.HGRAD, oracle.HVOL, oracle.HDIV
    spec = {"thermal": (oracle.PHYS_THERMAL, [H], [2], 3, {"thermal source": 1.5}, []),
            "porousMixed": (oracle.PHYS_POROUS_MIXED, [V, D], [0, 1], 3, {"source": 0.7}, []),
            "navierstokes": (oracle.PHYS_NAVIERSTOKES, [H] * 3, [2, 1, 2], 2, {"source ux": 1.0, "viscosity": 0.1}, [1, 1, 0]),
            "shallowwaterHybridized": (oracle.PHYS_SHALLOWWATER_HYBRIDIZED, [H] * 3, [1, 1, 1], 2, {"source H": 0.1}, [9.81])}
    pid, types, orders, dim, funcs, params = spec[physics]
    rng = np.random.default_rng(51)
    for ncell in ([1] * dim, [3, 1, 1][:dim], [5, 3, 1][:dim]):
        m = warp(oracle.mesh_multi(dim, ncell, types, orders))
        u = rng.uniform(0.5, 1.5, m["ndof"])
        qdeg = 2 * max(max(orders), 1)
        ref = oracle.assemble_block(m, pid, qdeg, u, funcs=funcs, params=params)
        blk = mrhyde_amd.Block(dim, quadrature=qdeg, physics=physics, variables=list(zip(types, orders)))
        blk.set_mesh(m["nodes"], m["lids"], m["offsets"], m["ndof"])
        if physics == "porousMixed":
            blk.set_orientation(m["orient"])
        blk.set_graph(ref["rowptr"], ref["colind"])
        for k, v in funcs.items():
            blk.set_function(k, v)
        for name, val in zip({"navierstokes": ("useSUPG", "usePSPG", "fix_uz_offsets"),
                              "shallowwaterHybridized": ("g",)}.get(physics, ()), params):
            blk.set_physics_parameter(name, val)
        ud = torch.tensor(u, device="cuda")
        res = torch.full((m["ndof"],), 3.0, dtype=torch.float64, device="cuda")
        vals = torch.full((len(ref["colind"]),), 3.0, dtype=torch.float64, device="cuda")
        blk.assemble_jacres(ud, res, vals, overwrite=True)
        assert crs_err(vals.cpu().numpy(), ref) < RTOL and rel_err(res.cpu().numpy(), ref["res"]) < RTOL
        res2 = torch.zeros_like(res)
        blk.assemble_jacres(ud, res2, None, compute_jacobian=False, path=mrhyde_amd.PATH_POINT_ENGINE)
        torch.cuda.synchronize()
        assert rel_err(res2.cpu().numpy(), ref["res"]) < RTOL


def test_launch_that_needs_too_much_scratch_is_refused(oracle, monkeypatch):
    """A kernel whose spills need a large scratch arena can take the process down inside the runtime (the round-1
    MHA_ENGINE_MINW variants of the engine did, on porousMixed 128^3).  The launchers check the kernel's private segment
    first: over the limit -> MHA_ERR_DEVICE and a message, not SIGABRT.  The deck-string instantiation of the engine
    carries the interpreter's stack and the spills around it (about 150 B per lane): with the limit lowered below that it must be refused, and run with the
    default limit."""
    torch = _torch()
    import mrhyde_amd
    m = warp(oracle.mesh_multi(3, (2, 2, 2), [oracle.HVOL, oracle.HDIV], [0, 1]))
    blk = make_block(m, "porousMixed", 2)
    blk.set_function("source", "sin(x)*y + 0.5")
    ud = torch.zeros(m["ndof"], dtype=torch.float64, device="cuda")
    res = torch.zeros(m["ndof"], dtype=torch.float64, device="cuda")
    monkeypatch.setenv("MHA_MAX_SCRATCH_BYTES", "16")
    with pytest.raises(mrhyde_amd.MhaError) as e:
        blk.assemble_jacres(ud, res, None, compute_jacobian=False, path=mrhyde_amd.PATH_POINT_ENGINE)
    assert e.value.code == 3 and "scratch" in str(e.value)  # MHA_ERR_DEVICE
    monkeypatch.delenv("MHA_MAX_SCRATCH_BYTES")
    blk.assemble_jacres(ud, res, None, compute_jacobian=False, path=mrhyde_amd.PATH_POINT_ENGINE)
    torch.cuda.synchronize()
    ref = oracle.assemble_block(m, oracle.PHYS_POROUS_MIXED, 2, np.zeros(m["ndof"]), funcs={"source": ("expr", "sin(x)*y + 0.5", 0.0)})
    assert rel_err(res.cpu().numpy(), ref["res"]) < RTOL


@pytest.mark.parametrize("physics", ["thermal", "porousMixed", "navierstokes", "shallowwaterHybridized"])
def test_deck_string_source_on_every_module(oracle, physics):
    """The deck-string (MHA_FUNC_EXPRESSION) instantiations of the point engine are separate kernels that carry the
    interpreter's stack and spill heavily: each module's, in 2-D and 3-D, on one-wave and whole-workgroup element sizes,
    against the oracle's independent evaluator -- residual and Jacobian, on the engine itself and on the module's default
    path.  (A finite result is not a check: a broken build of the 3-D porousMixed instantiation once produced finite
    garbage here.)"""
    torch = _torch()
    import mrhyde_amd
    H, V, D = oracle.HGRAD, oracle.HVOL, oracle.HDIV
    spec = {"thermal": (oracle.PHYS_THERMAL, [H], [2], "thermal source", []),
            "porousMixed": (oracle.PHYS_POROUS_MIXED, [V, D], [0, 1], "source", []),
            "navierstokes": (oracle.PHYS_NAVIERSTOKES, [H] * 4, [2, 1, 2, 2], "source ux", [1, 1, 0]),
            "shallowwaterHybridized": (oracle.PHYS_SHALLOWWATER_HYBRIDIZED, [H] * 3, [1, 1, 1], "source H", [9.81])}
    pid, types, orders, fname, params = spec[physics]
    expr = "sin(x)*y + 0.5"
    for dim in (2, 3):
        if physics == "shallowwaterHybridized" and dim == 3:
            continue
        ty, od = (types[:3], orders[:3]) if physics == "navierstokes" and dim == 2 else (types, orders)
        for ncell in ([2] * dim, [5, 3, 2][:dim]):
            m = warp(oracle.mesh_multi(dim, ncell, ty, od))
            u = np.random.default_rng(5).uniform(0.5, 1.5, m["ndof"])
            qdeg = 2 * max(max(od), 1)
            ref = oracle.assemble_block(m, pid, qdeg, u, funcs={fname: ("expr", expr, 0.0)}, params=params)
            blk = make_block(m, physics, qdeg, graph=(ref["rowptr"], ref["colind"]))
            blk.set_function(fname, expr)
            for name, val in zip({"navierstokes": ("useSUPG", "usePSPG", "fix_uz_offsets"),
                                  "shallowwaterHybridized": ("g",)}.get(physics, ()), params):
                blk.set_physics_parameter(name, val)
            ud = torch.tensor(u, device="cuda")
            for path in (mrhyde_amd.PATH_POINT_ENGINE, mrhyde_amd.PATH_AUTO):
                res = torch.zeros(m["ndof"], dtype=torch.float64, device="cuda")
                vals = torch.zeros(len(ref["colind"]), dtype=torch.float64, device="cuda")
                blk.assemble_jacres(ud, res, vals, overwrite=(path == mrhyde_amd.PATH_AUTO), path=path)
                torch.cuda.synchronize()
                assert rel_err(res.cpu().numpy(), ref["res"]) < RTOL, (physics, dim, ncell, path)
                assert crs_err(vals.cpu().numpy(), ref) < RTOL, (physics, dim, ncell, path)


def test_workset_views_of_multi_variable_blocks(oracle):
    """Geometry views (wts, x, y, z) work for any block; per-variable basis views are not built and say so."""
    _torch()
    import mrhyde_amd
    m = warp(oracle.mesh_multi(3, (3, 2, 2), [oracle.HVOL, oracle.HDIV], [0, 1]))
    blk = make_block(m, "porousMixed", 2)
    ref = oracle.physical_basis_var(3, oracle.HVOL, 0, 2, m["nodes"])
    blk.workset_update(0)
    assert rel_err(blk.workset_view_numpy("wts"), ref["wts"]) < RTOL
    for d, c in enumerate("xyz"):
        assert rel_err(blk.workset_view_numpy(c), ref["ip"][..., d]) < RTOL
    assert np.array_equal(blk.workset_view_numpy("LIDs"), m["lids"])
    with pytest.raises(mrhyde_amd.MhaError) as e:
        blk.workset_view("basis")
    assert e.value.code == 4


@pytest.mark.parametrize("case", ["thermal", "porousMixed", "navierstokes"])
def test_mass_matrices_match_oracle(oracle, case):
    """getMass / getWeightedMass (assemblyManager.cpp:7776-7925) on warped meshes: HGRAD, HVOL, HDIV blocks."""
    torch = _torch()
    H, V, D = oracle.HGRAD, oracle.HVOL, oracle.HDIV
    spec = {"thermal": ([H], [2], 3, (2, 3, 2), 4), "porousMixed": ([V, D], [0, 1], 3, (3, 2, 2), 2),
            "navierstokes": ([H] * 3, [2, 1, 2], 2, (3, 2), 4)}
    types, orders, dim, ncell, qdeg = spec[case]
    m = warp(oracle.mesh_multi(dim, ncell, types, orders))
    blk = make_block(m, case, qdeg)
    E, n = m["lids"].shape
    for wts in (None, [1.7, 0.4, 2.2][:len(types)]):
        ref = oracle.get_mass(m, qdeg, wts)
        mass = torch.zeros((E, n, n), dtype=torch.float64, device="cuda")
        blk.get_mass(mass, wts)
        torch.cuda.synchronize()
        assert rel_err(mass.cpu().numpy(), ref) < RTOL
        # sum of the HGRAD/HVOL entries of one variable = measure of the element (partition of unity)
        if wts is None and case == "thermal":
            vol = oracle.physical_basis_var(dim, H, 2, qdeg, m["nodes"])["wts"].sum(axis=1)
            assert rel_err(mass.cpu().numpy().sum(axis=(1, 2)), vol) < 1e-12
