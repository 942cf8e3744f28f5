"""GPU parity of shallowwaterHybridized (the part built): the side point functions against the
reference's own unit-test values and the oracle (with derivatives against finite differences of the oracle), and the
volume residual on the point engine against the oracle's AD-array restatement."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import swhdg_unit_values as V  # noqa: E402

pytestmark = pytest.mark.gpu
RTOL = 1e-12


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


def close(a, b, tol=RTOL):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


def dev(a):
    return _torch().tensor(np.atleast_2d(np.asarray(a, dtype=np.float64)), device="cuda")


def test_unit_test_values_on_device():
    import mrhyde_amd
    L, lam, R = mrhyde_amd.swhdg_eigendecomp(V.G, dev(V.EV2D["Shat"]), dev(V.EV2D["n"]))
    assert close(lam.cpu()[0], V.EV2D["lam"]) and close(L.cpu()[0], V.EV2D["L"]) and close(R.cpu()[0], V.EV2D["R"])
    s = V.STAB
    out = mrhyde_amd.swhdg_side_terms(mrhyde_amd.api.SWH_INTERFACE, True, V.G, dev(s["S"]), dev(s["Shat"]), dev(s["n"]))
    assert close(out["term"].cpu()[0], s["roe"])
    out = mrhyde_amd.swhdg_side_terms(mrhyde_amd.api.SWH_INTERFACE, False, V.G, dev(s["S"]), dev(s["Shat"]), dev(s["n"]))
    assert close(out["term"].cpu()[0], s["maxEV"])
    f = V.FLUX
    out = mrhyde_amd.swhdg_side_terms(mrhyde_amd.api.SWH_INTERFACE, True, V.G, dev(f["S"]), dev(f["Shat"]), dev([1.0, 0.0]))
    assert close(out["fluxvec"].cpu()[0, :, 0], f["Fx_hat"]) and close(out["fluxvec"].cpu()[0, :, 1], f["Fy_hat"])
    out = mrhyde_amd.swhdg_side_terms(mrhyde_amd.api.SWH_INTERFACE, True, V.G, dev(f["S"]), dev(f["S"]), dev([1.0, 0.0]))
    assert close(out["fluxvec"].cpu()[0, :, 0], f["Fx"]) and close(out["fluxvec"].cpu()[0, :, 1], f["Fy"])
    b = V.BOUND
    out = mrhyde_amd.swhdg_side_terms(mrhyde_amd.api.SWH_FARFIELD, True, V.G, dev(b["S"]), dev(b["Shat"]), dev(b["n"]),
                                      dev(b["Sinf"]))
    assert close(out["term"].cpu()[0], b["farfield"]) and close(out["iflux"].cpu()[0], b["farfield"])


@pytest.mark.parametrize("side_type,roe", [(0, True), (0, False), (1, True), (2, True)])
def test_side_terms_match_oracle(oracle, side_type, roe):
    torch = _torch()
    import mrhyde_amd
    rng = np.random.default_rng(41 + side_type)
    npts = 500
    S = np.column_stack([rng.uniform(0.5, 3.0, npts), rng.uniform(-4, 4, npts), rng.uniform(-4, 4, npts)])
    Sh = S + rng.uniform(-0.2, 0.2, (npts, 3))
    Sinf = np.column_stack([rng.uniform(0.5, 3.0, npts), rng.uniform(-2, 2, npts), rng.uniform(-2, 2, npts)])
    th = rng.uniform(0, 2 * np.pi, npts)
    n = np.column_stack([np.cos(th), np.sin(th)])
    out = mrhyde_amd.swhdg_side_terms(side_type, roe, V.G, dev(S), dev(Sh), dev(n), dev(Sinf))
    out = {k: v.cpu().numpy() for k, v in out.items()}
    for p in range(npts):
        ref = oracle.swh_interface_flux(2, side_type, roe, S[p], Sh[p], Sinf[p], n[p], V.G)
        assert close(out["iflux"][p], ref), p
        assert close(out["fluxvec"][p], oracle.swh_flux_vector(2, Sh[p], V.G))
        t = (oracle.swh_stab_term(2, S[p], Sh[p], n[p], V.G, roe) if side_type == 0 else
             oracle.swh_boundary_term(2, side_type, S[p], Sh[p], Sinf[p], n[p], V.G))
        assert close(out["term"][p], t)
    # derivatives (forward AD on the device) against central differences of the oracle, away from the kinks of |.|
    for p in range(0, npts, 25):
        for which, key in ((0, "d_dS"), (1, "d_dShat")):
            for j in range(3):
                h = 1e-6
                d = np.zeros(3)
                d[j] = h
                a = (S[p] + d, Sh[p]) if which == 0 else (S[p], Sh[p] + d)
                b = (S[p] - d, Sh[p]) if which == 0 else (S[p], Sh[p] - d)
                fd = (oracle.swh_interface_flux(2, side_type, roe, a[0], a[1], Sinf[p], n[p], V.G) -
                      oracle.swh_interface_flux(2, side_type, roe, b[0], b[1], Sinf[p], n[p], V.G)) / (2 * h)
                assert np.abs(out[key][p][:, j] - fd).max() < 2e-5 * max(1.0, np.abs(fd).max()), (p, key, j)


@pytest.mark.parametrize("order,qdeg,ncell", [(1, 2, (5, 4)), (2, 4, (3, 3))])
@pytest.mark.parametrize("transient", [False, True])
def test_volume_residual_matches_oracle(oracle, order, qdeg, ncell, transient):
    torch = _torch()
    import mrhyde_amd
    from test_multi_gpu import make_block, run_gpu, transient_state, warp
    rng = np.random.default_rng(43)
    H = oracle.HGRAD
    m = warp(oracle.mesh_multi(2, ncell, [H, H, H], [order] * 3))
    u = rng.uniform(-1, 1, m["ndof"])
    u[m["dof_var"] == 0] = rng.uniform(1.0, 2.0, (m["dof_var"] == 0).sum())
    tr = transient_state(rng, m["ndof"], u) if transient else None
    if tr is not None:  # keep H of the seeded stage value positive
        for k in ("u_prev", "u_stage"):
            tr[k][m["dof_var"] == 0] = rng.uniform(1.0, 2.0, ((m["dof_var"] == 0).sum(), 2))
    funcs = {"source H": 0.2, "source Hux": ("sinprod", 1.0, [1.0, 2.0]), "source Huy": -0.4}
    ref = oracle.assemble_block(m, oracle.PHYS_SHALLOWWATER_HYBRIDIZED, qdeg, u, funcs=funcs, params=[7.3], transient=tr,
                                want_local=True)
    blk = make_block(m, "shallowwaterHybridized", qdeg, graph=(ref["rowptr"], ref["colind"]))
    for k, v in funcs.items():
        blk.set_function(k, v)
    blk.set_physics_parameter("g", 7.3)
    out = run_gpu(blk, m, u, tr, len(ref["colind"]), local=True)
    for k in ("crs_vals", "res", "local_J", "local_res"):
        assert np.abs(out[k] - ref[k]).max() < RTOL * np.abs(ref[k]).max(), k


@pytest.mark.parametrize("order,qdeg,ncell", [(1, 2, (4, 3)), (2, 4, (3, 2))])
@pytest.mark.parametrize("bc,roe", [(10, 1), (10, 0), (11, 1), (12, 1)])
def test_boundary_residual_matches_oracle(oracle, order, qdeg, ncell, bc, roe):
    """shallowwaterHybridized::boundaryResidual on ALL sides of every element (the HDG interior problem), trace and
    far-field states as per-point arrays, against the oracle's AD-array restatement."""
    torch = _torch()
    import mrhyde_amd
    from test_multi_gpu import make_block, warp
    from test_oracle_swhdg import all_element_sides
    rng = np.random.default_rng(61)
    H = oracle.HGRAD
    m = warp(oracle.mesh_multi(2, ncell, [H, H, H], [order] * 3))
    u = rng.uniform(-1, 1, m["ndof"])
    u[m["dof_var"] == 0] = rng.uniform(1.0, 2.0, (m["dof_var"] == 0).sum())
    rowptr, colind = oracle.build_graph(m["ndof"], m["lids"])
    be, bs = all_element_sides(ncell)
    nqs = oracle.side_sizes(2, qdeg)[1]
    shp = (len(be), nqs)
    aux = np.dstack([rng.uniform(1.0, 2.0, shp), rng.uniform(-1, 1, shp), rng.uniform(-1, 1, shp)])
    ff = np.dstack([rng.uniform(1.0, 2.0, shp), rng.uniform(-1, 1, shp), rng.uniform(-1, 1, shp)])
    vals_ref, res_ref = np.zeros(rowptr[-1]), np.zeros(m["ndof"])
    oracle.assemble_block_boundary(m, oracle.PHYS_SHALLOWWATER_HYBRIDIZED, qdeg, u, be, bs, bc, 0.0, rowptr=rowptr,
                                   colind=colind, crs_vals=vals_ref, res=res_ref, params=[9.81, roe], aux=aux, farfield=ff)
    blk = make_block(m, "shallowwaterHybridized", qdeg, graph=(rowptr, colind))
    blk.set_physics_parameter("Roe-like stabilization", roe)
    blk.add_boundary_group("skeleton", bc, be, bs)
    keep = []
    for i, v in enumerate(("H", "Hux", "Huy")):
        t = torch.tensor(np.ascontiguousarray(aux[..., i]), device="cuda")
        f = torch.tensor(np.ascontiguousarray(ff[..., i]), device="cuda")
        keep += [t, f]
        blk.set_function("aux %s skeleton" % v, t)
        blk.set_function("Far-field %s skeleton" % v, f)
    res = torch.zeros(m["ndof"], dtype=torch.float64, device="cuda")
    vals = torch.zeros(len(colind), dtype=torch.float64, device="cuda")
    blk.assemble_boundary(torch.tensor(u, device="cuda"), res, vals)
    torch.cuda.synchronize()
    assert np.abs(res.cpu().numpy() - res_ref).max() < RTOL * np.abs(res_ref).max()
    assert np.abs(vals.cpu().numpy() - vals_ref).max() < RTOL * np.abs(vals_ref).max()


def test_hdg_element_residual_vanishes_for_constant_state(oracle):
    """volumeResidual + boundaryResidual on all element sides with trace = constant interior state: zero residual
    (divergence theorem), on the device."""
    torch = _torch()
    import mrhyde_amd
    from test_multi_gpu import make_block, warp
    from test_oracle_swhdg import all_element_sides
    H = oracle.HGRAD
    ncell = (4, 3)
    m = warp(oracle.mesh_multi(2, ncell, [H, H, H], [2, 2, 2]))
    blk = make_block(m, "shallowwaterHybridized", 4)
    be, bs = all_element_sides(ncell)
    blk.add_boundary_group("skeleton", mrhyde_amd.api.BC_SWH_INTERFACE, be, bs)
    const = (1.7, 0.4, -0.3)
    uc = np.zeros(m["ndof"])
    for k, v in enumerate(("H", "Hux", "Huy")):
        uc[m["dof_var"] == k] = const[k]
        blk.set_function("aux %s skeleton" % v, const[k])
    ud = torch.tensor(uc, device="cuda")
    res = torch.empty(m["ndof"], dtype=torch.float64, device="cuda")
    vals = torch.empty(len(blk.get_graph()[1]), dtype=torch.float64, device="cuda")
    blk.assemble_jacres(ud, res, vals, overwrite=True)
    assert res.abs().max().item() > 1e-3  # the volume term alone does not vanish
    blk.assemble_boundary(ud, res, vals)
    torch.cuda.synchronize()
    assert res.abs().max().item() < 1e-12


@pytest.mark.parametrize("roe", [True, False])
@pytest.mark.parametrize("transient", [False, True])
def test_hdg_element_blocks_match_oracle(oracle, roe, transient):
    """The HDG element (12 interior + 24 HFACE trace unknowns, mixed interface / far-field / slip sides): residual and
    the four derivative blocks against the oracle's width-36 AD-array restatement."""
    torch = _torch()
    from test_multi_gpu import make_block, transient_state
    from test_oracle_swhdg import hdg_case
    m, u, lam, st, ff = hdg_case(oracle, ncell=(5, 4), seed=13)
    rng = np.random.default_rng(14)
    tr = None
    if transient:
        tr = transient_state(rng, m["ndof"], u)
        for k in ("u_prev", "u_stage"):
            tr[k][m["dof_var"] == 0] = rng.uniform(1.0, 2.0, ((m["dof_var"] == 0).sum(), 2))
    res_ref, blk_ref = oracle.swh_hdg_element(m, 2, u, lam, st, ff, g=7.3, roe=roe, transient=tr)
    blk = make_block(m, "shallowwaterHybridized", 2)
    blk.set_physics_parameter("g", 7.3)
    blk.set_physics_parameter("Roe-like stabilization", 1 if roe else 0)
    kw = {}
    if tr is not None:
        blk.set_time_integration(True, 2, 2, 1, tr["dt"], tr["butcher_A"], tr["butcher_b"], tr["bdf"])
        kw = dict(u_prev=torch.tensor(tr["u_prev"], device="cuda"), u_stage=torch.tensor(tr["u_stage"], device="cuda"))
    E = m["nelem"]
    res = torch.zeros((E, 36), dtype=torch.float64, device="cuda")
    blocks = torch.zeros((E, 36, 36), dtype=torch.float64, device="cuda")
    blk.swhdg_element_blocks(torch.tensor(u, device="cuda"), torch.tensor(lam, device="cuda"), res, blocks,
                             side_types=torch.tensor(st, device="cuda"), farfield=ff, **kw)
    torch.cuda.synchronize()
    assert np.abs(res.cpu().numpy() - res_ref).max() < RTOL * np.abs(res_ref).max()
    assert np.abs(blocks.cpu().numpy() - blk_ref).max() < RTOL * np.abs(blk_ref).max()


def test_batched_condensation(oracle):
    """mha_batched_condense: Schur complement, condensed right-hand side and interior update of every element against
    numpy, on random diagonally-dominant blocks and on the real HDG element (volume + side blocks, transient)."""
    torch = _torch()
    import mrhyde_amd
    from test_multi_gpu import make_block, transient_state
    from test_oracle_swhdg import hdg_case
    rng = np.random.default_rng(71)

    def check(blocks, res, ni, nt):
        S, g, du, ns = mrhyde_amd.batched_condense(ni, nt, torch.tensor(blocks, device="cuda"), torch.tensor(res, device="cuda"))
        assert ns == 0
        Auu, Aul, Alu, All = blocks[:, :ni, :ni], blocks[:, :ni, ni:], blocks[:, ni:, :ni], blocks[:, ni:, ni:]
        X = np.linalg.solve(Auu, np.concatenate([Aul, res[:, :ni, None]], axis=2))
        S_ref = All - Alu @ X[:, :, :nt]
        g_ref = res[:, ni:] - (Alu @ X[:, :, nt:])[..., 0]
        for got, ref in ((S.cpu().numpy(), S_ref), (g.cpu().numpy(), g_ref), (du.cpu().numpy(), X[:, :, nt])):
            assert np.abs(got - ref).max() < 1e-10 * np.abs(ref).max()

    for ni, nt in ((12, 24), (5, 3), (32, 31), (1, 1)):
        E = 37
        blocks = rng.uniform(-1, 1, (E, ni + nt, ni + nt))
        blocks[:, np.arange(ni), np.arange(ni)] += 0.1 * ni * np.sign(rng.uniform(-1, 1, (E, ni)))  # pivoting still matters
        check(blocks, rng.uniform(-1, 1, (E, ni + nt)), ni, nt)
    # the real thing
    m, u, lam, st, ff = hdg_case(oracle, ncell=(4, 3), seed=15)
    tr = transient_state(rng, m["ndof"], u)
    for k in ("u_prev", "u_stage"):
        tr[k][m["dof_var"] == 0] = rng.uniform(1.0, 2.0, ((m["dof_var"] == 0).sum(), 2))
    res, blk = oracle.swh_hdg_element(m, 2, u, lam, st, ff, transient=tr)
    vol = oracle.assemble_block(m, oracle.PHYS_SHALLOWWATER_HYBRIDIZED, 2, u, params=[9.81], transient=tr, want_local=True)
    off = m["offsets"]
    blk[:, :12, :12] += vol["local_J"][:, off][:, :, off]       # LID-position order -> flattened (variable, dof)
    res[:, :12] += vol["local_res"][:, off]
    check(blk, res, 12, 24)
    # a singular interior block is reported, not silently divided through
    bad = np.zeros((2, 3, 3))
    bad[1] = np.eye(3)
    _, _, _, ns = mrhyde_amd.batched_condense(2, 1, torch.tensor(bad, device="cuda"), torch.zeros((2, 3), dtype=torch.float64, device="cuda"))
    assert ns == 1


def trace_lids(nx, ny):
    """Global trace numbering of an nx x ny macro mesh of HDG quads: 24 unknowns per element = (variable, edge, function)
    with HFACE edges left, bottom, right, top and two linear functions per edge along the edge's coordinate (the same
    direction for the two elements that share it: Basis_HFACE needs no orientation).  -> lids [E][24], number of rows."""
    nvert, nhor = (nx + 1) * ny, nx * (ny + 1)            # vertical edges (x = const), horizontal edges (y = const)
    vert = lambda i, j: j * (nx + 1) + i                   # edge at x_i between y_j and y_j+1
    hor = lambda i, j: nvert + j * nx + i                  # edge at y_j between x_i and x_i+1
    lids = np.zeros((nx * ny, 24), np.int32)
    for j in range(ny):
        for i in range(nx):
            e = j * nx + i
            edges = [vert(i, j), hor(i, j), vert(i + 1, j), hor(i, j + 1)]
            for v in range(3):
                for k, edge in enumerate(edges):
                    for f in range(2):
                        lids[e, (v * 4 + k) * 2 + f] = (edge * 3 + v) * 2 + f
    return lids, (nvert + nhor) * 6


@pytest.mark.parametrize("ncell", [(7, 5), (64, 48)])
def test_flux_to_trace_scatter(ncell):
    """The condensed trace blocks [E][24][24] and flux vectors [E][24] into the macro trace system through
    mha_scatter_plan_* (one wavefront per CRS row, no atomics) against a plain COO accumulation; store and accumulate
    semantics, fixed rows (the macro boundary), matrix-only and vector-only calls."""
    torch = _torch()
    import mrhyde_amd
    import scipy.sparse as sp
    nx, ny = ncell
    lids, nrows = trace_lids(nx, ny)
    E = nx * ny
    rng = np.random.default_rng(41)
    S = rng.uniform(-1, 1, (E, 24, 24))
    gv = rng.uniform(-1, 1, (E, 24))
    fixed = np.zeros(nrows, np.uint8)
    cnt = np.bincount(lids.ravel(), minlength=nrows)
    fixed[cnt == 1] = 1                                   # traces on the macro boundary belong to one element only
    plan = mrhyde_amd.ScatterPlan(lids, nrows, fixed=fixed)
    rowptr, colind = plan.graph()
    assert plan.nnz == len(colind) and np.all(np.diff(rowptr) >= 0)
    # reference: COO sum, fixed rows zero
    rows = np.repeat(lids, 24, axis=1).ravel()
    cols = np.tile(lids, (1, 24)).ravel()
    keep = fixed[rows] == 0
    A = sp.coo_matrix((S.ravel()[keep], (rows[keep], cols[keep])), shape=(nrows, nrows)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    ref_vals = np.zeros(len(colind))
    for r in range(nrows):                                # same graph by construction; place by column
        lo, hi = rowptr[r], rowptr[r + 1]
        a_lo, a_hi = A.indptr[r], A.indptr[r + 1]
        ref_vals[lo + np.searchsorted(colind[lo:hi], A.indices[a_lo:a_hi])] = A.data[a_lo:a_hi]
    ref_res = np.zeros(nrows)
    np.add.at(ref_res, lids.ravel(), gv.ravel())
    ref_res[fixed == 1] = 0.0
    Sd, gd = torch.tensor(S, device="cuda"), torch.tensor(gv, device="cuda")
    vals = torch.full((plan.nnz,), 9.0, dtype=torch.float64, device="cuda")
    res = torch.full((nrows,), -4.0, dtype=torch.float64, device="cuda")
    plan.apply(Sd, gd, res, vals, overwrite=True)
    torch.cuda.synchronize()
    assert np.abs(vals.cpu().numpy() - ref_vals).max() < 1e-13 * np.abs(ref_vals).max()
    assert np.abs(res.cpu().numpy() - ref_res).max() < 1e-13 * np.abs(ref_res).max()
    plan.apply(Sd, gd, res, vals)                          # accumulate on top
    plan.apply(None, gd, res, None)                        # vector only
    plan.apply(Sd, None, None, vals)                       # matrix only
    torch.cuda.synchronize()
    assert np.abs(vals.cpu().numpy() - 3 * ref_vals).max() < 1e-13 * np.abs(ref_vals).max()
    assert np.abs(res.cpu().numpy() - 3 * ref_res).max() < 1e-13 * np.abs(ref_res).max()
    with pytest.raises(mrhyde_amd.MhaError):
        plan.apply(Sd, None, None, None)
    bad = lids.copy()
    bad[0, 0] = nrows
    with pytest.raises(mrhyde_amd.MhaError, match="out of range"):
        mrhyde_amd.ScatterPlan(bad, nrows)
    plan.close()


def dg_hdg_case(oracle, ncell, seed):
    """hdg_case with element-local (discontinuous) interior unknowns: LIDs[e][i] = 12 e + i, as in a subgrid of one HDG
    element per macro element."""
    from test_multi_gpu import transient_state
    from test_oracle_swhdg import hdg_case
    m, u, lam, st, ff = hdg_case(oracle, ncell=ncell, seed=seed)
    E = m["nelem"]
    ue = u[m["lids"]]                                          # [E][12] in LID-position order
    var_pos = m["dof_var"][m["lids"][0]]                      # variable of every LID position (same for all elements)
    m["lids"] = np.arange(12 * E, dtype=np.int32).reshape(E, 12)
    m["ndof"] = 12 * E
    m["dof_var"] = np.tile(var_pos, E).astype(np.int32)
    m["side_mask"] = np.zeros(12 * E, np.uint8)
    u = ue.reshape(-1).copy()
    rng = np.random.default_rng(seed + 100)
    tr = transient_state(rng, m["ndof"], u)
    tr["dt"] = 0.05 / max(ncell)   # a time step ~ h at which Newton on these random states converges (5x larger: most diverge)
    for k in ("u_prev", "u_stage"):
        tr[k][m["dof_var"] == 0] = rng.uniform(1.0, 2.0, ((m["dof_var"] == 0).sum(), 2))
    return m, u, lam, st, ff, tr


def run_subgrid(blk, torch, u, lam, st, ff, tr, max_iter, tol):
    blk.set_time_integration(True, 2, 2, 1, tr["dt"], tr["butcher_A"], tr["butcher_b"], tr["bdf"])
    ud = torch.tensor(u, device="cuda")
    out = blk.swhdg_subgrid_solve(ud, torch.tensor(lam, device="cuda"), max_iter, tol, side_types=torch.tensor(st, device="cuda"),
                                  farfield=ff, u_prev=torch.tensor(tr["u_prev"], device="cuda"),
                                  u_stage=torch.tensor(tr["u_stage"], device="cuda"))
    torch.cuda.synchronize()
    return ud.cpu().numpy(), {k: v.cpu().numpy() for k, v in out.items()}


def test_subgrid_sub_iteration_matches_oracle(oracle):
    """SubGridDtN_Solver::nonlinearSolver on the device (side blocks -> volume block -> element-local solve -> update, a
    fixed schedule of passes without host synchronisation) against the oracle's restatement of the loop: final interior
    state, the reference's iteration count and scaled residual norm per element, and the condensed blocks at the final
    state; shared interior unknowns are refused."""
    torch = _torch()
    import mrhyde_amd
    from test_multi_gpu import make_block
    m, u, lam, st, ff, tr = dg_hdg_case(oracle, (5, 4), 21)
    max_iter, tol = 8, 1e-9
    u_ref, it_ref, sc_ref = oracle.subgrid_nonlinear_solver(m, 2, u, lam, st, ff, max_iter, tol, g=7.3, transient=tr)
    assert it_ref.max() < max_iter and it_ref.min() >= 2 and sc_ref.max() <= tol      # converged inside the budget
    blk = make_block(m, "shallowwaterHybridized", 2)
    blk.set_physics_parameter("g", 7.3)
    u_gpu, out = run_subgrid(blk, torch, u, lam, st, ff, tr, max_iter, tol)
    assert out["num_singular"][0] == 0
    assert np.array_equal(out["iters"], it_ref)
    assert np.abs(u_gpu - u_ref).max() < 1e-10 * np.abs(u_ref).max()
    assert np.all(out["resnorm"] <= tol) and np.abs(out["resnorm"] - sc_ref).max() < 1e-3 * tol + 1e-6 * sc_ref.max()
    # closing pass: Schur complement and condensed right-hand side at the final state
    res, blk36 = oracle.swh_hdg_element(m, 2, u_ref, lam, st, ff, g=7.3, transient=tr)
    vol = oracle.assemble_block(m, oracle.PHYS_SHALLOWWATER_HYBRIDIZED, 2, u_ref, params=[7.3], transient=tr, want_local=True)
    off = m["offsets"]
    blk36[:, :12, :12] += vol["local_J"][:, off][:, :, off]
    res[:, :12] += vol["local_res"][:, off]
    X = np.linalg.solve(blk36[:, :12, :12], np.concatenate([blk36[:, :12, 12:], res[:, :12, None]], axis=2))
    S_ref = blk36[:, 12:, 12:] - blk36[:, 12:, :12] @ X[:, :, :24]
    g_ref = res[:, 12:] - (blk36[:, 12:, :12] @ X[:, :, 24:])[..., 0]
    assert np.abs(out["schur"] - S_ref).max() < 1e-9 * np.abs(S_ref).max()
    assert np.abs(out["gvec"] - g_ref).max() < 1e-8 * max(np.abs(g_ref).max(), np.abs(res).max())
    # one pass only: every element has done exactly one assembly and one update
    u1, o1 = run_subgrid(blk, torch, u, lam, st, ff, tr, 1, tol)
    u1_ref, it1, _ = oracle.subgrid_nonlinear_solver(m, 2, u, lam, st, ff, 1, tol, g=7.3, transient=tr)
    assert np.all(o1["iters"] == 1) and np.array_equal(it1, o1["iters"]) and np.abs(u1 - u1_ref).max() < 1e-11 * np.abs(u1_ref).max()
    # continuous (shared) interior unknowns are not a subgrid of independent elements
    from test_oracle_swhdg import hdg_case
    mc, uc, lamc, stc, ffc = hdg_case(oracle, ncell=(3, 2), seed=3)
    blkc = make_block(mc, "shallowwaterHybridized", 2)
    with pytest.raises(mrhyde_amd.MhaError, match="element-local"):
        blkc.swhdg_subgrid_solve(torch.tensor(uc, device="cuda"), torch.tensor(lamc, device="cuda"), 2, tol)


def test_subgrid_sub_iteration_at_config5_size(oracle):
    """The driver at BASELINE.json's config-5 size (256^2 HDG elements): every element converges within the budget, and --
    the elements being independent -- the first 48 of them agree with the oracle's loop run on those 48 alone."""
    torch = _torch()
    from test_multi_gpu import make_block
    m, u, lam, st, ff, tr = dg_hdg_case(oracle, (256, 256), 33)
    max_iter, tol = 8, 1e-9
    blk = make_block(m, "shallowwaterHybridized", 2)
    u_gpu, out = run_subgrid(blk, torch, u, lam, st, ff, tr, max_iter, tol)
    assert out["num_singular"][0] == 0
    assert out["iters"].max() < max_iter and out["iters"].min() >= 2 and np.all(out["resnorm"] <= tol)
    assert np.all(np.isfinite(out["schur"])) and np.all(np.isfinite(out["gvec"]))
    k = 48
    sub = dict(m)
    sub.update(nelem=k, ndof=12 * k, lids=m["lids"][:k].copy(), nodes=m["nodes"][:k].copy(), cell2vert=m["cell2vert"][:k].copy(),
               orient=m["orient"][:k].copy(), dof_var=m["dof_var"][:12 * k].copy(), side_mask=m["side_mask"][:12 * k].copy())
    trs = dict(tr, u_prev=tr["u_prev"][:12 * k].copy(), u_stage=tr["u_stage"][:12 * k].copy())
    u_ref, it_ref, _ = oracle.subgrid_nonlinear_solver(sub, 2, u[:12 * k], lam[:k], st[:k], ff, max_iter, tol, transient=trs)
    assert np.array_equal(out["iters"][:k], it_ref)
    assert np.abs(u_gpu[:12 * k] - u_ref).max() < 1e-10 * np.abs(u_ref).max()


@pytest.mark.parametrize("steady", [False, True])
def test_fused_element_step_equals_the_unfused_pipeline(oracle, steady, monkeypatch):
    """mha_swhdg_condensed_element (side + volume assembly + static condensation in one kernel, the 36 x 36 block never
    leaves the chip) against the independent four-kernel pipeline -- mha_swhdg_element_blocks + mha_compute_local_jacres +
    the combine + mha_batched_condense -- and against the oracle's blocks condensed with numpy: S, g, du per element, on a
    warped mesh with mixed side types, transient (2-stage tableau, BDF-2: mass term and history) and steady.  The
    sub-iteration driver run with MHA_SUBGRID_UNFUSED=1 gives the same final state as the fused default."""
    torch = _torch()
    import mrhyde_amd
    from test_multi_gpu import make_block
    m, u, lam, st, ff, tr = dg_hdg_case(oracle, (6, 5), 27)
    E = m["nelem"]
    blk = make_block(m, "shallowwaterHybridized", 2)
    blk.set_physics_parameter("g", 7.3)
    blk.set_function("source Hux", ("sinprod", 0.3, [1.3, 0.7, 0.0]))
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")
    kw = {}
    if not steady:
        blk.set_time_integration(True, 2, 2, 1, tr["dt"], tr["butcher_A"], tr["butcher_b"], tr["bdf"])
        kw = dict(u_prev=t(tr["u_prev"]), u_stage=t(tr["u_stage"]))
    ud, ld, sd = t(u), t(lam), t(st)
    # fused
    S = torch.zeros((E, 24, 24), dtype=torch.float64, device="cuda")
    g = torch.zeros((E, 24), dtype=torch.float64, device="cuda")
    du = torch.zeros((E, 12), dtype=torch.float64, device="cuda")
    ns = torch.zeros(1, dtype=torch.int32, device="cuda")
    blk.swhdg_condensed_element(ud, ld, schur=S, gvec=g, du=du, num_singular=ns, side_types=sd, farfield=ff, **kw)
    torch.cuda.synchronize()
    assert int(ns[0]) == 0
    # unfused pipeline
    res = torch.zeros((E, 36), dtype=torch.float64, device="cuda")
    blocks = torch.zeros((E, 36, 36), dtype=torch.float64, device="cuda")
    blk.swhdg_element_blocks(ud, ld, res, blocks, side_types=sd, farfield=ff, **kw)
    lJ = torch.zeros((E, 12, 12), dtype=torch.float64, device="cuda")
    lr = torch.zeros((E, 12), dtype=torch.float64, device="cuda")
    blk.compute_local_jacres(ud, lJ, lr, **kw)
    off = torch.tensor(m["offsets"], device="cuda", dtype=torch.long)
    blocks[:, :12, :12] += lJ[:, off][:, :, off]
    res[:, :12] += lr[:, off]
    S2, g2, du2, nsing = mrhyde_amd.batched_condense(12, 24, blocks, res)
    assert nsing == 0
    for a, b_, name in ((S, S2, "schur"), (g, g2, "gvec"), (du, du2, "du")):
        d = float((a - b_).abs().max() / b_.abs().max())
        assert d < 1e-11, (name, d)
    # oracle blocks, condensed with numpy
    tro = None if steady else tr
    r_o, b_o = oracle.swh_hdg_element(m, 2, u, lam, st, ff, g=7.3, transient=tro)
    vol = oracle.assemble_block(m, oracle.PHYS_SHALLOWWATER_HYBRIDIZED, 2, u, params=[7.3], transient=tro, want_local=True,
                                funcs={"source Hux": ("sinprod", 0.3, [1.3, 0.7, 0.0])})
    o = m["offsets"]
    b_o[:, :12, :12] += vol["local_J"][:, o][:, :, o]
    r_o[:, :12] += vol["local_res"][:, o]
    X = np.linalg.solve(b_o[:, :12, :12], np.concatenate([b_o[:, :12, 12:], r_o[:, :12, None]], axis=2))
    S_ref = b_o[:, 12:, 12:] - b_o[:, 12:, :12] @ X[:, :, :24]
    g_ref = r_o[:, 12:] - (b_o[:, 12:, :12] @ X[:, :, 24:])[..., 0]
    assert np.abs(S.cpu().numpy() - S_ref).max() < 1e-10 * np.abs(S_ref).max()
    assert np.abs(g.cpu().numpy() - g_ref).max() < 1e-10 * max(np.abs(g_ref).max(), np.abs(r_o).max())
    assert np.abs(du.cpu().numpy() - X[:, :, 24]).max() < 1e-10 * np.abs(X[:, :, 24]).max()
    if not steady:  # the driver through both pipelines
        u_f, out_f = run_subgrid(blk, torch, u, lam, st, ff, tr, 6, 1e-9)
        monkeypatch.setenv("MHA_SUBGRID_UNFUSED", "1")
        u_u, out_u = run_subgrid(blk, torch, u, lam, st, ff, tr, 6, 1e-9)
        monkeypatch.delenv("MHA_SUBGRID_UNFUSED")
        assert np.array_equal(out_f["iters"], out_u["iters"])
        assert np.abs(u_f - u_u).max() < 1e-11 * np.abs(u_u).max()
        assert np.abs(out_f["schur"] - out_u["schur"]).max() < 1e-10 * np.abs(out_u["schur"]).max()
