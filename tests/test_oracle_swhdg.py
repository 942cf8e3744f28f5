"""Oracle restatement of shallowwaterHybridized's point functions against the reference's own unit-test values
(tests/golden/swhdg_unit_values.py) and consistency of the pieces (A = R Lambda L, flux Jacobian)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import swhdg_unit_values as V  # noqa: E402

TOL = 1e-12  # relative; the golds carry 15-17 significant digits


def close(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.abs(a - b).max() <= TOL * max(1.0, np.abs(b).max())


def test_matvec(oracle):
    assert np.array_equal(oracle.swh_matvec(V.MATVEC["A"], V.MATVEC["x"]), np.array(V.MATVEC["y"], dtype=float))


def test_eigendecomposition_golds(oracle):
    L, lam, R = oracle.swh_eigendecomp(1, V.EV1D["Shat"], [1.0], V.G)
    assert close(lam, V.EV1D["lam"]) and close(L, V.EV1D["L"]) and close(R, V.EV1D["R"])
    L, lam, R = oracle.swh_eigendecomp(2, V.EV2D["Shat"], V.EV2D["n"], V.G)
    assert close(lam, V.EV2D["lam"]) and close(L, V.EV2D["L"]) and close(R, V.EV2D["R"])
    # L R = I and R diag(lam) L = d(F.n)/dS (finite differences of the flux vector)
    assert close(L @ R, np.eye(3))
    A = R @ np.diag(lam) @ L
    S0, n = np.array(V.EV2D["Shat"]), np.array(V.EV2D["n"])
    fd = np.zeros((3, 3))
    for j in range(3):
        d = np.zeros(3)
        d[j] = 1e-6 * max(1.0, abs(S0[j]))
        fd[:, j] = ((oracle.swh_flux_vector(2, S0 + d, V.G) - oracle.swh_flux_vector(2, S0 - d, V.G)) @ n) / (2 * d[j])
    assert np.abs(A - fd).max() < 1e-6 * np.abs(A).max()


def test_flux_vector_golds(oracle):
    F = oracle.swh_flux_vector(2, V.FLUX["S"], V.G)
    assert close(F[:, 0], V.FLUX["Fx"]) and close(F[:, 1], V.FLUX["Fy"])
    F = oracle.swh_flux_vector(2, V.FLUX["Shat"], V.G)
    assert close(F[:, 0], V.FLUX["Fx_hat"]) and close(F[:, 1], V.FLUX["Fy_hat"])


def test_stabilization_golds(oracle):
    s = V.STAB
    assert close(oracle.swh_stab_term(2, s["S"], s["Shat"], s["n"], V.G, roe=True), s["roe"])
    assert close(oracle.swh_stab_term(2, s["S"], s["Shat"], s["n"], V.G, roe=False), s["maxEV"])


def test_boundary_term_gold(oracle):
    b = V.BOUND
    assert close(oracle.swh_boundary_term(2, 1, b["S"], b["Shat"], b["Sinf"], b["n"], V.G), b["farfield"])
    # slip (no reference value): depth jump, tangential velocity jump, zero normal velocity of the state part
    out = oracle.swh_boundary_term(2, 2, b["S"], b["Shat"], b["Sinf"], b["n"], V.G)
    S, Sh, n = np.array(b["S"]), np.array(b["Shat"]), np.array(b["n"])
    u = S[1:] / S[0]
    assert close(out[0], S[0] - Sh[0]) and close(out[1:], (u - (u @ n) * n) - Sh[1:] / Sh[0])
    # interface flux = F(Shat).n + Stab (S - Shat)
    f = oracle.swh_interface_flux(2, 0, True, b["S"], b["Shat"], b["Sinf"], b["n"], V.G)
    ref = oracle.swh_flux_vector(2, b["Shat"], V.G) @ n + oracle.swh_stab_term(2, b["S"], b["Shat"], b["n"], V.G, True)
    assert close(f, ref)


def test_volume_residual_is_consistent(oracle):
    """shallowwaterHybridized::volumeResidual through the AD-array restatement: Jacobian = derivative of the residual,
    constant state with zero sources gives a residual that sums to zero per equation (sum_i grad N_i = 0)."""
    import scipy.sparse as sp
    H = oracle.HGRAD
    m = oracle.mesh_multi(2, (4, 3), [H, H, H], [1, 1, 1])
    rng = np.random.default_rng(2)
    u = rng.uniform(-1, 1, m["ndof"])
    u[m["dof_var"] == 0] = rng.uniform(1.0, 2.0, (m["dof_var"] == 0).sum())  # H > 0
    tr = dict(u_prev=rng.uniform(-1, 1, (m["ndof"], 1)), u_stage=u[:, None].copy(), stage=0,
              butcher_A=np.array([[1.0]]), butcher_b=np.array([1.0]), bdf=np.array([1.0, -1.0]), dt=0.1)
    kw = dict(funcs={"source H": 0.2, "source Hux": ("sinprod", 1.0, [1.0, 2.0])}, params=[9.81], transient=tr)
    a = oracle.assemble_block(m, oracle.PHYS_SHALLOWWATER_HYBRIDIZED, 2, u, **kw)
    J = sp.csr_matrix((a["crs_vals"], a["colind"], a["rowptr"]), shape=(m["ndof"],) * 2)
    du = 1e-6 * rng.uniform(-1, 1, m["ndof"])
    b = oracle.assemble_block(m, oracle.PHYS_SHALLOWWATER_HYBRIDIZED, 2, u + du,
                              **dict(kw, transient=dict(tr, u_stage=(u + du)[:, None].copy())))
    lin = -(J @ du)
    assert np.abs((b["res"] - a["res"]) - lin).max() < 1e-4 * np.abs(lin).max()
    uc = np.zeros(m["ndof"])
    for v, val in enumerate((1.7, 0.4, -0.3)):
        uc[m["dof_var"] == v] = val
    c = oracle.assemble_block(m, oracle.PHYS_SHALLOWWATER_HYBRIDIZED, 2, uc, params=[9.81])
    for v in range(3):
        assert abs(c["res"][m["dof_var"] == v].sum()) < 1e-12


def all_element_sides(ncell):
    """Every (element, local side) of a structured quad mesh: the HDG residual visits all four sides of every element."""
    E = ncell[0] * ncell[1]
    return np.repeat(np.arange(E, dtype=np.int32), 4), np.tile(np.arange(4, dtype=np.int32), E)


def test_hdg_element_residual_is_consistent(oracle):
    """boundaryResidual (:190-263) + volumeResidual: with the trace equal to a constant interior state the element
    residual vanishes (divergence theorem: int_dK F.n N - int_K F.grad N = 0), the stabilisation drops out, and the
    Jacobian of the side term is the derivative of its residual (Roe-like, max-EV, far-field, slip)."""
    import scipy.sparse as sp
    H = oracle.HGRAD
    ncell = (3, 2)
    m = oracle.mesh_multi(2, ncell, [H, H, H], [1, 1, 1])
    v = m["verts"].copy()
    v[:, 0] += 0.07 * np.sin(2.0 * v[:, 1])
    v[:, 1] += 0.05 * v[:, 0] ** 2
    m["verts"], m["nodes"] = v, np.ascontiguousarray(v[m["cell2vert"]])
    rowptr, colind = oracle.build_graph(m["ndof"], m["lids"])
    be, bs = all_element_sides(ncell)
    nqs = oracle.side_sizes(2, 2)[1]
    const = np.array([1.7, 0.4, -0.3])
    uc = np.zeros(m["ndof"])
    for k in range(3):
        uc[m["dof_var"] == k] = const[k]
    vol = oracle.assemble_block(m, oracle.PHYS_SHALLOWWATER_HYBRIDIZED, 2, uc, params=[9.81, 1], rowptr=rowptr,
                                colind=colind)
    aux = np.tile(const, (len(be), nqs, 1))
    oracle.assemble_block_boundary(m, oracle.PHYS_SHALLOWWATER_HYBRIDIZED, 2, uc, be, bs, 10, 0.0, rowptr=rowptr,
                                   colind=colind, crs_vals=vol["crs_vals"], res=vol["res"], params=[9.81, 1], aux=aux)
    assert np.abs(vol["res"]).max() < 1e-12
    rng = np.random.default_rng(8)
    u = rng.uniform(-1, 1, m["ndof"])
    u[m["dof_var"] == 0] = rng.uniform(1.0, 2.0, (m["dof_var"] == 0).sum())
    aux = np.dstack([rng.uniform(1.0, 2.0, (len(be), nqs)), rng.uniform(-1, 1, (len(be), nqs)), rng.uniform(-1, 1, (len(be), nqs))])
    ff = np.dstack([rng.uniform(1.0, 2.0, (len(be), nqs)), rng.uniform(-1, 1, (len(be), nqs)), rng.uniform(-1, 1, (len(be), nqs))])
    for bc, roe in ((10, 1), (10, 0), (11, 1), (12, 1)):
        def bnd(uu):
            vals, res = np.zeros(rowptr[-1]), np.zeros(m["ndof"])
            oracle.assemble_block_boundary(m, oracle.PHYS_SHALLOWWATER_HYBRIDIZED, 2, uu, be, bs, bc, 0.0, rowptr=rowptr,
                                           colind=colind, crs_vals=vals, res=res, params=[9.81, roe], aux=aux, farfield=ff)
            return sp.csr_matrix((vals, colind, rowptr), shape=(m["ndof"],) * 2), res
        J, r0 = bnd(u)
        du = 1e-6 * rng.uniform(-1, 1, m["ndof"])
        _, r1 = bnd(u + du)
        lin = -(J @ du)
        assert np.abs((r1 - r0) - lin).max() < 1e-4 * np.abs(lin).max(), (bc, roe)


def hdg_case(oracle, ncell=(3, 2), seed=12):
    H = oracle.HGRAD
    m = oracle.mesh_multi(2, ncell, [H, H, H], [1, 1, 1])
    v = m["verts"].copy()
    v[:, 0] += 0.07 * np.sin(2.0 * v[:, 1])
    v[:, 1] += 0.05 * v[:, 0] ** 2
    m["verts"], m["nodes"] = v, np.ascontiguousarray(v[m["cell2vert"]])
    rng = np.random.default_rng(seed)
    u = rng.uniform(-1, 1, m["ndof"])
    u[m["dof_var"] == 0] = rng.uniform(1.0, 2.0, (m["dof_var"] == 0).sum())
    E = m["nelem"]
    lam = rng.uniform(-1, 1, (E, 3, 4, 2))
    lam[:, 0] = rng.uniform(1.0, 2.0, (E, 4, 2))
    st = rng.integers(0, 3, (E, 4)).astype(np.uint8)
    return m, u, lam.reshape(E, 24), st, np.array([1.4, -0.3, 0.5])


def test_hdg_element_blocks(oracle):
    """HDG element (interior rows = boundaryResidual on the four sides, trace rows = computeFlux against the HFACE
    basis): blocks are the derivative of the residual with respect to interior AND trace unknowns; the interior rows
    agree with the boundary-group restatement fed with the traces evaluated at the side points."""
    from test_oracle_swhdg import all_element_sides  # noqa: F401 (self-import keeps the helper usable from GPU tests)
    m, u, lam, st, ff = hdg_case(oracle)
    E = m["nelem"]
    for roe in (True, False):
        res, blk = oracle.swh_hdg_element(m, 2, u, lam, st, ff, roe=roe)
        rng = np.random.default_rng(3)
        du, dl = 1e-6 * rng.uniform(-1, 1, m["ndof"]), 1e-6 * rng.uniform(-1, 1, lam.shape)
        res2, _ = oracle.swh_hdg_element(m, 2, u + du, lam + dl, st, ff, roe=roe)
        ue = du[m["lids"][:, m["offsets"]]]                      # [E][12] interior perturbation, flattened (var, dof)
        lin = -np.einsum("erc,ec->er", blk, np.hstack([ue, dl]))
        assert np.abs((res2 - res) - lin).max() < 2e-4 * np.abs(lin).max()
    # interior rows against orc_assemble_block_boundary: one group per (element, side) type, aux = traces at the points
    res, blk = oracle.swh_hdg_element(m, 2, u, lam, st, ff, roe=True)
    rowptr, colind = oracle.build_graph(m["ndof"], m["lids"])
    vals, rg = np.zeros(rowptr[-1]), np.zeros(m["ndof"])
    sip = oracle.side_tables(2, 1, 2)["sip"]                       # [4][nqs][2] reference side points
    hedge = [1, 2, 3, 0]
    for t in range(3):
        be, bs = np.nonzero(st == t)
        if len(be) == 0:
            continue
        nqs = sip.shape[1]
        aux = np.zeros((len(be), nqs, 3))
        for k, (e, s) in enumerate(zip(be, bs)):
            tc = sip[s, :, 1] if hedge[s] in (0, 2) else sip[s, :, 0]
            mu = np.stack([0.5 * (1 - tc), 0.5 * (1 + tc)], axis=1)    # [nqs][2]
            aux[k] = np.einsum("vk,qk->qv", lam[e].reshape(3, 4, 2)[:, hedge[s]], mu)
        oracle.assemble_block_boundary(m, oracle.PHYS_SHALLOWWATER_HYBRIDIZED, 2, u, be.astype(np.int32),
                                       bs.astype(np.int32), 10 + t, 0.0, rowptr=rowptr, colind=colind, crs_vals=vals,
                                       res=rg, params=[9.81, 1], aux=aux, farfield=np.tile(ff, (len(be), nqs, 1)))
    rsum = np.zeros(m["ndof"])
    np.add.at(rsum, m["lids"][:, m["offsets"]].ravel(), res[:, :12].ravel())
    assert np.abs(rsum - rg).max() < 1e-12 * np.abs(rg).max()
