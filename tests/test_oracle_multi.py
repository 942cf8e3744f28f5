"""Oracle (CPU restatement) of the multi-variable physics against the reference's golds:
regression/porous/Mixed (2-D, weak Dirichlet p through porousMixed::boundaryResidual), porous/Mixed_3d (HVOL + HDIV hex)
and navierstokes/channel (Q1/Q1 + PSPG, Newton).  Fixtures: tests/golden/reference (make_reference_fixtures.sh)."""
import os
import re

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference")


def gold_errors(name):
    txt = open(os.path.join(GOLD, name)).read()
    return {k: float(v) for k, v in re.findall(r"L2 norm of the error for (\w+) = ([-0-9.e]+)", txt)}


def fmt(x):
    return "%.6g" % x


def test_mesh_multi_maps(oracle):
    """Dof-map invariants of the generator: every global dof reached, LIDs of a variable consistent across neighbours,
    HGRAD single-variable map identical to the single-variable generator (bit-exact)."""
    for dim, nc in ((2, (3, 2)), (3, (2, 3, 2))):
        for order in (1, 2):
            m1 = oracle.mesh_structured(dim, order, nc)
            mm = oracle.mesh_multi(dim, nc, [oracle.HGRAD], [order])
            assert np.array_equal(m1["lids"], mm["lids"]) and np.array_equal(m1["offsets"], mm["offsets"])
            assert np.array_equal(m1["boundary"] != 0, mm["side_mask"] != 0)
        m = oracle.mesh_multi(dim, nc, [oracle.HGRAD, oracle.HGRAD] + [oracle.HGRAD] * (dim - 1), [2, 1] + [2] * (dim - 1))
        assert sorted(np.unique(m["lids"])) == list(range(m["ndof"]))
        assert sorted(m["offsets"]) == list(range(m["n_tot"]))
        # the first LID positions interleave the variables at vertex 0 (subcell-major layout)
        assert list(m["dof_var"][m["lids"][0, :dim + 1]]) == list(range(dim + 1))
        mp = oracle.mesh_multi(dim, nc, [oracle.HVOL, oracle.HDIV], [0, 1])
        E = mp["nelem"]
        assert mp["n_tot"] == 1 + 2 * dim and sorted(np.unique(mp["lids"])) == list(range(mp["ndof"]))
        # a face dof is shared by at most two cells, and the two cells give the effective (sign x raw) function the same
        # global direction: raw functions all point along +x_c, so the signs agree
        u0 = mp["varptr"][1]
        seen = {}
        for e in range(E):
            for f in range(2 * dim):
                g = mp["lids"][e, mp["offsets"][u0 + f]]
                seen.setdefault(g, []).append(mp["orient"][e, u0 + f])
        assert all(len(v) <= 2 and len(set(v)) == 1 for v in seen.values())


def _solve_porous(oracle, dim, ncell, pD):
    m = oracle.mesh_multi(dim, ncell, [oracle.HVOL, oracle.HDIV], [0, 1])
    qdeg = 2
    funcs = {"source": ("sinprod", 4 * dim * np.pi ** 2, [2 * np.pi] * dim)}
    u = np.zeros(m["ndof"])
    out = None
    for _ in range(2):  # "max nonlinear iters: 2" (linear problem: second residual ~ 0)
        out = oracle.assemble_block(m, oracle.PHYS_POROUS_MIXED, qdeg, u, funcs=funcs)
        if pD is not None:
            names = ["left", "right", "bottom", "top"] + (["back", "front"] if dim == 3 else [])
            for name in names:
                be, bs = oracle.boundary_sides(dim, ncell, name)
                oracle.assemble_block_boundary(m, oracle.PHYS_POROUS_MIXED, qdeg, u, be, bs, 1, ("const", pD),
                                               rowptr=out["rowptr"], colind=out["colind"], crs_vals=out["crs_vals"],
                                               res=out["res"], funcs=funcs)
        if np.abs(out["res"]).max() < 1e-9:
            break
        J = sp.csr_matrix((out["crs_vals"], out["colind"], out["rowptr"]), shape=(m["ndof"],) * 2)
        u = u + spla.spsolve(J.tocsc(), out["res"])
    # errors (postprocessManager.cpp:1255-1268 "L2", :1358-1420 "L2 VECTOR")
    pb = oracle.physical_basis_var(dim, oracle.HVOL, 0, qdeg, m["nodes"])
    ub = oracle.physical_basis_var(dim, oracle.HDIV, 1, qdeg, m["nodes"], m["orient"][:, m["varptr"][1]:])
    x, w = pb["ip"], pb["wts"]
    s, c = np.sin(2 * np.pi * x), np.cos(2 * np.pi * x)
    p_true = (pD or 0.0) + np.prod(s, axis=-1)
    ph = u[m["lids"][:, m["offsets"][0]]][:, None]
    ep = np.sqrt(np.sum((ph - p_true) ** 2 * w))
    ue = u[m["lids"][:, m["offsets"][m["varptr"][1]:]]]                       # [E][2d] in basis order
    uh = np.einsum("ef,efqd->eqd", ue, ub["basis"])
    eu = 0.0
    for d in range(dim):
        others = [k for k in range(dim) if k != d]
        ut = -2 * np.pi * c[..., d] * np.prod(s[..., others], axis=-1)
        eu += np.sum((uh[..., d] - ut) ** 2 * w)
    return ep, np.sqrt(eu), m, out


def test_porous_mixed_2d_gold(oracle):
    """regression/porous/Mixed: HVOL(0)+HDIV(1) quads, p = 1 imposed weakly on all four sides."""
    ep, eu, _, _ = _solve_porous(oracle, 2, (8, 8), 1.0)
    g = gold_errors("porous_Mixed.gold")
    assert fmt(ep) == fmt(g["p"]) == "0.158697" and fmt(eu) == fmt(g["u"]) == "1.02259"


def test_porous_mixed_3d_gold(oracle):
    """regression/porous/Mixed_3d: HVOL(0)+HDIV(1) hexes, natural p = 0."""
    ep, eu, m, out = _solve_porous(oracle, 3, (8, 8, 8), None)
    g = gold_errors("porous_Mixed_3d.gold")
    assert fmt(ep) == fmt(g["p"]) == "0.135175" and fmt(eu) == fmt(g["u"]) == "1.22176"
    J = sp.csr_matrix((out["crs_vals"], out["colind"], out["rowptr"]))
    assert abs(J - J.T).max() < 1e-13  # saddle-point system is symmetric


def solve_ns_channel(assemble, oracle, ncell=(50, 10)):
    """regression/navierstokes/channel: 5x1 channel, Q1/Q1 + PSPG, ux = uy = 0 on top/bottom, body force 1,
    Newton from 0.  `assemble(m, u)` -> (J csr with DBC rows, rhs)."""
    H, V = oracle.HGRAD, None
    m = oracle.mesh_multi(2, ncell, [H, H, H], [1, 1, 1], hi=[5.0, 1.0, 1.0])
    m["fixed"] = (((m["side_mask"] & 0b1100) != 0) & (m["dof_var"] != 1)).astype(np.uint8)
    u = np.zeros(m["ndof"])
    for it in range(10):  # defaults: "max nonlinear iters" 10, "nonlinear TOL" 1e-6 (solverManager.cpp)
        J, rhs = assemble(m, u)
        if it > 0 and np.linalg.norm(rhs) < 1e-12:
            break
        u = u + spla.spsolve(J.tocsc(), rhs)
    errs = {}
    pb = oracle.physical_basis_var(2, H, 1, 2, m["nodes"])
    y, w = pb["ip"][..., 1], pb["wts"]
    true = {0: 0.5 * y * (1 - y), 1: 0 * y, 2: 0 * y}
    for v, name in ((0, "ux"), (1, "pr"), (2, "uy")):
        ue = u[m["lids"][:, m["offsets"][m["varptr"][v]:m["varptr"][v + 1]]]]
        uh = np.einsum("ef,efq->eq", ue, pb["basis"][..., 0])
        errs[name] = np.sqrt(np.sum((uh - true[v]) ** 2 * w))
    return errs


def test_navierstokes_channel_gold(oracle):
    def assemble(m, u):
        out = oracle.assemble_block(m, oracle.PHYS_NAVIERSTOKES, 2, u, funcs={"source ux": 1.0}, params=[0, 1, 0],
                                    fixed=m["fixed"])
        oracle.apply_dbc_diag(m["fixed"], out["rowptr"], out["colind"], out["crs_vals"])
        return sp.csr_matrix((out["crs_vals"], out["colind"], out["rowptr"]), shape=(m["ndof"],) * 2), out["res"]

    errs = solve_ns_channel(assemble, oracle)
    g = gold_errors("navierstokes_channel.gold")
    for k in ("ux", "pr", "uy"):
        assert fmt(errs[k]) == fmt(g[k]), (k, errs[k], g[k])


def test_block_thermal_equals_single_variable_oracle(oracle):
    """The multi-variable restatement run on thermal must equal the single-variable one entry for entry."""
    rng = np.random.default_rng(4)
    for dim, order, qdeg, nc in ((2, 2, 4, (3, 2)), (3, 1, 2, (2, 2, 3))):
        m = oracle.mesh_multi(dim, nc, [oracle.HGRAD], [order])
        u = rng.uniform(-1, 1, m["ndof"])
        tr = dict(u_prev=rng.uniform(-1, 1, (m["ndof"], 1)), u_stage=u[:, None].copy(), stage=0,
                  butcher_A=np.array([[1.0]]), butcher_b=np.array([1.0]), bdf=np.array([1.0, -1.0]), dt=0.1)
        src = ("sinprod", 2.0, [1.0, 2.0, 3.0][:dim])
        a = oracle.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, source=src, diff=1.3,
                                    rho=0.7, cp=1.9, transient=tr, want_local=True)
        b = oracle.assemble_block(m, oracle.PHYS_THERMAL, qdeg, u, transient=tr, want_local=True,
                                  funcs={"thermal source": src, "thermal diffusion": 1.3, "density": 0.7,
                                         "specific heat": 1.9})
        for k in ("crs_vals", "res", "local_J", "local_res"):
            assert np.abs(a[k] - b[k]).max() <= 1e-13 * np.abs(a[k]).max(), k


def test_navierstokes_jacobian_is_derivative_of_residual(oracle):
    """AD restatement self-check on a nonlinear module: J du matches the residual difference (SUPG + PSPG, 2-D and
    3-D, transient), and the reference's uz-offset quirk (navierstokes.cpp:688) leaves the uz rows empty."""
    rng = np.random.default_rng(9)
    H = oracle.HGRAD
    for dim, nc in ((2, (3, 2)), (3, (2, 2, 2))):
        m = oracle.mesh_multi(dim, nc, [H] * (dim + 1), [2, 1] + [2] * (dim - 1))
        u = rng.uniform(-1, 1, m["ndof"])
        tr = dict(u_prev=rng.uniform(-1, 1, (m["ndof"], 1)), u_stage=u[:, None].copy(), stage=0,
                  butcher_A=np.array([[1.0]]), butcher_b=np.array([1.0]), bdf=np.array([1.0, -1.0]), dt=0.1)
        funcs = {"source ux": 0.3, "source uy": ("sinprod", 1.0, [1.0, 2.0, 0.5][:dim]), "viscosity": 0.05, "density": 1.3}
        kw = dict(funcs=funcs, params=[1, 1, 1], transient=tr)
        a = oracle.assemble_block(m, oracle.PHYS_NAVIERSTOKES, 4, u, **kw)
        J = sp.csr_matrix((a["crs_vals"], a["colind"], a["rowptr"]), shape=(m["ndof"],) * 2)
        du = 1e-6 * rng.uniform(-1, 1, m["ndof"])
        tr2 = dict(tr, u_stage=(u + du)[:, None].copy())
        b = oracle.assemble_block(m, oracle.PHYS_NAVIERSTOKES, 4, u + du, **dict(kw, transient=tr2))
        lin = -(J @ du)
        assert np.abs((b["res"] - a["res"]) - lin).max() < 1e-4 * np.abs(lin).max()
        if dim == 3:
            q = oracle.assemble_block(m, oracle.PHYS_NAVIERSTOKES, 4, u, funcs=funcs, params=[0, 0, 0])
            uz_rows = np.flatnonzero(m["dof_var"] == 3)
            assert np.all(q["res"][uz_rows] == 0.0)
            f = oracle.assemble_block(m, oracle.PHYS_NAVIERSTOKES, 4, u, funcs=funcs, params=[0, 0, 1])
            assert np.abs(f["res"][uz_rows]).max() > 0


def test_thermal_advection_term_of_the_oracle(oracle):
    """Pins the advection restatement (thermal.cpp:150-160) without the GPU: the added matrix is the one of
    (b . grad e, v) -- it is linear in u, so J u reproduces the residual difference exactly, its row sums vanish (grad of
    a constant) and for constant b on a uniform mesh it is skew plus a boundary term, i.e. NOT symmetric."""
    rng = np.random.default_rng(4)
    for dim, order, qdeg, nc in ((2, 2, 4, (4, 3)), (3, 1, 2, (3, 2, 2))):
        m = oracle.mesh_structured(dim, order, nc)
        u = rng.uniform(-1, 1, m["ndof"])
        b = [0.7, -1.1, 0.4][:dim]
        kw = dict(diff=0.9, source=("const", 0.3))
        base = oracle.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, **kw)
        adv = oracle.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, advection=b, **kw)
        n = m["ndof"]
        A = sp.csr_matrix((adv["crs_vals"] - base["crs_vals"], adv["colind"], adv["rowptr"]), shape=(n, n))
        assert abs(A).max() > 1e-3
        assert np.abs(A @ np.ones(n)).max() < 1e-13
        assert np.abs((adv["res"] - base["res"]) + A @ u).max() < 1e-12 * max(1.0, np.abs(adv["res"]).max())
        assert abs(A - A.T).max() > 1e-3
        # the same b given per integration point
        pb = oracle.physical_basis(dim, order, qdeg, m["nodes"])
        E, nq = pb["wts"].shape
        arr = np.broadcast_to(np.asarray(b), (E, nq, dim)).copy()
        adv2 = oracle.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, advection=arr, **kw)
        assert np.array_equal(adv2["crs_vals"], adv["crs_vals"]) and np.array_equal(adv2["res"], adv["res"])


def test_flux_condition_of_the_oracle(oracle):
    """fluxConditions restated (physicsInterface.cpp:1702-1762): against a numpy einsum, fixed rows skipped, and a unit
    flux on a closed boundary integrates the HGRAD partition of unity to the boundary measure."""
    rng = np.random.default_rng(5)
    dim, order, qdeg, nc = 2, 2, 4, (4, 3)
    m = oracle.mesh_structured(dim, order, nc)
    total = 0.0
    for side in ("bottom", "right", "top", "left"):
        belem, bside = oracle.boundary_sides(dim, nc, side)
        sb = oracle.physical_side_basis(dim, order, qdeg, m["nodes"], belem, bside)
        flux = rng.uniform(-1, 1, sb["wts"].shape)
        fixed = np.zeros(m["ndof"], np.uint8)
        fixed[m["lids"][belem[1], 0]] = 1
        got = np.zeros(m["ndof"])
        oracle.flux_condition(belem, m["lids"], m["offsets"], flux, sb["wts"], sb["basis"][..., None], got, fixed=fixed)
        want = np.zeros(m["ndof"])
        np.add.at(want, m["lids"][belem][:, m["offsets"]], np.einsum("kq,kq,kfq->kf", flux, sb["wts"], sb["basis"]))
        want[fixed != 0] = 0.0
        assert np.abs(got - want).max() < 1e-14
        one = np.zeros(m["ndof"])
        oracle.flux_condition(belem, m["lids"], m["offsets"], np.ones_like(flux), sb["wts"], sb["basis"][..., None], one)
        total += one.sum()
    assert abs(total - 4.0) < 1e-13


def test_mass_storage_and_database_restatements(oracle):
    """Pins the oracle's restatements of Sparse3DView, identifyVolumetricDatabase and applyMassMatrixFree against each
    other and against plain linear algebra (no GPU): the four ways of applying M agree, M x equals the assembled
    (scipy) block mass matrix, the compressed storage reproduces the dense one above the threshold, and the database
    of a mesh made of two element shapes has two representatives in order of first appearance."""
    rng = np.random.default_rng(6)
    H, V, D = oracle.HGRAD, oracle.HVOL, oracle.HDIV
    for types, orders, dim, nc, qdeg in (([H], [2], 2, (4, 4), 4), ([V, D], [0, 1], 3, (4, 2, 2), 2)):
        m = oracle.mesh_multi(dim, nc, types, orders)
        # two element shapes, bitwise repeated: stretch the right half of the mesh in x by 1.5
        v = m["verts"].copy()
        right = v[:, 0] > 0.5
        v[right, 0] = 0.5 + 1.5 * (v[right, 0] - 0.5)
        m["verts"] = v
        m["nodes"] = np.ascontiguousarray(v[m["cell2vert"]])
        n, E = m["n_tot"], m["nelem"]
        x = rng.uniform(-1, 1, m["ndof"])
        w = [1.7, 0.6][:len(types)]
        mass = oracle.get_mass(m, qdeg, w)
        y0 = np.zeros(m["ndof"])
        oracle.apply_mass_matrix_free(m, qdeg, x, y0, w)
        rows = np.repeat(m["lids"], n, axis=1).ravel()
        cols = np.tile(m["lids"], (1, n)).ravel()
        M = sp.coo_matrix((mass.ravel(), (rows, cols)), shape=(m["ndof"],) * 2).tocsr()
        assert np.abs(y0 - M @ x).max() < 1e-13 * np.abs(y0).max()
        y1 = np.zeros(m["ndof"])
        oracle.apply_mass_stored(m, mass, x, y1)
        assert np.abs(y1 - y0).max() < 1e-13 * np.abs(y0).max()
        idx, fu = oracle.identify_database(m, qdeg)
        assert len(fu) == 2 and fu[0] == 0 and np.array_equal(np.unique(idx), [0, 1]) and idx[fu[1]] == 1
        if D not in types:   # (orientation signs differ between HDIV elements of the same shape: more representatives)
            assert np.all(np.abs(mass - mass[fu][idx]) < 1e-14)
        y2 = np.zeros(m["ndof"])
        oracle.apply_mass_stored(m, mass[fu], x, y2, index=idx)
        assert np.abs(y2 - y0).max() < 1e-12 * np.abs(y0).max()
        vals, cls, nnz, me = oracle.sparse3d(mass[fu], 1e-12)
        assert me <= n and nnz.max() == me
        dense = np.zeros_like(mass[fu])
        for e in range(len(fu)):
            for i in range(n):
                dense[e, i, cls[e, i, :nnz[e, i]]] = vals[e, i, :nnz[e, i]]
        assert np.abs(dense - mass[fu]).max() <= 1e-12 * np.abs(mass).max()
        y3 = np.zeros(m["ndof"])
        oracle.apply_mass_sparse(m, vals, cls, nnz, x, y3, index=idx)
        assert np.abs(y3 - y0).max() < 1e-12 * np.abs(y0).max()


def test_projection_restatements_partition_of_unity(oracle):
    """setInitial / setDirichlet restatements (oracle/mrhyde_oracle_multi.c) against closed forms: with data == 1 the
    projected right-hand side of an HGRAD variable is the row sum of its mass matrix (partition of unity), the lumped
    matrix carries exactly that on the diagonal, the mass of the whole block integrates 1 to the domain's measure, and the
    Dirichlet system has the boundary row sums on fixed rows and the identity elsewhere."""
    dim, ncell, qdeg = 2, (4, 3), 4
    m = oracle.mesh_multi(dim, ncell, [oracle.HGRAD, oracle.HGRAD], [2, 1], hi=[2.0, 1.5, 1.0])
    nd, n_tot = m["ndof"], m["lids"].shape[1]
    rowptr, colind = oracle.build_graph(nd, m["lids"])
    rhs = np.zeros(nd)
    for v, order in enumerate((2, 1)):
        pb = oracle.physical_basis_var(dim, oracle.HGRAD, order, qdeg, m["nodes"])
        off = m["offsets"][m["varptr"][v]:m["varptr"][v + 1]]
        oracle.project_rhs(m["lids"], off, np.ones(pb["wts"].shape + (1,)), pb["basis"], pb["wts"], rhs)
    mass = oracle.get_mass(m, qdeg)
    vals, lumped = np.zeros(len(colind)), np.zeros(len(colind))
    oracle.set_initial_mass(m["lids"], mass, False, rowptr, colind, vals)
    oracle.set_initial_mass(m["lids"], mass, True, rowptr, colind, lumped)
    rowsum = np.add.reduceat(vals, rowptr[:-1])
    assert np.allclose(rowsum, rhs, rtol=1e-13, atol=1e-15)
    diag = np.array([lumped[rowptr[r]:rowptr[r + 1]][colind[rowptr[r]:rowptr[r + 1]] == r][0] for r in range(nd)])
    assert np.allclose(diag, rhs, rtol=1e-13) and np.isclose(lumped.sum(), diag.sum())
    assert np.isclose(rhs[m["dof_var"] == 0].sum(), 3.0) and np.isclose(rhs[m["dof_var"] == 1].sum(), 3.0)   # |domain| = 2 * 1.5

    # Dirichlet data 1 for variable 0 on the right side (x = 2): length 1.5
    belem, bside = oracle.boundary_sides(dim, ncell, "right")
    sb = oracle.physical_side_basis(dim, 2, qdeg, m["nodes"], belem, bside)
    fixed = ((m["dof_var"] == 0) & ((m["side_mask"] >> 1) & 1 == 1)).astype(np.uint8)
    off = m["offsets"][m["varptr"][0]:m["varptr"][1]]
    dvals, bmass = oracle.dirichlet_boundary(n_tot, off, np.ones(sb["wts"].shape), sb["basis"][..., None], sb["wts"])
    for lump in (False, True):
        dv, dr = np.zeros(len(colind)), np.zeros(nd)
        oracle.set_dirichlet_group(belem, m["lids"], fixed, dvals, bmass, lump, rowptr, colind, dv, dr)
        oracle.set_dirichlet_identity(m["lids"], fixed, rowptr, colind, dv)
        assert np.isclose(dr.sum(), 1.5) and np.all(dr[fixed == 0] == 0.0)
        rs = np.add.reduceat(dv, rowptr[:-1])
        assert np.allclose(rs[fixed == 1], dr[fixed == 1], rtol=1e-13)       # row sums of the boundary mass = (1, basis)
        free = np.flatnonzero(fixed == 0)
        assert np.all(rs[free] == 1.0) and all(dv[rowptr[r]:rowptr[r + 1]][colind[rowptr[r]:rowptr[r + 1]] == r][0] == 1.0 for r in free)
    # nodal values replace entries
    init = np.full(nd, -1.0)
    m1 = oracle.mesh_multi(dim, ncell, [oracle.HGRAD], [1])
    init = np.full(m1["ndof"], -1.0)
    oracle.set_initial_nodal(m1["lids"], m1["offsets"], m1["nodes"][..., 0] + 2 * m1["nodes"][..., 1], init)
    assert np.allclose(init, m1["verts"][:, 0] + 2 * m1["verts"][:, 1])
