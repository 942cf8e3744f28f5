"""Pins the CPU oracle against the reference's own golden data.

Golden files are mirrored DATA from /root/reference/regression (see
tests/golden/make_reference_fixtures.sh)."""
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference")


def _parse_hgrad_gold():
    """-> list of 4 sections: (kind, {(dof,pt): value or tuple})"""
    secs = []
    cur = None
    val_re = re.compile(r"^dof (\d+), point (\d+): ([-0-9.e]+)$")
    grad_re = re.compile(r"^dof (\d+), point (\d+) grad: \(([-0-9.e,]+)\)$")
    for line in open(os.path.join(GOLD, "discretization_HGRAD.gold")):
        line = line.strip()
        m, g = val_re.match(line), grad_re.match(line)
        if m or g:
            kind = "val" if m else "grad"
            if cur is None or cur[0] != kind or (int((m or g).group(1)), int((m or g).group(2))) in cur[1]:
                cur = (kind, {})
                secs.append(cur)
            if m:
                cur[1][(int(m.group(1)), int(m.group(2)))] = float(m.group(3))
            else:
                cur[1][(int(g.group(1)), int(g.group(2)))] = tuple(float(x) for x in g.group(3).split(","))
        elif line.startswith("description"):
            pass
    return secs


def _sig6(x, ref):
    """x reproduces ref to the 6 significant digits the reference prints (std::cout default)."""
    return float("%.6g" % x) == pytest.approx(ref, rel=1e-12, abs=1e-300)


def test_gauss_rules(oracle):
    for n in range(1, 8):
        p, w = oracle.gauss_line(n)
        assert np.all(np.diff(p) < 0), "points must be descending (HGRAD gold)"
        # exact for monomials up to degree 2n-1
        for k in range(2 * n):
            exact = 0.0 if k % 2 else 2.0 / (k + 1)
            assert abs(np.dot(w, p ** k) - exact) < 5e-15
    p, w = oracle.gauss_line(2)
    assert abs(p[0] - 0.5773502691896257) < 2e-16 and abs(w[0] - 1) < 1e-15
    p, w = oracle.gauss_line(3)
    assert abs(p[0] - np.sqrt(0.6)) < 2e-16 and p[1] == 0.0 and abs(w[1] - 8 / 9) < 1e-15


def test_hgrad_gold_quad_and_hex(oracle):
    secs = _parse_hgrad_gold()
    assert [s[0] for s in secs] == ["val", "grad", "val", "grad"]
    assert [len(s[1]) for s in secs] == [16, 16, 64, 64]
    for dim, (vsec, gsec) in ((2, secs[0:2]), (3, secs[2:4])):
        m = oracle.mesh_structured(dim, 1, [1] * dim)
        pb = oracle.physical_basis(dim, 1, 2, m["nodes"])
        for (dof, pt), ref in vsec[1].items():
            assert _sig6(pb["basis"][0, dof, pt], ref), (dim, dof, pt)
        for (dof, pt), ref in gsec[1].items():
            for d in range(dim):
                assert _sig6(pb["basis_grad"][0, dof, pt, d], ref[d]), (dim, dof, pt, d)


def test_simplemesh_map_bit_exact(oracle):
    """SimpleMeshManager_Rectangle cell->node map (simplemeshmanager.hpp:659-675), offsets {0,1,3,2}."""
    nx, ny = 5, 3
    m = oracle.mesh_structured(2, 1, [nx, ny])
    c = 0
    for j in range(ny):
        for i in range(nx):
            exp = [j * (nx + 1) + i, j * (nx + 1) + i + 1, (j + 1) * (nx + 1) + i + 1, (j + 1) * (nx + 1) + i]
            assert m["lids"][c].tolist() == exp
            assert m["cell2vert"][c].tolist() == exp
            c += 1
    assert m["offsets"].tolist() == [0, 1, 3, 2]
    k = 0
    for j in range(ny + 1):
        for i in range(nx + 1):
            assert m["verts"][k, 0] == 0.0 + i * (1.0 / nx) and m["verts"][k, 1] == 0.0 + j * (1.0 / ny)
            k += 1


def _solve_thermal(oracle, dim, order, qdeg, ncell, amp, workset):
    m = oracle.mesh_structured(dim, order, ncell)
    pb = oracle.physical_basis(dim, order, qdeg, m["nodes"])
    fixed = m["boundary"]
    u = np.zeros(m["ndof"])  # initial condition 0, Dirichlet 0 (lifted)
    freq = [2 * np.pi] * dim
    for _ in range(2):  # "max nonlinear iters: 2"
        out = oracle.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, fixed=fixed, pb=pb,
                                      workset_size=workset, source=("sinprod", amp, freq))
        oracle.apply_dbc_diag(fixed, out["rowptr"], out["colind"], out["crs_vals"])
        J = sp.csr_matrix((out["crs_vals"], out["colind"], out["rowptr"]), shape=(m["ndof"],) * 2)
        du = spla.spsolve(J.tocsc(), out["res"])
        u = u + du
        if np.max(np.abs(out["res"])) < 1e-11:
            break
    return oracle.l2_error_sinprod(dim, order, qdeg, m["lids"], m["offsets"], pb, u, freq), out


def _gold_l2(name):
    txt = open(os.path.join(GOLD, name)).read()
    return float(re.search(r"L2 norm of the error for e = ([-0-9.e]+)", txt).group(1))


def test_thermal_2d_verification_gold(oracle):
    err, _ = _solve_thermal(oracle, 2, 1, 2, [40, 40], 8 * np.pi ** 2, 100)
    assert "%.6g" % err == "%.6g" % _gold_l2("thermal_2D_verification.gold") == "0.00102776"
    assert _gold_l2("thermal_2D_verification_mpi.gold") == _gold_l2("thermal_2D_verification.gold")


def test_thermal_3d_verification_gold(oracle):
    err, _ = _solve_thermal(oracle, 3, 1, 2, [10, 10, 10], 12 * np.pi ** 2, 100)
    assert "%.6g" % err == "%.6g" % _gold_l2("thermal_3D_verification.gold") == "0.0116656"


def test_thermal_2d_highorder_gold(oracle):
    err, _ = _solve_thermal(oracle, 2, 4, 8, [10, 10], 8 * np.pi ** 2, 10)
    assert "%.6g" % err == "%.6g" % _gold_l2("thermal_2D_verification_highorder.gold") == "8.59709e-07"


def test_q2_hex_invariants(oracle):
    """Config 2 (Q2 hex) has no reference test: analytic invariants on a perturbed mesh."""
    rng = np.random.default_rng(3)
    m = oracle.mesh_structured(3, 2, [3, 3, 3])
    verts = m["verts"].copy()
    interior = np.all((verts > 1e-12) & (verts < 1 - 1e-12), axis=1)
    verts[interior] += 0.15 / 3 * rng.uniform(-1, 1, size=(interior.sum(), 3))
    nodes = verts[m["cell2vert"]]
    u = rng.uniform(-1, 1, m["ndof"])
    out = oracle.assemble_thermal(3, 2, 4, nodes, m["lids"], m["offsets"], u, source=("const", 1.0), want_local=True)
    J = sp.csr_matrix((out["crs_vals"], out["colind"], out["rowptr"]))
    assert abs(J - J.T).max() < 1e-13                       # symmetric stiffness
    assert np.abs(J @ np.ones(m["ndof"])).max() < 1e-13     # constants in the kernel
    # residual = K u - f with f_i = int N_i  => sum_i(-res_i) = -(1^T K u) + |Omega| = 1
    assert abs(out["res"].sum() - 1.0) < 1e-12
    # local_J follows the same values element by element
    E, n = m["lids"].shape
    dense = np.zeros((m["ndof"],) * 2)
    for e in range(E):
        dense[np.ix_(m["lids"][e], m["lids"][e])] += out["local_J"][e]
    assert np.abs(dense - J.toarray()).max() < 1e-13


def _transient_golds():
    txt = open(os.path.join(GOLD, "thermal_2D_verification_transient.gold")).read()
    return [(float(t), float(v)) for v, t in re.findall(r"for e = ([-0-9.e]+)\s+\(time = ([-0-9.e]+)\)", txt)]


def run_transient_bwe(assemble, oracle, nsteps=20):
    """regression/thermal/2D_verification_transient: backward Euler (Butcher 'BWE', BDF order 1), 20 steps on
    [0,1]; source (8 pi^2 sin 2pi t + 2pi cos 2pi t) sin 2pi x sin 2pi y evaluated at the stage time
    t_n + c*dt with c = 1 (solverManager.cpp:500-503).  `assemble(u, u_prev, t, dt)` -> (J csr, rhs)."""
    dim, order, qdeg = 2, 1, 2
    m = oracle.mesh_structured(dim, order, (40, 40))
    pb = oracle.physical_basis(dim, order, qdeg, m["nodes"])
    freq = [2 * np.pi] * 2
    dt = 1.0 / nsteps
    u = np.zeros(m["ndof"])
    errs = [(0.0, 0.0)]
    for n in range(nsteps):
        t = (n + 1) * dt
        u_prev = u.copy()
        for _ in range(3):  # Newton: the problem is linear, the second pass only confirms convergence
            J, rhs = assemble(m, pb, u, u_prev, t, dt)
            if np.abs(rhs).max() < 1e-10:
                break
            u = u + spla.spsolve(J.tocsc(), rhs)
        # L2 error at the assembly quadrature points (postprocessManager.cpp:1255-1268)
        uh = np.einsum("ed,edp->ep", u[m["lids"][:, m["offsets"]]], pb["basis"])
        ut = np.sin(2 * np.pi * t) * np.prod(np.sin(2 * np.pi * pb["ip"]), axis=2)
        errs.append((t, float(np.sqrt(np.sum((uh - ut) ** 2 * pb["wts"])))))
    return errs


def test_thermal_2d_transient_gold(oracle):
    """Pins the transient seeding (workset.cpp:589-623: alpha_u, alpha_t, beta_t) end to end: 20 printed L2 errors."""
    A, b, bdf = np.array([[1.0]]), np.array([1.0]), np.array([1.0, -1.0])

    def assemble(m, pb, u, u_prev, t, dt):
        amp = 8 * np.pi ** 2 * np.sin(2 * np.pi * t) + 2 * np.pi * np.cos(2 * np.pi * t)
        tr = dict(u_prev=u_prev[:, None].copy(), u_stage=u[:, None].copy(), stage=0, butcher_A=A, butcher_b=b, bdf=bdf,
                  dt=dt)
        out = oracle.assemble_thermal(2, 1, 2, m["nodes"], m["lids"], m["offsets"], u, fixed=m["boundary"], pb=pb,
                                      transient=tr, source=("sinprod", amp, [2 * np.pi] * 2))
        oracle.apply_dbc_diag(m["boundary"], out["rowptr"], out["colind"], out["crs_vals"])
        return sp.csr_matrix((out["crs_vals"], out["colind"], out["rowptr"]), shape=(m["ndof"],) * 2), out["res"]

    errs = run_transient_bwe(assemble, oracle)
    gold = _transient_golds()
    assert len(gold) == 21 and len(errs) == 21
    for (t, e), (tg, eg) in zip(errs, gold):
        assert abs(t - tg) < 1e-12
        if tg == 0.0:
            assert eg == 0.0
            continue
        assert "%.6g" % e == "%.6g" % eg, (t, e, eg)


# ---- boundary terms ---------------------------------------------------------------------------------------------
def test_side_tables_unit_square_and_cube(oracle):
    """Side weights sum to the side measure, normals point outward, side ip lie on the side."""
    for dim, ncell in ((2, (3, 2)), (3, (2, 3, 2))):
        m = oracle.mesh_structured(dim, 1, ncell)
        names = ["bottom", "right", "top", "left"] + (["back", "front"] if dim == 3 else [])
        expect = {"bottom": (1, 0.0, -1), "top": (1, 1.0, 1), "left": (0, 0.0, -1), "right": (0, 1.0, 1),
                  "back": (2, 0.0, -1), "front": (2, 1.0, 1)}
        for name in names:
            be, bs = oracle.boundary_sides(dim, ncell, name)
            sb = oracle.physical_side_basis(dim, 1, 2, m["nodes"], be, bs)
            axis, coord, sign = expect[name]
            assert abs(sb["wts"].sum() - 1.0) < 1e-14
            n_expect = np.zeros(dim)
            n_expect[axis] = sign
            assert np.abs(sb["normals"] - n_expect).max() < 1e-14, name
            assert np.abs(sb["ip"][..., axis] - coord).max() < 1e-14
            assert np.abs(sb["basis"].sum(axis=1) - 1.0).max() < 1e-14      # partition of unity on the side
            assert np.abs(sb["basis_grad"].sum(axis=1)).max() < 1e-12


def solve_mixed_bcs(assemble, oracle):
    """regression/thermal/2D_mixed_bcs: Dirichlet left/right, Neumann top (2 pi sin 2pi x cos 2pi y) and bottom
    (minus that), two Newton iterations.  `assemble(m, u, groups)` -> (J csr with DBC diag, rhs); groups =
    [(elements, sides, data[nb][nqs])]."""
    dim, order, qdeg, ncell = 2, 1, 2, (40, 40)
    m = oracle.mesh_structured(dim, order, ncell)
    pb = oracle.physical_basis(dim, order, qdeg, m["nodes"])
    x = np.zeros((m["ndof"], 2))
    x[m["lids"][:, m["offsets"]].ravel()] = np.repeat(m["nodes"][:, [0, 1, 3, 2]], 1, axis=0).reshape(-1, 2)
    fixed = ((np.abs(x[:, 0]) < 1e-12) | (np.abs(x[:, 0] - 1) < 1e-12)).astype(np.uint8)
    groups = []
    for name, sgn in (("top", 1.0), ("bottom", -1.0)):
        be, bs = oracle.boundary_sides(dim, ncell, name)
        sb = oracle.physical_side_basis(dim, order, qdeg, m["nodes"], be, bs)
        g = sgn * 2 * np.pi * np.sin(2 * np.pi * sb["ip"][..., 0]) * np.cos(2 * np.pi * sb["ip"][..., 1])
        groups.append((be, bs, g))
    m["fixed"] = fixed
    u = np.zeros(m["ndof"])
    for _ in range(4):  # "max nonlinear iters: 4", converges in one
        J, rhs = assemble(m, pb, u, groups)
        u = u + spla.spsolve(J.tocsc(), rhs)
        if np.max(np.abs(rhs)) < 1e-10:
            break
    return oracle.l2_error_sinprod(dim, order, qdeg, m["lids"], m["offsets"], pb, u, [2 * np.pi] * 2)


def test_thermal_2d_mixed_bcs_gold(oracle):
    """Pins the Neumann branch of thermal::boundaryResidual (thermal.cpp:217-226) + side integration data."""
    def assemble(m, pb, u, groups):
        out = oracle.assemble_thermal(2, 1, 2, m["nodes"], m["lids"], m["offsets"], u, fixed=m["fixed"], pb=pb,
                                      source=("sinprod", 8 * np.pi ** 2, [2 * np.pi] * 2))
        for be, bs, g in groups:
            oracle.assemble_thermal_boundary(2, 1, 2, m["nodes"], m["lids"], m["offsets"], u, be, bs, oracle.BC_NEUMANN,
                                             ("array", g), rowptr=out["rowptr"], colind=out["colind"],
                                             crs_vals=out["crs_vals"], res=out["res"], fixed=m["fixed"])
        oracle.apply_dbc_diag(m["fixed"], out["rowptr"], out["colind"], out["crs_vals"])
        return sp.csr_matrix((out["crs_vals"], out["colind"], out["rowptr"]), shape=(m["ndof"],) * 2), out["res"]

    err = solve_mixed_bcs(assemble, oracle)
    assert "%.6g" % err == "%.6g" % _gold_l2("thermal_2D_mixed_bcs.gold") == "0.00102733"


def test_weak_dirichlet_consistency(oracle):
    """Weak Dirichlet (Nitsche, thermal.cpp:237-273) has no reference gold: check that the exact Jacobian from the AD
    restatement matches finite differences of the residual, that it is symmetric for form_param = 1, and that a weakly
    imposed problem converges to the same manufactured solution."""
    dim, order, qdeg, ncell = 2, 2, 4, (8, 8)
    m = oracle.mesh_structured(dim, order, ncell)
    pb = oracle.physical_basis(dim, order, qdeg, m["nodes"])
    rowptr, colind = oracle.build_graph(m["ndof"], m["lids"])
    rng = np.random.default_rng(5)
    sides = [oracle.boundary_sides(dim, ncell, s) for s in ("left", "right", "bottom", "top")]

    def bnd(u, jac=True, sf=1.0):
        vals, res = np.zeros(rowptr[-1]), np.zeros(m["ndof"])
        for be, bs in sides:
            oracle.assemble_thermal_boundary(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, be, bs,
                                             oracle.BC_WEAK_DIRICHLET, ("const", 0.3), rowptr=rowptr, colind=colind,
                                             crs_vals=vals, res=res, diff=1.7, form_param=sf, compute_jacobian=jac)
        return sp.csr_matrix((vals, colind, rowptr), shape=(m["ndof"],) * 2), res

    u = rng.uniform(-1, 1, m["ndof"])
    J, r0 = bnd(u)
    assert abs(J - J.T).max() < 1e-11
    Jm, _ = bnd(u, sf=-1.0)
    assert abs(Jm - Jm.T).max() > 1e-3  # the non-symmetric variant really differs
    du = rng.uniform(-1, 1, m["ndof"])
    _, r1 = bnd(u + du, jac=False)
    # residual is affine in u: res(u+du) - res(u) = -(J du)   (scatter stores -res.val())
    assert np.abs((r1 - r0) + J @ du).max() < 1e-10 * max(1.0, np.abs(J @ du).max())
    # full weakly-imposed solve: -lap u = 8 pi^2 sin sin, u = 0 on the boundary
    vol = oracle.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], np.zeros(m["ndof"]), pb=pb,
                                  source=("sinprod", 8 * np.pi ** 2, [2 * np.pi] * 2), rowptr=rowptr, colind=colind)
    for be, bs in sides:
        oracle.assemble_thermal_boundary(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], np.zeros(m["ndof"]), be,
                                         bs, oracle.BC_WEAK_DIRICHLET, ("const", 0.0), rowptr=rowptr, colind=colind,
                                         crs_vals=vol["crs_vals"], res=vol["res"])
    Jf = sp.csr_matrix((vol["crs_vals"], colind, rowptr), shape=(m["ndof"],) * 2)
    uh = spla.spsolve(Jf.tocsc(), vol["res"])
    err = oracle.l2_error_sinprod(dim, order, qdeg, m["lids"], m["offsets"], pb, uh, [2 * np.pi] * 2)
    assert err < 5e-3
