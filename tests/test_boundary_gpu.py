"""GPU parity tests of the boundary-group path: side integration data and
thermal::boundaryResidual (Neumann + weak Dirichlet) through the C ABI against the CPU oracle; the Neumann branch is
additionally pinned end to end by the reference's regression/thermal/2D_mixed_bcs gold."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

RTOL = 1e-12
SIDES2, SIDES3 = ["bottom", "right", "top", "left"], ["bottom", "right", "top", "left", "back", "front"]
CASES = [  # dim, order, qdeg, ncell
    (2, 1, 2, (5, 4)),
    (2, 2, 4, (4, 3)),
    (2, 4, 8, (3, 2)),
    (3, 1, 2, (3, 2, 2)),
    (3, 2, 4, (2, 3, 2)),
]


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def warped(oracle, dim, order, ncell):
    """Smoothly warped mesh: every side is curved / non-axis-aligned, so normals and side measures are non-trivial."""
    m = oracle.mesh_structured(dim, order, ncell)
    v = m["verts"].copy()
    w = v.copy()
    w[:, 0] += 0.08 * np.sin(1.3 * v[:, 1] + 0.4) + (0.05 * v[:, 2] ** 2 if dim == 3 else 0.0)
    w[:, 1] += 0.06 * np.cos(1.1 * v[:, 0]) * (1 + 0.5 * v[:, 1])
    if dim == 3:
        w[:, 2] += 0.07 * v[:, 0] * v[:, 1] + 0.03 * np.sin(2.0 * v[:, 2])
    m["verts"] = w
    m["nodes"] = np.ascontiguousarray(w[m["cell2vert"]])
    return m


def make_block(m, dim, order, qdeg, fixed=None, graph=None):
    import mrhyde_amd
    blk = mrhyde_amd.Block(dim, order, quadrature=qdeg)
    blk.set_mesh(m["nodes"], m["lids"], m["offsets"], m["ndof"], fixed)
    blk.set_graph(*graph) if graph is not None else blk.set_graph()
    return blk


@pytest.mark.parametrize("dim,order,qdeg,ncell", CASES)
def test_boundary_views_match_oracle(oracle, dim, order, qdeg, ncell):
    _torch()
    import mrhyde_amd
    m = warped(oracle, dim, order, ncell)
    blk = make_block(m, dim, order, qdeg)
    for name in (SIDES2 if dim == 2 else SIDES3):
        be, bs = oracle.boundary_sides(dim, ncell, name)
        ref = oracle.physical_side_basis(dim, order, qdeg, m["nodes"], be, bs)
        gid = blk.add_boundary_group(name, mrhyde_amd.BC_NEUMANN, be, bs)
        blk.boundary_update(gid)
        assert rel_err(blk.boundary_view_numpy(gid, "wts side"), ref["wts"]) < RTOL
        for d, c in enumerate("xyz"[:dim]):
            assert rel_err(blk.boundary_view_numpy(gid, c), ref["ip"][..., d]) < RTOL
            assert np.abs(blk.boundary_view_numpy(gid, "n[%s]" % c) - ref["normals"][..., d]).max() < RTOL
        b = blk.boundary_view_numpy(gid, "basis side")
        assert b.shape == ref["basis"].shape + (1,)
        assert np.array_equal(b[..., 0], ref["basis"])  # reference values copied, bit-exact
        assert rel_err(blk.boundary_view_numpy(gid, "basis_grad side"), ref["basis_grad"]) < RTOL
    assert blk.num_boundary_groups() == 2 * dim
    blk.clear_boundary_groups()
    assert blk.num_boundary_groups() == 0


def _groups(oracle, dim, ncell, rng, nqs):
    """Two Neumann sides (array data / closed form) and the remaining sides weak Dirichlet."""
    names = SIDES2 if dim == 2 else SIDES3
    out = []
    for i, name in enumerate(names):
        be, bs = oracle.boundary_sides(dim, ncell, name)
        if i == 0:
            out.append((name, 1, be, bs, ("array", rng.uniform(-2, 2, (len(be), nqs)))))
        elif i == 1:
            out.append((name, 1, be, bs, ("sinprod", 1.3, [2.0, 1.0, 0.7][:dim])))
        elif i == 2:
            out.append((name, 2, be, bs, ("const", 0.3)))
        else:
            out.append((name, 2, be, bs, ("sinprod", 0.8, [1.0, 2.0, 1.5][:dim])))
    return out


@pytest.mark.parametrize("dim,order,qdeg,ncell", CASES)
@pytest.mark.parametrize("mode", ["steady", "transient", "sf-1"])
def test_boundary_jacres_matches_oracle(oracle, dim, order, qdeg, ncell, mode):
    torch = _torch()
    m = warped(oracle, dim, order, ncell)
    rng = np.random.default_rng(21)
    u = rng.uniform(-1, 1, m["ndof"])
    fixed = (rng.uniform(size=m["ndof"]) < 0.1).astype(np.uint8)
    rowptr, colind = oracle.build_graph(m["ndof"], m["lids"])
    nqs = oracle.side_sizes(dim, qdeg)[1]
    groups = _groups(oracle, dim, ncell, rng, nqs)
    sf = -1.0 if mode == "sf-1" else 1.0
    tr = None
    if mode == "transient":  # a 2-stage DIRK-like tableau, second stage
        A, b, bdf = np.array([[0.5, 0.0], [0.3, 0.7]]), np.array([0.4, 0.6]), np.array([1.5, -2.0, 0.5])
        tr = dict(u_prev=rng.uniform(-1, 1, (m["ndof"], 2)), u_stage=rng.uniform(-1, 1, (m["ndof"], 2)), stage=1,
                  butcher_A=A, butcher_b=b, bdf=bdf, dt=0.05)
    vals_ref, res_ref = np.zeros(rowptr[-1]), np.zeros(m["ndof"])
    for name, bc, be, bs, data in groups:
        oracle.assemble_thermal_boundary(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, be, bs, bc, data,
                                         rowptr=rowptr, colind=colind, crs_vals=vals_ref, res=res_ref, fixed=fixed,
                                         transient=tr, diff=1.7, form_param=sf)
    blk = make_block(m, dim, order, qdeg, fixed=fixed, graph=(rowptr, colind))
    blk.set_function("thermal diffusion", 1.7)
    blk.set_physics_parameter("form_param", sf)
    keep = []
    for name, bc, be, bs, data in groups:
        blk.add_boundary_group(name, bc, be, bs)
        fname = ("Neumann e " if bc == 1 else "Dirichlet e ") + name
        if data[0] == "const":
            blk.set_function(fname, data[1])
        elif data[0] == "sinprod":
            blk.set_function(fname, data)
        else:
            t = torch.tensor(data[1], device="cuda")
            keep.append(t)
            blk.set_function(fname, t)
    kw = {}
    if tr is not None:
        blk.set_time_integration(True, 2, 2, 1, tr["dt"], tr["butcher_A"], tr["butcher_b"], tr["bdf"])
        kw = dict(u_prev=torch.tensor(tr["u_prev"], device="cuda"), u_stage=torch.tensor(tr["u_stage"], device="cuda"))
    ud = torch.tensor(u, device="cuda")
    res = torch.zeros(m["ndof"], dtype=torch.float64, device="cuda")
    vals = torch.zeros(len(colind), dtype=torch.float64, device="cuda")
    blk.assemble_boundary(ud, res, vals, **kw)
    torch.cuda.synchronize()
    assert np.abs(vals_ref).max() > 0 and np.abs(res_ref).max() > 0
    assert rel_err(vals.cpu().numpy(), vals_ref) < RTOL
    assert rel_err(res.cpu().numpy(), res_ref) < RTOL
    # accumulation: a second call doubles; residual-only pass leaves the matrix alone
    blk.assemble_boundary(ud, res, vals, compute_jacobian=False, **kw)
    torch.cuda.synchronize()
    assert rel_err(vals.cpu().numpy(), vals_ref) < RTOL
    assert rel_err(res.cpu().numpy(), 2 * res_ref) < RTOL


def test_mixed_bcs_gold_end_to_end(oracle):
    """regression/thermal/2D_mixed_bcs through the HIP path: volume (auto path) + two Neumann groups + DBC rows;
    the L2 error prints as the reference's gold, 0.00102733."""
    torch = _torch()
    import mrhyde_amd
    from test_oracle_golden import _gold_l2, solve_mixed_bcs
    state = {}

    def assemble(m, pb, u, groups):
        if "blk" not in state:
            blk = make_block(m, 2, 1, 2, fixed=m["fixed"])
            blk.set_function("thermal source", ("sinprod", 8 * np.pi ** 2, [2 * np.pi] * 2))
            for name, (be, bs, g) in zip(("top", "bottom"), groups):
                blk.add_boundary_group(name, mrhyde_amd.BC_NEUMANN, be, bs)
                t = torch.tensor(g, device="cuda")
                state[name] = t
                blk.set_function("Neumann e " + name, t)
            state["blk"] = blk
            state["graph"] = blk.get_graph()
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


def test_boundary_error_behaviour(oracle):
    torch = _torch()
    import mrhyde_amd
    m = oracle.mesh_structured(2, 1, (3, 3))
    blk = make_block(m, 2, 1, 2)
    be, bs = oracle.boundary_sides(2, (3, 3), "top")
    with pytest.raises(mrhyde_amd.MhaError) as e:
        blk.add_boundary_group("top", mrhyde_amd.BC_NEUMANN, be, bs + 4)
    assert e.value.code == 1
    with pytest.raises(mrhyde_amd.MhaError) as e:
        blk.add_boundary_group("top", 7, be, bs)
    assert e.value.code == 1
    with pytest.raises(mrhyde_amd.MhaError) as e:
        blk.add_boundary_group("top", mrhyde_amd.BC_NEUMANN, be + 100, bs)
    assert e.value.code == 1
    gid = blk.add_boundary_group("top", mrhyde_amd.BC_NEUMANN, be, bs)
    with pytest.raises(mrhyde_amd.MhaError) as e:  # views before update
        blk.boundary_view(gid, "wts side")
    assert e.value.code == 2
    blk.boundary_update(gid)
    with pytest.raises(mrhyde_amd.MhaError) as e:
        blk.boundary_view(gid, "n[z]")
    assert e.value.code == 4
    u = torch.zeros(m["ndof"], dtype=torch.float64, device="cuda")
    res, vals = torch.zeros_like(u), torch.zeros(len(blk.get_graph()[1]), dtype=torch.float64, device="cuda")
    with pytest.raises(mrhyde_amd.MhaError) as e:  # "Neumann e top" never defined (functionManager.cpp:573)
        blk.assemble_boundary(u, res, vals)
    assert e.value.code == 1 and "Neumann e top" in str(e.value)
    blk.set_function("Neumann e top", 1.0)
    with pytest.raises(mrhyde_amd.MhaError) as e:  # boundary terms accumulate
        blk.assemble_boundary(u, res, vals, flags=3)
    assert e.value.code == 1
    with pytest.raises(mrhyde_amd.MhaError) as e:
        blk.set_physics_parameter("no such", 1.0)
    assert e.value.code == 1
    blk.assemble_boundary(u, res, vals)
    torch.cuda.synchronize()
    # constant unit flux over the top side: sum of -res = -(-1 * |top|) ... res receives -(-g w N) = +g w N
    assert abs(res.sum().item() - 1.0) < 1e-13


@pytest.mark.parametrize("dim,order,qdeg,ncell", [(2, 2, 4, (4, 3)), (3, 1, 2, (3, 2, 2)), (3, 2, 4, (2, 3, 2))])
def test_thermal_compute_flux_and_interface_branch(oracle, dim, order, qdeg, ncell):
    """thermal::computeFlux (src/physics/thermal.cpp:288-347) on a boundary group of a warped mesh -- the flux view
    (10/h) kappa (lambda - T) + kappa grad T . n and its derivative arrays -- against the formula evaluated with the
    oracle's side views; and the "interface" branch of boundaryResidual (thermal.cpp:227-243) against the oracle's
    weak-Dirichlet restatement fed with the trace values as data."""
    torch = _torch()
    import mrhyde_amd
    m = warped(oracle, dim, order, ncell)
    rng = np.random.default_rng(5)
    u = rng.uniform(-1, 1, m["ndof"])
    name = "right"
    be, bs = oracle.boundary_sides(dim, ncell, name)
    sv = oracle.physical_side_basis(dim, order, qdeg, m["nodes"], be, bs)
    nqs = sv["wts"].shape[1]
    lam = rng.uniform(-1, 1, (len(be), nqs))
    kappa = 1.7
    off = np.asarray(m["offsets"])
    ue = u[m["lids"][be][:, off]]                                         # [k][n], basis-function order
    T = np.einsum("kj,kjq->kq", ue, sv["basis"])
    gn = np.einsum("kjqd,kqd->kjq", sv["basis_grad"], sv["normals"])       # grad N_j . n
    gTn = np.einsum("kj,kjq->kq", ue, gn)
    h = sv["wts"].sum(axis=1) ** (1.0 / (dim - 1))
    flux_ref = (10.0 / h)[:, None] * kappa * (lam - T) + kappa * gTn
    dfdu_ref = kappa * (-(10.0 / h)[:, None, None] * np.transpose(sv["basis"], (0, 2, 1)) + np.transpose(gn, (0, 2, 1)))
    rowptr, colind = oracle.build_graph(m["ndof"], m["lids"])
    blk = make_block(m, dim, order, qdeg, graph=(rowptr, colind))
    blk.set_function("thermal diffusion", kappa)
    gid = blk.add_boundary_group(name, mrhyde_amd.BC_INTERFACE, be, bs)
    lam_d = torch.tensor(lam, device="cuda")
    blk.set_function("aux e " + name, lam_d)
    ud = torch.tensor(u, device="cuda")
    n = m["lids"].shape[1]
    flux = torch.zeros((len(be), nqs), dtype=torch.float64, device="cuda")
    dfdu = torch.zeros((len(be), nqs, n), dtype=torch.float64, device="cuda")
    dfda = torch.zeros((len(be), nqs), dtype=torch.float64, device="cuda")
    blk.compute_flux(gid, ud, flux, dflux_du=dfdu, dflux_daux=dfda)
    torch.cuda.synchronize()
    assert rel_err(flux.cpu().numpy(), flux_ref) < RTOL
    assert rel_err(dfdu.cpu().numpy(), dfdu_ref) < RTOL
    assert rel_err(dfda.cpu().numpy(), np.broadcast_to((10.0 / h)[:, None] * kappa, lam.shape)) < RTOL
    # interface branch of boundaryResidual = weak Dirichlet with the trace as data
    vals_ref, res_ref = np.zeros(rowptr[-1]), np.zeros(m["ndof"])
    oracle.assemble_thermal_boundary(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, be, bs, 2, ("array", lam),
                                     rowptr=rowptr, colind=colind, crs_vals=vals_ref, res=res_ref, diff=kappa)
    res = torch.zeros(m["ndof"], dtype=torch.float64, device="cuda")
    vals = torch.zeros(len(colind), dtype=torch.float64, device="cuda")
    blk.assemble_boundary(ud, res, vals)
    torch.cuda.synchronize()
    assert rel_err(vals.cpu().numpy(), vals_ref) < RTOL and rel_err(res.cpu().numpy(), res_ref) < RTOL
