"""GPU parity tests of the L2-projection systems of SURVEY 8(f) rank 4 -- AssemblyManager::setInitial (projection and
nodal forms) and setDirichlet -- against the CPU oracle's restatement, through the C ABI."""
import numpy as np
import pytest

from test_multi_gpu import make_block, rel_err
from test_workset_api_gpu import build, var_orient

pytestmark = pytest.mark.gpu
RTOL = 1e-12


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


def thermal_multi(oracle, dim, order, ncell):
    """The single-variable thermal block as a one-variable `mesh_multi` description (warped)."""
    from test_multi_gpu import warp
    return warp(oracle.mesh_multi(dim, ncell, [oracle.HGRAD], [order]))


def make(oracle, kind, dim, ncell, rng, order=2):
    if kind == "thermal":
        return "thermal", ["e"], [oracle.HGRAD], [order], 2 * order, thermal_multi(oracle, dim, order, ncell)
    return build(oracle, kind, dim, ncell, rng)


def expr_for(v, c, dim):
    return "%g + %g*x - %g*y*x + sin(%g*x + y)" % (0.3 + v, 1.0 + c, 0.5 + v, 2.0 + c) + (" + %g*z*z" % (0.2 + c) if dim == 3 else "")


CASES = [("thermal", 3, (3, 2, 2)), ("thermal", 2, (5, 4)), ("porous", 2, (5, 4)), ("porous", 3, (3, 2, 3)), ("ns", 2, (4, 3)),
         ("ns", 3, (2, 2, 3))]


@pytest.mark.parametrize("lump", [False, True])
@pytest.mark.parametrize("kind,dim,ncell", CASES)
def test_set_initial(oracle, kind, dim, ncell, lump):
    """setInitial (assemblyManager.cpp:1185-1305): rhs = (initial, basis) per variable (vector data for HDIV), mass in CRS
    (lumped or not), and a one on the diagonal of a row nothing touches (fix_zero_rows)."""
    torch = _torch()
    rng = np.random.default_rng(81)
    physics, names, types, orders, qdeg, m = make(oracle, kind, dim, ncell, rng)
    nd = m["ndof"] + 1                                     # one extra row no element touches
    rowptr, colind = oracle.build_graph(m["ndof"], m["lids"])
    rowptr = np.concatenate([rowptr, [rowptr[-1] + 1]]).astype(np.int32)
    colind = np.concatenate([colind, [m["ndof"]]]).astype(np.int32)
    want_rhs, want_vals = np.zeros(nd), np.zeros(len(colind))
    funcs = {}
    for v, (name, typ, order) in enumerate(zip(names, types, orders)):
        pb = oracle.physical_basis_var(dim, typ, order, qdeg, m["nodes"], var_orient(oracle, m, v, typ))
        nc = pb["basis"].shape[3]
        data = np.zeros(pb["wts"].shape + (nc,))
        for c in range(nc):
            fname = "initial " + name + (["[x]", "[y]", "[z]"][c] if typ == oracle.HDIV else "")
            if v == 0 and c == 0:                          # per-point data array for one of them
                data[..., c] = rng.uniform(-1, 1, pb["wts"].shape)
                funcs[fname] = data[..., c].copy()
            else:
                funcs[fname] = expr_for(v, c, dim)
                data[..., c] = [[oracle.eval_expression(funcs[fname], pb["ip"][e, q]) for q in range(pb["ip"].shape[1])]
                                for e in range(pb["ip"].shape[0])]
        off = m["offsets"][m["varptr"][v]:m["varptr"][v + 1]]
        oracle.project_rhs(m["lids"], off, data, pb["basis"], pb["wts"], want_rhs)
    oracle.set_initial_mass(m["lids"], oracle.get_mass(m, qdeg), lump, rowptr, colind, want_vals)
    assert want_vals[-1] == 1.0 and np.abs(want_rhs).max() > 0

    import mrhyde_amd
    blk = mrhyde_amd.Block(dim, quadrature=qdeg, physics=physics, variables=list(zip(m["types"].tolist(), m["orders"].tolist())))
    blk.set_mesh(m["nodes"], m["lids"], m["offsets"], nd)
    blk.set_orientation(m["orient"])
    blk.set_graph(rowptr, colind)
    for fname, f in funcs.items():
        blk.set_function(fname, torch.tensor(f, device="cuda") if isinstance(f, np.ndarray) else f)
    rhs = torch.zeros(nd, dtype=torch.float64, device="cuda")
    vals = torch.zeros(len(colind), dtype=torch.float64, device="cuda")
    blk.set_initial(rhs, vals, lump_mass=lump)
    torch.cuda.synchronize()
    assert rel_err(rhs.cpu().numpy(), want_rhs) < RTOL
    assert rel_err(vals.cpu().numpy(), want_vals) < RTOL
    assert vals[-1].item() == 1.0


def test_set_initial_nodal(oracle):
    """setInitial(set, initial, useadjoint) (:1830-1850): vertex values of "initial <var>" replace the vector entries; HGRAD
    order 1 only, as in the reference."""
    torch = _torch()
    import mrhyde_amd
    rng = np.random.default_rng(82)
    for dim, ncell in ((2, (5, 4)), (3, (3, 2, 2))):
        physics, names, types, orders, qdeg, m = make(oracle, "thermal", dim, ncell, rng, order=1)
        expr = expr_for(1, 0, dim)
        nodal = np.array([[oracle.eval_expression(expr, x) for x in xe] for xe in m["nodes"]])
        want = np.full(m["ndof"], -7.0)
        oracle.set_initial_nodal(m["lids"], m["offsets"], nodal, want)
        blk = make_block(m, physics, qdeg)
        blk.set_function("initial e", expr)
        got = torch.full((m["ndof"],), -7.0, dtype=torch.float64, device="cuda")
        blk.set_initial_nodal(got)
        assert rel_err(got.cpu().numpy(), want) < RTOL and not np.any(want == -7.0)
    physics, names, types, orders, qdeg, m = make(oracle, "thermal", 2, (3, 3), rng, order=2)
    blk = make_block(m, physics, qdeg)
    blk.set_function("initial e", 1.0)
    with pytest.raises(mrhyde_amd.MhaError):
        blk.set_initial_nodal(torch.zeros(m["ndof"], dtype=torch.float64, device="cuda"))


DIRICHLET_CASES = [  # block kind, dim, ncell, [(variable, side)], data
    ("thermal", 2, (5, 4), [("e", "top"), ("e", "left")]),
    ("thermal", 3, (3, 2, 2), [("e", "right")]),
    ("porous", 2, (5, 4), [("u", "right"), ("u", "bottom")]),
    ("porous", 3, (3, 2, 3), [("u", "front")]),
    ("ns", 2, (4, 3), [("ux", "left"), ("uy", "left"), ("pr", "top")]),
    ("ns", 3, (2, 2, 3), [("uz", "back"), ("ux", "top")]),
]
SIDE_BIT = {"left": 0, "right": 1, "bottom": 2, "top": 3, "back": 4, "front": 5}


@pytest.mark.parametrize("lump", [False, True])
@pytest.mark.parametrize("kind,dim,ncell,conds", DIRICHLET_CASES)
def test_set_dirichlet(oracle, kind, dim, ncell, conds, lump):
    """setDirichlet (assemblyManager.cpp:1855-1943): boundary mass + data on the fixed rows of the Dirichlet groups (HGRAD
    values, HDIV normal traces), identity on every other row; the lumped form with the reference's column."""
    torch = _torch()
    rng = np.random.default_rng(83)
    physics, names, types, orders, qdeg, m = make(oracle, kind, dim, ncell, rng)
    nd, n_tot = m["ndof"], m["lids"].shape[1]
    rowptr, colind = oracle.build_graph(nd, m["lids"])
    fixed = np.zeros(nd, np.uint8)
    for var, side in conds:                                # isFixedDOF: the variable's dofs on the side set
        v = names.index(var)
        fixed[(m["dof_var"] == v) & ((m["side_mask"] >> SIDE_BIT[side]) & 1 == 1)] = 1
    assert fixed.sum() > 0
    want_rhs, want_vals = np.zeros(nd), np.zeros(len(colind))
    blk = make_block(m, physics, qdeg, fixed=fixed, graph=(rowptr, colind))
    for gi, (var, side) in enumerate(conds):
        v = names.index(var)
        typ, order = types[v], orders[v]
        belem, bside = oracle.boundary_sides(dim, ncell, side)
        sb = oracle.physical_side_basis(dim, max(order, 1), qdeg, m["nodes"], belem, bside)
        nb, nqs = sb["wts"].shape
        if typ == oracle.HGRAD:
            basis, normals = sb["basis"][..., None], None
        else:
            basis = oracle.physical_side_basis_hdiv(dim, qdeg, m["nodes"], belem, bside, var_orient(oracle, m, v, typ))
            normals = sb["normals"]
        if gi == 0:
            dip = rng.uniform(-2, 2, (nb, nqs))
            blk.set_function("Dirichlet %s %s" % (var, side), torch.tensor(dip, device="cuda"))
        else:
            expr = expr_for(v, gi, dim) + " + 0.25*nx"
            dip = np.array([[oracle.eval_expression(expr, sb["ip"][k, q], 0.0, sb["normals"][k, q]) for q in range(nqs)]
                            for k in range(nb)])
            blk.set_function("Dirichlet %s %s" % (var, side), expr)
        off = m["offsets"][m["varptr"][v]:m["varptr"][v + 1]]
        dvals, mass = oracle.dirichlet_boundary(n_tot, off, dip, basis, sb["wts"], normals)
        oracle.set_dirichlet_group(belem, m["lids"], fixed, dvals, mass, lump, rowptr, colind, want_vals, want_rhs)
        blk.add_dirichlet_group(side, var, belem, bside)
    oracle.set_dirichlet_identity(m["lids"], fixed, rowptr, colind, want_vals)
    assert np.abs(want_rhs).max() > 0 and np.abs(want_rhs[fixed == 0]).max() == 0

    rhs = torch.zeros(nd, dtype=torch.float64, device="cuda")
    vals = torch.zeros(len(colind), dtype=torch.float64, device="cuda")
    blk.set_dirichlet(rhs, vals, lump_mass=lump)
    torch.cuda.synchronize()
    assert rel_err(rhs.cpu().numpy(), want_rhs) < RTOL
    assert rel_err(vals.cpu().numpy(), want_vals) < RTOL
    # strong conditions add nothing to the boundary assembly
    res = torch.zeros(nd, dtype=torch.float64, device="cuda")
    blk.assemble_boundary(torch.zeros(nd, dtype=torch.float64, device="cuda"), res, compute_jacobian=False)
    assert res.abs().max().item() == 0.0
