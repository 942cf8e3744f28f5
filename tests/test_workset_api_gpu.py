"""GPU parity tests of the Workset-level accessors and generic conditions of SURVEY 8(b)(v) / a12 / a13 on any block:
per-variable basis views (getBasis / getBasisGrad / getBasisDiv / getBasisSide), solution fields (getSolutionField),
the residual view (getResidual), PhysicsInterface::fluxConditions and thermal's advection term -- each against the CPU
oracle's restatement, through the C ABI."""
import numpy as np
import pytest

from test_multi_gpu import make_block, rel_err, transient_state, warp

pytestmark = pytest.mark.gpu
RTOL = 1e-12


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


def flip_some_faces(m, rng, var):
    """Flip orientation signs consistently per global face dof: the views must take the caller's signs verbatim."""
    flip = rng.uniform(size=m["ndof"]) < 0.3
    u0, dim = m["varptr"][var], m["dim"]
    for e in range(m["nelem"]):
        for f in range(2 * dim):
            if flip[m["lids"][e, m["offsets"][u0 + f]]]:
                m["orient"][e, u0 + f] *= -1


def seeded(u, tr):
    """Workset::computeSolnTransientSeeded value parts (workset.cpp:589-623): (u_AD.val(), u_dot_AD.val()) per dof."""
    if tr is None:
        return u.copy(), np.zeros_like(u)
    st, A, b, bdf = tr["stage"], tr["butcher_A"], tr["butcher_b"], tr["bdf"]
    up, us = tr["u_prev"], tr["u_stage"]
    alpha_u = A[st, st] / b[st]
    timewt = 1.0 / tr["dt"] / b[st]
    alpha_t = bdf[0] * timewt
    beta_u = (1.0 - alpha_u) * up[:, 0]
    for s in range(st):
        beta_u = beta_u + A[st, s] / b[s] * (us[:, s] - up[:, 0])
    beta_t = sum(bdf[s] * up[:, s - 1] for s in range(1, up.shape[1] + 1)) * timewt
    return alpha_u * u + beta_u, alpha_t * u + beta_t


BLOCKS = {  # physics, variable names, types, orders, quadrature degree
    "porous": ("porousMixed", ["p", "u"], ["HVOL", "HDIV"], [0, 1], 2),
    "ns": ("navierstokes", ["ux", "pr", "uy", "uz"], ["HGRAD"] * 4, [2, 1, 2, 2], 4),
}


def build(oracle, kind, dim, ncell, rng):
    physics, names, types, orders, qdeg = BLOCKS[kind]
    nv = len(names) if kind == "porous" else dim + 1
    names, types, orders = names[:nv], [getattr(oracle, t) for t in types[:nv]], orders[:nv]
    m = warp(oracle.mesh_multi(dim, ncell, types, orders))
    if kind == "porous":
        flip_some_faces(m, rng, 1)
    return physics, names, types, orders, qdeg, m


def var_orient(oracle, m, v, typ):
    return m["orient"][:, m["varptr"][v]:m["varptr"][v + 1]] if typ == oracle.HDIV else None


@pytest.mark.parametrize("kind,dim,ncell", [("porous", 2, (5, 4)), ("porous", 3, (3, 2, 3)), ("ns", 2, (4, 3)), ("ns", 3, (2, 2, 3))])
def test_per_variable_views_fields_and_residual(oracle, kind, dim, ncell):
    """getBasis / getBasisGrad / getBasisDiv per variable (discretizationInterface.cpp:898-1127), getSolutionField
    (workset.cpp:1017-1190 with the transient seeding of :589-623) and getResidual, workset by workset."""
    torch = _torch()
    rng = np.random.default_rng(71)
    physics, names, types, orders, qdeg, m = build(oracle, kind, dim, ncell, rng)
    nd = m["ndof"]
    u = rng.uniform(-1, 1, nd)
    tr = transient_state(rng, nd, u)
    ue, ud = seeded(u, tr)
    ref = oracle.assemble_block(m, getattr(oracle, "PHYS_POROUS_MIXED" if kind == "porous" else "PHYS_NAVIERSTOKES"), qdeg,
                                u, transient=tr, want_local=True)
    import mrhyde_amd
    blk = mrhyde_amd.Block(dim, quadrature=qdeg, physics=physics, workset_size=5,
                           variables=list(zip([int(t) for t in types], orders)))
    blk.set_mesh(m["nodes"], m["lids"], m["offsets"], nd)
    blk.set_orientation(m["orient"])
    blk.set_graph(ref["rowptr"], ref["colind"])
    blk.set_time_integration(True, 2, 2, 1, tr["dt"], tr["butcher_A"], tr["butcher_b"], tr["bdf"])
    t = lambda a: torch.tensor(a, device="cuda")
    ud_, up_, us_ = t(u), t(tr["u_prev"]), t(tr["u_stage"])
    pbs = [oracle.physical_basis_var(dim, typ, o, qdeg, m["nodes"], var_orient(oracle, m, v, typ))
           for v, (typ, o) in enumerate(zip(types, orders))]
    xyz = ["[x]", "[y]", "[z]"][:dim]
    assert blk.num_worksets() == (m["nelem"] + 4) // 5
    for w in range(blk.num_worksets()):
        e0, e1 = 5 * w, min(5 * w + 5, m["nelem"])
        blk.workset_update(w)
        blk.workset_compute_solution(ud_, up_, us_)
        blk.workset_compute_residual(ud_, True, up_, us_)
        for v, (name, typ) in enumerate(zip(names, types)):
            pb = pbs[v]
            off = m["offsets"][m["varptr"][v]:m["varptr"][v + 1]]
            rows = m["lids"][e0:e1][:, off]                   # [ne][card]
            b = blk.workset_view_numpy("basis " + name)
            assert b.shape == pb["basis"][e0:e1].shape and rel_err(b, pb["basis"][e0:e1]) < RTOL, name
            if typ == oracle.HGRAD:
                g = blk.workset_view_numpy("basis_grad " + name)
                assert rel_err(g, pb["grad"][e0:e1]) < RTOL, name
                assert rel_err(blk.workset_view_numpy(name), np.einsum("ef,efq->eq", ue[rows], pb["basis"][e0:e1, :, :, 0])) < RTOL
                assert rel_err(blk.workset_view_numpy(name + "_t"), np.einsum("ef,efq->eq", ud[rows], pb["basis"][e0:e1, :, :, 0])) < RTOL
                for d in range(dim):
                    want = np.einsum("ef,efq->eq", ue[rows], pb["grad"][e0:e1, :, :, d])
                    assert rel_err(blk.workset_view_numpy("grad(" + name + ")" + xyz[d]), want) < RTOL
                with pytest.raises(mrhyde_amd.MhaError):
                    blk.workset_view("basis_div " + name)
            elif typ == oracle.HVOL:
                assert rel_err(blk.workset_view_numpy(name), np.einsum("ef,efq->eq", ue[rows], pb["basis"][e0:e1, :, :, 0])) < RTOL
            else:
                dv = blk.workset_view_numpy("basis_div " + name)
                assert rel_err(dv, pb["div"][e0:e1]) < RTOL
                for d in range(dim):
                    want = np.einsum("ef,efq->eq", ue[rows], pb["basis"][e0:e1, :, :, d])
                    assert rel_err(blk.workset_view_numpy(name + xyz[d]), want) < RTOL
                    want = np.einsum("ef,efq->eq", ud[rows], pb["basis"][e0:e1, :, :, d])
                    assert rel_err(blk.workset_view_numpy(name + "_t" + xyz[d]), want) < RTOL
                assert rel_err(blk.workset_view_numpy("div(" + name + ")"), np.einsum("ef,efq->eq", ue[rows], pb["div"][e0:e1])) < RTOL
        # getResidual: res(e,i).val() and res(e,i).dx(j); the oracle's local arrays follow updateRes / updateJac
        assert rel_err(blk.workset_view_numpy("res"), -ref["local_res"][e0:e1]) < RTOL
        assert rel_err(blk.workset_view_numpy("res.dx"), ref["local_J"][e0:e1]) < RTOL
    with pytest.raises(mrhyde_amd.MhaError):
        blk.workset_view("basis nosuchvar")
    blk.workset_update(0)
    with pytest.raises(mrhyde_amd.MhaError):      # fields / residual belong to the workset they were computed on
        blk.workset_view("res")


def test_single_variable_block_has_the_named_views_too(oracle):
    torch = _torch()
    import mrhyde_amd
    dim, order, qdeg, ncell = 2, 2, 4, (4, 3)
    m = warp(oracle.mesh_multi(dim, ncell, [oracle.HGRAD], [order]))
    rng = np.random.default_rng(3)
    u = rng.uniform(-1, 1, m["ndof"])
    funcs = {"thermal source": 0.7, "thermal diffusion": 1.3}
    ref = oracle.assemble_block(m, oracle.PHYS_THERMAL, qdeg, u, funcs=funcs, want_local=True)
    blk = mrhyde_amd.Block(dim, order, quadrature=qdeg, workset_size=0)
    blk.set_mesh(m["nodes"], m["lids"], m["offsets"], m["ndof"])
    blk.set_graph(ref["rowptr"], ref["colind"])
    for k, v in funcs.items():
        blk.set_function(k, v)
    pb = oracle.physical_basis_var(dim, oracle.HGRAD, order, qdeg, m["nodes"])
    blk.workset_update(0)
    ud = torch.tensor(u, device="cuda")
    blk.workset_compute_solution(ud)
    blk.workset_compute_residual(ud)
    assert rel_err(blk.workset_view_numpy("basis e"), pb["basis"]) < RTOL
    assert rel_err(blk.workset_view_numpy("basis_grad e"), pb["grad"]) < RTOL
    rows = m["lids"][:, m["offsets"]]
    assert rel_err(blk.workset_view_numpy("e"), np.einsum("ef,efq->eq", u[rows], pb["basis"][..., 0])) < RTOL
    assert rel_err(blk.workset_view_numpy("grad(e)[y]"), np.einsum("ef,efq->eq", u[rows], pb["grad"][..., 1])) < RTOL
    assert np.abs(blk.workset_view_numpy("e_t")).max() == 0.0       # steady: e_t stays 0 (workset.cpp:942-945)
    assert rel_err(blk.workset_view_numpy("res"), -ref["local_res"]) < RTOL
    assert rel_err(blk.workset_view_numpy("res.dx"), ref["local_J"]) < RTOL


FLUX_CASES = [  # block kind, dim, ncell, variable, side, flux data
    ("porous", 2, (5, 4), "p", "right", "expr"),
    ("porous", 2, (5, 4), "u", "top", "array"),
    ("porous", 3, (3, 2, 3), "u", "front", "expr"),
    ("ns", 2, (4, 3), "pr", "left", "array"),
    ("ns", 3, (2, 2, 3), "uy", "bottom", "expr"),
]


@pytest.mark.parametrize("kind,dim,ncell,var,side,data", FLUX_CASES)
def test_flux_condition_any_variable(oracle, kind, dim, ncell, var, side, data):
    """PhysicsInterface::fluxConditions (physicsInterface.cpp:1702-1762): res(elem, off(dof)) += -flux wts basis(.,0) for
    a "Flux" variable of any block -- HVOL, HDIV (component 0 of the Piola-mapped, oriented side basis, as the reference
    has it), HGRAD of either order -- plus the per-variable side views it reads."""
    torch = _torch()
    rng = np.random.default_rng(72)
    physics, names, types, orders, qdeg, m = build(oracle, kind, dim, ncell, rng)
    v = names.index(var)
    typ, order = types[v], orders[v]
    belem, bside = oracle.boundary_sides(dim, ncell, side)
    sb = oracle.physical_side_basis(dim, max(order, 1), qdeg, m["nodes"], belem, bside)   # wts / normals / ip (+ HGRAD basis)
    nb, nqs = sb["wts"].shape
    if typ == oracle.HGRAD:
        basis = sb["basis"][..., None]
    elif typ == oracle.HVOL:
        basis = np.ones((nb, 1, nqs, 1))
    else:
        basis = oracle.physical_side_basis_hdiv(dim, qdeg, m["nodes"], belem, bside, var_orient(oracle, m, v, typ))
    expr = "1.5 + x*nx - 2*y*ny + 0.5*sin(3*x+y)" + (" + z*nz" if dim == 3 else "")
    if data == "expr":
        flux = np.array([[oracle.eval_expression(expr, sb["ip"][k, q], 0.0, sb["normals"][k, q]) for q in range(nqs)]
                         for k in range(nb)])
    else:
        flux = rng.uniform(-2, 2, (nb, nqs))
    fixed = np.zeros(m["ndof"], np.uint8)
    off = m["offsets"][m["varptr"][v]:m["varptr"][v + 1]]
    fixed[m["lids"][belem[0], off[0]]] = 1               # one fixed row on the side: the scatter must skip it
    want = np.zeros(m["ndof"])
    oracle.flux_condition(belem, m["lids"], off, flux, sb["wts"], basis, want, fixed=fixed)
    assert np.abs(want).max() > 0

    blk = make_block(m, physics, qdeg, fixed=fixed)
    blk.set_function("Flux %s %s" % (var, side), expr if data == "expr" else torch.tensor(flux, device="cuda"))
    gid = blk.add_flux_group(side, var, belem, bside)
    blk.boundary_update(gid)
    assert rel_err(blk.boundary_view_numpy(gid, "basis side " + var), basis) < RTOL
    assert rel_err(blk.boundary_view_numpy(gid, "wts side"), sb["wts"]) < RTOL
    if typ == oracle.HGRAD:
        assert rel_err(blk.boundary_view_numpy(gid, "basis_grad side " + var), sb["basis_grad"]) < RTOL
    res = torch.zeros(m["ndof"], dtype=torch.float64, device="cuda")
    blk.assemble_boundary(torch.zeros(m["ndof"], dtype=torch.float64, device="cuda"), res, compute_jacobian=False)
    torch.cuda.synchronize()
    assert rel_err(res.cpu().numpy(), want) < RTOL
    import mrhyde_amd
    with pytest.raises(mrhyde_amd.MhaError):
        blk.add_flux_group(side, "nosuchvar", belem, bside)


def test_flux_condition_thermal_equals_neumann(oracle):
    """On the single-variable thermal block "Flux e <side>" is the same arithmetic as thermal's own Neumann branch."""
    torch = _torch()
    import mrhyde_amd
    dim, order, qdeg, ncell = 3, 2, 4, (3, 2, 2)
    m = oracle.mesh_structured(dim, order, ncell)
    belem, bside = oracle.boundary_sides(dim, ncell, "right")
    u = np.zeros(m["ndof"])
    rowptr, colind = oracle.build_graph(m["ndof"], m["lids"])
    want = np.zeros(m["ndof"])
    oracle.assemble_thermal_boundary(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, belem, bside,
                                     oracle.BC_NEUMANN, ("const", 2.5), rowptr=rowptr, colind=colind, res=want,
                                     compute_jacobian=False)
    blk = mrhyde_amd.Block(dim, order, quadrature=qdeg)
    blk.set_mesh(m["nodes"], m["lids"], m["offsets"], m["ndof"])
    blk.set_graph(rowptr, colind)
    blk.set_function("Flux e right", 2.5)
    blk.add_flux_group("right", "e", belem, bside)
    res = torch.zeros(m["ndof"], dtype=torch.float64, device="cuda")
    blk.assemble_boundary(torch.tensor(u, device="cuda"), res, compute_jacobian=False)
    assert np.abs(want).max() > 0 and rel_err(res.cpu().numpy(), want) < RTOL


@pytest.mark.parametrize("dim,order,qdeg,ncell,transient", [(2, 1, 2, (6, 5), False), (2, 3, 6, (3, 3), True),
                                                            (3, 2, 4, (3, 2, 2), True)])
def test_thermal_advection(oracle, dim, order, qdeg, ncell, transient):
    """settings "include advection" (thermal.cpp:39): (b . grad e, v) with b = ("bx","by","bz") (thermal.cpp:150-160)."""
    torch = _torch()
    import mrhyde_amd
    from test_thermal_gpu import perturbed
    rng = np.random.default_rng(73)
    m = perturbed(oracle, dim, order, ncell, seed=5)
    nd = m["ndof"]
    u = rng.uniform(-1, 1, nd)
    fixed = m["boundary"]
    b = [0.8, -1.3, 0.6][:dim]
    tr = transient_state(rng, nd, u) if transient else None
    ref = oracle.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, fixed=fixed, transient=tr,
                                  rho=1.3, cp=0.7, diff=0.9, source=("const", 0.4), advection=b)
    plain = oracle.assemble_thermal(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, fixed=fixed, transient=tr,
                                    rho=1.3, cp=0.7, diff=0.9, source=("const", 0.4))
    assert rel_err(plain["crs_vals"], ref["crs_vals"]) > 1e-3       # the term is there (and makes J non-symmetric)
    blk = mrhyde_amd.Block(dim, order, quadrature=qdeg)
    blk.set_mesh(m["nodes"], m["lids"], m["offsets"], nd, fixed)
    blk.set_graph(ref["rowptr"], ref["colind"])
    for k, v in {"thermal source": 0.4, "thermal diffusion": 0.9, "density": 1.3, "specific heat": 0.7}.items():
        blk.set_function(k, v)
    for k, v in zip(("bx", "by", "bz"), b):
        blk.set_function(k, v)
    blk.set_physics_parameter("include advection", 1)
    kw = {}
    t = lambda a: torch.tensor(a, device="cuda")
    if tr is not None:
        blk.set_time_integration(True, 2, 2, 1, tr["dt"], tr["butcher_A"], tr["butcher_b"], tr["bdf"])
        kw = dict(u_prev=t(tr["u_prev"]), u_stage=t(tr["u_stage"]))
    for path in (mrhyde_amd.PATH_AUTO, mrhyde_amd.PATH_POINT_ENGINE, mrhyde_amd.PATH_LOCAL_THEN_SCATTER):
        res = torch.zeros(nd, dtype=torch.float64, device="cuda")
        vals = torch.zeros(len(ref["colind"]), dtype=torch.float64, device="cuda")
        blk.assemble_jacres(t(u), res, vals, path=path, **kw)
        torch.cuda.synchronize()
        assert rel_err(vals.cpu().numpy(), ref["crs_vals"]) < RTOL, path
        assert rel_err(res.cpu().numpy(), ref["res"]) < RTOL, path
    assert blk.info("last_path") == mrhyde_amd.PATH_LOCAL_THEN_SCATTER
    with pytest.raises(mrhyde_amd.MhaError):      # the fused affine kernels do not have the term: refused, not ignored
        blk.assemble_jacres(t(u), res, vals, path=mrhyde_amd.PATH_ROW_OWNER, **kw)
    # switched off again, the block is back on its fast paths
    blk.set_physics_parameter("include advection", 0)
    res = torch.zeros(nd, dtype=torch.float64, device="cuda")
    vals = torch.zeros(len(ref["colind"]), dtype=torch.float64, device="cuda")
    blk.assemble_jacres(t(u), res, vals, **kw)
    assert rel_err(vals.cpu().numpy(), plain["crs_vals"]) < RTOL


MASS_CASES = [  # variable types, orders, dim, ncell, quadrature, physics
    (["HGRAD"], [2], 2, (6, 4), 4, "thermal"),
    (["HVOL", "HDIV"], [0, 1], 3, (4, 2, 3), 2, "porousMixed"),
    (["HGRAD"] * 3, [2, 1, 2], 2, (4, 2), 4, "navierstokes"),
]


@pytest.mark.parametrize("tnames,orders,dim,ncell,qdeg,physics", MASS_CASES)
def test_mass_matrix_free_database_and_sparse3d(oracle, tnames, orders, dim, ncell, qdeg, physics):
    """AssemblyManager::applyMassMatrixFree in its four forms (assemblyManager.cpp:1582-1778), Sparse3DView
    (sparse3DView.hpp:21-161) and the basis database (identifyVolumetricDatabase :4314-4467, exact matching) against the
    oracle's restatements: a warped mesh for the on-the-fly / stored forms (HDIV with flipped signs), a mesh of two
    bitwise-repeated element shapes for the database forms."""
    torch = _torch()
    import mrhyde_amd
    rng = np.random.default_rng(81)
    types = [getattr(oracle, t) for t in tnames]
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")
    w = [1.7, 0.6, 1.2][:len(types)]

    def block(m):
        blk = mrhyde_amd.Block(dim, quadrature=qdeg, physics=physics, variables=list(zip([int(x) for x in types], orders)))
        blk.set_mesh(m["nodes"], m["lids"], m["offsets"], m["ndof"])
        blk.set_orientation(m["orient"])
        return blk

    # --- warped mesh: on the fly and stored element masses
    m = warp(oracle.mesh_multi(dim, ncell, types, orders))
    if oracle.HDIV in types:
        flip_some_faces(m, rng, types.index(oracle.HDIV))
    nd, n, E = m["ndof"], m["n_tot"], m["nelem"]
    x, y0 = rng.uniform(-1, 1, nd), rng.uniform(-1, 1, nd)
    want = y0.copy()
    oracle.apply_mass_matrix_free(m, qdeg, x, want, w)
    blk = block(m)
    y = t(y0)
    blk.apply_mass_matrix_free(t(x), y, mrhyde_amd.MASS_ON_THE_FLY, masswts=w)
    torch.cuda.synchronize()
    assert rel_err(y.cpu().numpy(), want) < RTOL
    mass = torch.zeros((E, n, n), dtype=torch.float64, device="cuda")
    blk.get_mass(mass, w)
    y = t(y0)
    blk.apply_mass_matrix_free(t(x), y, mrhyde_amd.MASS_LOCAL, mass=mass)
    torch.cuda.synchronize()
    assert rel_err(y.cpu().numpy(), want) < RTOL
    with pytest.raises(mrhyde_amd.MhaError):       # database forms need the database
        blk.apply_mass_matrix_free(t(x), y, mrhyde_amd.MASS_DATABASE, mass=mass)

    # --- two element shapes, bitwise repeated (+ one perturbed element): database, dense and sparse.  Exact matching
    # wants bitwise repeats: a mesh width that is a power of two (offsets like 1/6 differ in their last bits from
    # element to element; the reference's 1e-10 tolerance would merge those, the exact database keeps them apart)
    m = oracle.mesh_multi(dim, (8, 4) if dim == 2 else (4, 2, 2), types, orders)
    nd, n, E = m["ndof"], m["n_tot"], m["nelem"]
    x, y0 = rng.uniform(-1, 1, nd), rng.uniform(-1, 1, nd)
    v = m["verts"].copy()
    right = v[:, 0] > 0.5
    v[right, 0] = 0.5 + 1.5 * (v[right, 0] - 0.5)
    v[m["cell2vert"][E - 1][-1]] += 0.03           # the last vertex of the last element: one more shape
    m["verts"] = v
    m["nodes"] = np.ascontiguousarray(v[m["cell2vert"]])
    blk = block(m)
    idx, fu = blk.database_build()
    oidx, ofu = oracle.identify_database(m, qdeg, 1e-10)
    assert np.array_equal(idx, oidx) and np.array_equal(fu, ofu) and 2 < len(fu) <= 8
    want = y0.copy()
    oracle.apply_mass_matrix_free(m, qdeg, x, want, w)
    mass = torch.zeros((E, n, n), dtype=torch.float64, device="cuda")
    blk.get_mass(mass, w)
    dbmass = mass[torch.tensor(fu.astype(np.int64), device="cuda")].contiguous()
    y = t(y0)
    blk.apply_mass_matrix_free(t(x), y, mrhyde_amd.MASS_DATABASE, mass=dbmass)
    torch.cuda.synchronize()
    assert rel_err(y.cpu().numpy(), want) < RTOL
    s3 = mrhyde_amd.Sparse3D(dbmass, 1e-12)
    ovals, ocols, onnz, ome = oracle.sparse3d(dbmass.cpu().numpy(), 1e-12)
    vals, cols, nnz = s3.numpy()
    assert s3.maxent == ome and np.array_equal(nnz, onnz) and s3.size() == int(onnz.sum())
    keep = np.arange(ome)[None, None, :] < onnz[:, :, None]
    assert np.array_equal(cols[keep], ocols[keep]) and np.array_equal(vals[keep], ovals[keep])
    assert ome < n or len(types) == 1            # block-diagonal masses compress
    y = t(y0)
    blk.apply_mass_matrix_free(t(x), y, mrhyde_amd.MASS_DATABASE_SPARSE, sparse=s3)
    torch.cuda.synchronize()
    assert rel_err(y.cpu().numpy(), want) < RTOL
    s3.close()


@pytest.mark.parametrize("kind,dim,ncell,transient", [("porous", 2, (5, 4), False), ("porous", 3, (3, 2, 3), True), ("ns", 2, (4, 3), False)])
def test_compute_flux_of_the_multi_variable_modules(oracle, kind, dim, ncell, transient):
    """porousMixed::computeFlux (src/physics/porousMixed.cpp:440-500): flux(elem, auxp, pt) = u . n at the side points of a
    boundary group, u from the Piola-mapped HDIV side basis with the caller's orientation signs, and its derivative with
    respect to the element's unknowns -- against the formula evaluated with the oracle's side views (warped mesh, flipped
    faces, 2-stage tableau when transient).  navierstokes::computeFlux is empty in the reference (navierstokes.cpp:1016-1018):
    the view stays zero."""
    torch = _torch()
    import mrhyde_amd
    rng = np.random.default_rng(73)
    physics, names, types, orders, qdeg, m = build(oracle, kind, dim, ncell, rng)
    side = "top"
    belem, bside = oracle.boundary_sides(dim, ncell, side)
    sb = oracle.physical_side_basis(dim, 1, qdeg, m["nodes"], belem, bside)
    nb, nqs = sb["wts"].shape
    u = rng.uniform(-1, 1, m["ndof"])
    tr, kw = None, {}
    if transient:
        A, b, bdf = np.array([[0.5, 0.0], [0.3, 0.7]]), np.array([0.4, 0.6]), np.array([1.5, -2.0, 0.5])
        tr = dict(u_prev=rng.uniform(-1, 1, (m["ndof"], 2)), u_stage=rng.uniform(-1, 1, (m["ndof"], 2)), stage=1,
                  butcher_A=A, butcher_b=b, bdf=bdf, dt=0.05)
    blk = make_block(m, physics, qdeg)
    if tr is not None:
        blk.set_time_integration(True, 2, 2, 1, tr["dt"], tr["butcher_A"], tr["butcher_b"], tr["bdf"])
        kw = dict(u_prev=torch.tensor(tr["u_prev"], device="cuda"), u_stage=torch.tensor(tr["u_stage"], device="cuda"))
    gid = blk.add_boundary_group(side, mrhyde_amd.BC_WEAK_DIRICHLET, belem, bside)
    n = m["lids"].shape[1]
    flux = torch.full((nb, nqs), 7.0, dtype=torch.float64, device="cuda")
    dfdu = torch.full((nb, nqs, n), 7.0, dtype=torch.float64, device="cuda")
    blk.compute_flux(gid, torch.tensor(u, device="cuda"), flux, dflux_du=dfdu, **kw)
    torch.cuda.synchronize()
    if kind == "ns":
        assert float(flux.abs().max()) == 0.0 and float(dfdu.abs().max()) == 0.0
        return
    v = names.index("u")
    hd = oracle.physical_side_basis_hdiv(dim, qdeg, m["nodes"], belem, bside, var_orient(oracle, m, v, oracle.HDIV))  # [nb][card][nqs][dim]
    vn = np.einsum("kjqd,kqd->kjq", hd, sb["normals"])
    off = m["offsets"][m["varptr"][v]:m["varptr"][v + 1]]
    ue_all, _ = seeded(u, tr)
    ue = ue_all[m["lids"][belem][:, off]]
    alpha_u = 1.0 if tr is None else tr["butcher_A"][1, 1] / tr["butcher_b"][1]
    assert rel_err(flux.cpu().numpy(), np.einsum("kj,kjq->kq", ue, vn)) < RTOL
    want = np.zeros((nb, nqs, n))
    want[:, :, m["varptr"][v]:m["varptr"][v + 1]] = alpha_u * np.transpose(vn, (0, 2, 1))
    assert rel_err(dfdu.cpu().numpy(), want) < RTOL
