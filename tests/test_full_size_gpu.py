"""Parity at BASELINE.json's full sizes, through properties that need no oracle run of that size.

Config 2 (64^3 Q2 hexes, 2.1e6 rows, 1.4e8 CRS entries), affine and perturbed meshes:
  * the affine fast path (row-owner kernels) and the general path (element matrices + row gather) are independent
    implementations: their CRS values and residuals must agree to 1e-12 on the same affine mesh;
  * thermal with constant coefficients is linear: scattered residual(u1) - residual(u2) = -J (u1 - u2) on free rows,
    which ties the Jacobian to the residual at full size (the vector receives -res.val(), the matrix +dres/du:
    src/managers/assemblyManager.cpp:4094, 4138);
  * pure diffusion annihilates constants: free rows of J sum to zero and a constant state with no source has zero residual;
  * J is symmetric (x'Jy = y'Jx on random vectors supported on free rows);
  * the CRS rows of an interior vertex dof equal the oracle's on a 4^3 mesh of the same element size (an interior
    stencil does not know how large the mesh is): pins full-size VALUES, not only relations, to the oracle.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-12


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


def _rel(a, b):
    torch = _torch()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


def _crs_rel(a, b, rowptr):
    """Per-entry relative difference of two CRS value arrays on the device: |a_ij - b_ij| / max(|b_ij|, 1e-3 rowmax_i)
    (the north star's "Jacobian entries within 1e-12 relative", with a cancellation floor), beside the array-relative one."""
    torch = _torch()
    nrows = len(rowptr) - 1
    counts = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(nrows, device=a.device), counts)
    rowmax = torch.zeros(nrows, dtype=a.dtype, device=a.device)
    rowmax.scatter_reduce_(0, rows, b.abs(), reduce="amax")
    den = torch.maximum(b.abs(), 1e-3 * rowmax[rows]).clamp_min(1e-300)
    return max(float(((a - b).abs() / den).max()), _rel(a, b))


def _spmv(torch, rowptr, colind, vals, x):
    """y = A x for a CRS matrix, with plain tensor ops (no sparse library in the loop)."""
    nrows = len(rowptr) - 1
    counts = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(nrows, device=x.device), counts)
    y = torch.zeros(nrows, dtype=x.dtype, device=x.device)
    y.index_add_(0, rows, vals * x[colind.to(torch.int64)])
    return y


@pytest.fixture(scope="module")
def config2():
    """64^3 Q2-hex thermal block on the unit cube, Dirichlet rows on the boundary, graph on the device."""
    torch = _torch()
    import mrhyde_amd
    nc, order = 64, 2
    m = mrhyde_amd.mesh_structured(3, order, (nc,) * 3)
    blk = mrhyde_amd.Block(3, order, quadrature=2 * order, workset_size=100)
    blk.set_mesh(m["nodes"], m["lids"], m["offsets"], m["ndof"], m["boundary"])
    blk.set_graph()
    rowptr, colind = blk.get_graph()
    dev = torch.device("cuda")
    return dict(m=m, blk=blk, nc=nc, rowptr=torch.tensor(rowptr, device=dev), colind=torch.tensor(colind, device=dev),
                free=torch.tensor(m["boundary"] == 0, device=dev), nrows=m["ndof"], nnz=len(colind))


def _assemble(torch, c, u, path, source=None):
    import mrhyde_amd
    blk = c["blk"]
    blk.set_function("thermal source", source if source is not None else ("sinprod", 12 * np.pi ** 2, [2 * np.pi] * 3))
    blk.set_function("thermal diffusion", 1.0)
    res = torch.full((c["nrows"],), 3.0, dtype=torch.float64, device="cuda")
    vals = torch.full((c["nnz"],), -2.0, dtype=torch.float64, device="cuda")
    blk.assemble_jacres(u, res, vals, path=path, overwrite=True)
    torch.cuda.synchronize()
    assert blk.info("last_path") == path
    return res, vals


def test_config2_fast_path_equals_general_path(config2):
    torch = _torch()
    import mrhyde_amd
    c = config2
    u = torch.rand(c["nrows"], dtype=torch.float64, device="cuda", generator=torch.Generator("cuda").manual_seed(2)) * 2 - 1
    r_fast, v_fast = _assemble(torch, c, u, mrhyde_amd.PATH_ROW_OWNER)
    assert c["blk"].info("num_affine_elems") == c["m"]["nelem"]
    r_gen, v_gen = _assemble(torch, c, u, mrhyde_amd.PATH_ROW_GATHER)
    assert _crs_rel(v_fast, v_gen, c["rowptr"]) < RTOL
    assert _rel(r_fast, r_gen) < RTOL
    c["vals"], c["u"], c["res"] = v_fast, u, r_fast  # reused below


def test_config2_linearity_row_sums_symmetry(config2):
    torch = _torch()
    import mrhyde_amd
    c = config2
    if "vals" not in c:
        test_config2_fast_path_equals_general_path(c)
    vals, u1, r1 = c["vals"], c["u"], c["res"]
    free = c["free"]
    g = torch.Generator("cuda").manual_seed(3)
    u2 = torch.rand(c["nrows"], dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    r2, _ = _assemble(torch, c, u2, mrhyde_amd.PATH_ROW_OWNER)
    Jd = _spmv(torch, c["rowptr"], c["colind"], vals, u1 - u2)
    lhs = (r1 - r2)[free]
    assert float((lhs + Jd[free]).abs().max() / lhs.abs().max()) < 1e-11  # 125-term sums of O(1) entries
    # fixed rows: zero matrix rows and zero residual entries
    assert float(Jd[~free].abs().max()) == 0.0 and float(r1[~free].abs().max()) == 0.0
    # constants are in the kernel of the diffusion operator
    ones = torch.ones(c["nrows"], dtype=torch.float64, device="cuda")
    rs = _spmv(torch, c["rowptr"], c["colind"], vals, ones)
    assert float(rs.abs().max() / vals.abs().max()) < 1e-12
    r0, _ = _assemble(torch, c, 0.7 * ones, mrhyde_amd.PATH_ROW_OWNER, source=0.0)
    assert float(r0.abs().max()) < 1e-12 * float(vals.abs().max())
    # symmetry on the free rows / columns
    x = torch.rand(c["nrows"], dtype=torch.float64, device="cuda", generator=g) * free
    y = torch.rand(c["nrows"], dtype=torch.float64, device="cuda", generator=g) * free
    a = float(torch.dot(x, _spmv(torch, c["rowptr"], c["colind"], vals, y)))
    b = float(torch.dot(y, _spmv(torch, c["rowptr"], c["colind"], vals, x)))
    assert abs(a - b) < 1e-11 * abs(a)


def test_config2_interior_rows_equal_the_oracle_stencil(config2, oracle):
    """Rows of dofs around the centre of the 64^3 mesh against the oracle on a 4^3 mesh with the same element size."""
    torch = _torch()
    import mrhyde_amd
    c = config2
    if "vals" not in c:
        test_config2_fast_path_equals_general_path(c)
    nc, order, h = c["nc"], 2, 1.0 / c["nc"]
    small = oracle.mesh_structured(3, order, (4, 4, 4))
    small["verts"] = small["verts"] * (4 * h)
    small["nodes"] = np.ascontiguousarray(small["verts"][small["cell2vert"]])
    us = np.zeros(small["ndof"])
    ref = oracle.assemble_thermal(3, order, 2 * order, small["nodes"], small["lids"], small["offsets"], us,
                                  source=("sinprod", 0.0, [1.0, 1.0, 1.0]), diff=1.0)
    P, Ps = order * nc + 1, order * 4 + 1          # dofs per direction (lexicographic numbering, x fastest)
    vals = c["vals"]
    rowptr = c["rowptr"].cpu().numpy()
    for off in [(0, 0, 0), (1, 0, 0), (1, 1, 0), (1, 1, 1)]:  # vertex, edge, face and cell-interior dofs
        i, j, k = (order * nc // 2 + off[0], order * nc // 2 + off[1], order * nc // 2 + off[2])
        r = (k * P + j) * P + i
        i_s, j_s, k_s = (4 + off[0], 4 + off[1], 4 + off[2])  # centre vertex of the small mesh
        r_s = (k_s * Ps + j_s) * Ps + i_s
        big = vals[rowptr[r]:rowptr[r + 1]].cpu().numpy()
        sm = ref["crs_vals"][ref["rowptr"][r_s]:ref["rowptr"][r_s + 1]]
        assert len(big) == len(sm)
        assert np.abs(big - sm).max() < RTOL * np.abs(sm).max()  # same stencil order: both graphs are lexicographic


def test_config2_perturbed_general_paths_agree():
    """Perturbed 64^3 mesh (seed 3): matrix-core element kernel + row gather against the same kernel with
    the atomic CRS scatter, and the linearity relation."""
    torch = _torch()
    import mrhyde_amd
    nc, order = 64, 2
    m = mrhyde_amd.mesh_structured(3, order, (nc,) * 3)
    rng = np.random.default_rng(3)
    v = m["verts"]
    interior = np.all((v > 1e-9) & (1 - v > 1e-9), axis=1)
    v[interior] += 0.15 / nc * rng.uniform(-1, 1, (int(interior.sum()), 3))
    m["nodes"] = np.ascontiguousarray(v[m["cell2vert"]])
    blk = mrhyde_amd.Block(3, order, quadrature=2 * order, workset_size=100)
    blk.set_mesh(m["nodes"], m["lids"], m["offsets"], m["ndof"], m["boundary"])
    blk.set_graph()
    rowptr, colind = blk.get_graph()
    blk.set_function("thermal source", ("sinprod", 12 * np.pi ** 2, [2 * np.pi] * 3))
    blk.set_function("thermal diffusion", 1.0)
    n, nnz = m["ndof"], len(colind)
    g = torch.Generator("cuda").manual_seed(5)
    u1 = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    u2 = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    out = {}
    for path in (mrhyde_amd.PATH_ROW_GATHER, mrhyde_amd.PATH_ELEMENT_ATOMIC):
        res = torch.zeros(n, dtype=torch.float64, device="cuda")
        vals = torch.zeros(nnz, dtype=torch.float64, device="cuda")
        blk.assemble_jacres(u1, res, vals, path=path, overwrite=(path == mrhyde_amd.PATH_ROW_GATHER))
        torch.cuda.synchronize()
        out[path] = (res, vals)
    (r_g, v_g), (r_a, v_a) = out[mrhyde_amd.PATH_ROW_GATHER], out[mrhyde_amd.PATH_ELEMENT_ATOMIC]
    assert _rel(v_a, v_g) < RTOL and _rel(r_a, r_g) < RTOL
    res2 = torch.zeros(n, dtype=torch.float64, device="cuda")
    blk.assemble_jacres(u2, res2, None, compute_jacobian=False, path=mrhyde_amd.PATH_ROW_GATHER, overwrite=True)
    torch.cuda.synchronize()
    dev = torch.device("cuda")
    Jd = _spmv(torch, torch.tensor(rowptr, device=dev), torch.tensor(colind, device=dev), v_g, u1 - u2)
    free = torch.tensor(m["boundary"] == 0, device=dev)
    lhs = (r_g - res2)[free]
    assert float((lhs + Jd[free]).abs().max() / lhs.abs().max()) < 1e-11


def _warp(m):
    v = m["verts"].copy()
    w = v.copy()
    w[:, 0] += 0.06 * np.sin(1.3 * v[:, 1] + 0.4) + 0.04 * v[:, 2] ** 2
    w[:, 1] += 0.05 * np.cos(1.1 * v[:, 0]) * (1 + 0.5 * v[:, 1])
    w[:, 2] += 0.05 * v[:, 0] * v[:, 1] + 0.03 * np.sin(2.0 * v[:, 2])
    m["verts"] = w
    m["nodes"] = np.ascontiguousarray(w[m["cell2vert"]])
    return m


def _multi_block(m, physics, qdeg, fixed=None):
    import mrhyde_amd
    blk = mrhyde_amd.Block(m["dim"], quadrature=qdeg, physics=physics,
                           variables=list(zip(m["types"].tolist(), m["orders"].tolist())))
    blk.set_mesh(m["nodes"], m["lids"], m["offsets"], m["ndof"], fixed)
    blk.set_orientation(m["orient"])
    blk.set_graph()
    return blk


def test_config3_porous_mixed_128_cubed(oracle):
    """porousMixed at BASELINE config 3 (128^3 hexes, p in HVOL + u in HDIV-I1, 8.4e6 rows) on a warped mesh:
    the dedicated element kernel + row gather against the generic point engine with the atomic scatter (independent
    implementations), the linearity relation residual(u1) - residual(u2) = -J (u1 - u2), saddle-point symmetry."""
    torch = _torch()
    import mrhyde_amd
    m = _warp(oracle.mesh_multi(3, (128, 128, 128), [oracle.HVOL, oracle.HDIV], [0, 1]))
    blk = _multi_block(m, "porousMixed", 2)
    for k, v in {"source": ("sinprod", 2.0, [np.pi, 2 * np.pi, np.pi]), "Kinv_xx": 1.3, "Kinv_yy": 0.7, "Kinv_zz": 2.1,
                 "total_mobility": 1.9}.items():
        blk.set_function(k, v)
    rowptr, colind = blk.get_graph()
    n, nnz = m["ndof"], len(colind)
    dev = torch.device("cuda")
    g = torch.Generator("cuda").manual_seed(4)
    u1 = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    u2 = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    r_g = torch.full((n,), 5.0, dtype=torch.float64, device="cuda")
    v_g = torch.full((nnz,), 5.0, dtype=torch.float64, device="cuda")
    blk.assemble_jacres(u1, r_g, v_g, path=mrhyde_amd.PATH_ROW_GATHER, overwrite=True)
    r_e = torch.zeros(n, dtype=torch.float64, device="cuda")
    v_e = torch.zeros(nnz, dtype=torch.float64, device="cuda")
    blk.assemble_jacres(u1, r_e, v_e, path=mrhyde_amd.PATH_POINT_ENGINE)
    torch.cuda.synchronize()
    assert _rel(v_g, v_e) < RTOL and _rel(r_g, r_e) < RTOL
    r2 = torch.zeros(n, dtype=torch.float64, device="cuda")
    blk.assemble_jacres(u2, r2, None, compute_jacobian=False, path=mrhyde_amd.PATH_ROW_GATHER, overwrite=True)
    torch.cuda.synchronize()
    rp, ci = torch.tensor(rowptr, device=dev), torch.tensor(colind, device=dev)
    Jd = _spmv(torch, rp, ci, v_g, u1 - u2)
    d = r_g - r2
    assert float((d + Jd).abs().max() / d.abs().max()) < 1e-11
    x = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    y = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    a, b = float(torch.dot(x, _spmv(torch, rp, ci, v_g, y))), float(torch.dot(y, _spmv(torch, rp, ci, v_g, x)))
    assert abs(a - b) < 1e-11 * max(abs(a), float(v_g.abs().max()))


def test_navierstokes_32_cubed_jacobian_is_the_derivative_of_the_residual(oracle):
    """navierstokes (Q2 velocity / Q1 pressure, 89 dofs per element, the config-4 element at 32^3): the assembled
    Jacobian against a central difference of the assembled residual along a random direction -- the module is
    nonlinear, so this is the full-size stand-in for the Sacado derivative array.  Second-order accurate: the error
    scales with eps^2 ~ 1e-10 relative, far above round-off at this size but far below any wrong entry."""
    torch = _torch()
    import mrhyde_amd
    H = oracle.HGRAD
    m = _warp(oracle.mesh_multi(3, (32, 32, 32), [H] * 4, [2, 1, 2, 2]))
    blk = _multi_block(m, "navierstokes", 4)
    for k, v in {"source ux": 0.3, "source uy": ("sinprod", 1.0, [1.0, 2.0, 0.5]), "source uz": -0.2, "viscosity": 0.05,
                 "density": 1.3}.items():
        blk.set_function(k, v)
    blk.set_physics_parameter("fix_uz_offsets", 1)
    rowptr, colind = blk.get_graph()
    n, nnz = m["ndof"], len(colind)
    dev = torch.device("cuda")
    g = torch.Generator("cuda").manual_seed(5)
    u = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    dlt = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    res = torch.zeros(n, dtype=torch.float64, device="cuda")
    vals = torch.zeros(nnz, dtype=torch.float64, device="cuda")
    blk.assemble_jacres(u, res, vals, path=mrhyde_amd.PATH_ROW_GATHER, overwrite=True)
    eps = 1e-5
    rp_, rm_ = torch.zeros_like(res), torch.zeros_like(res)
    blk.assemble_jacres(u + eps * dlt, rp_, None, compute_jacobian=False, path=mrhyde_amd.PATH_ROW_GATHER, overwrite=True)
    blk.assemble_jacres(u - eps * dlt, rm_, None, compute_jacobian=False, path=mrhyde_amd.PATH_ROW_GATHER, overwrite=True)
    torch.cuda.synchronize()
    Jd = _spmv(torch, torch.tensor(rowptr, device=dev), torch.tensor(colind, device=dev), vals, dlt)
    fd = -(rp_ - rm_) / (2 * eps)  # the vector holds -res.val()
    assert float((fd - Jd).abs().max() / Jd.abs().max()) < 1e-7


def test_config4_navierstokes_64_cubed(oracle):
    """navierstokes at BASELINE config 4 as stated: Q2 velocity / Q1 pressure on 64^3 hexes, 6.7e6 rows, 1.42e9 CRS
    entries (more than 2^30: the guard for the column search's midpoint overflow that killed the first run of this
    size), 16.6 GB of element matrices through the row gather.
    (1) constant state: the CRS rows of the dofs of the centre element equal the ORACLE's rows on a 4^3 mesh of the same
        element size (values, entry by entry: both graphs order a row's columns the same way);
    (2) random state: the Jacobian times a random direction equals the central difference of the residual (1e-7)."""
    torch = _torch()
    import mrhyde_amd
    H = oracle.HGRAD
    nc, types, orders = 64, [H] * 4, [2, 1, 2, 2]
    m = mrhyde_amd.mesh_multi(3, (nc,) * 3, types, orders)
    m.update(types=np.array(types), orders=np.array(orders))
    blk = _multi_block(m, "navierstokes", 4)
    funcs = {"source ux": 0.3, "source uy": -0.1, "source uz": 0.2, "viscosity": 0.05, "density": 1.3}
    for k, v in funcs.items():
        blk.set_function(k, v)
    rowptr, colind = blk.get_graph()
    n, nnz = m["ndof"], len(colind)
    assert nnz > 2 ** 30
    dev = torch.device("cuda")
    const = np.array([0.3, 0.1, -0.2, 0.15])                      # ux, pr, uy, uz
    u = torch.tensor(const[m["dof_var"]], device=dev)
    res = torch.zeros(n, dtype=torch.float64, device=dev)
    vals = torch.zeros(nnz, dtype=torch.float64, device=dev)
    blk.assemble_jacres(u, res, vals, path=mrhyde_amd.PATH_ROW_GATHER, overwrite=True)
    torch.cuda.synchronize()
    small = oracle.mesh_multi(3, (4, 4, 4), types, orders, hi=[4.0 / nc] * 3)
    ref = oracle.assemble_block(small, oracle.PHYS_NAVIERSTOKES, 4, const[small["dof_var"]], funcs=funcs, params=[0, 0, 0])
    e_big = ((nc // 2) * nc + nc // 2) * nc + nc // 2               # element (32, 32, 32): its corner 0 is the mesh centre
    e_small = (2 * 4 + 2) * 4 + 2                                   # element (2, 2, 2) of the 4^3 mesh
    for pos in range(0, m["lids"].shape[1], 7):                     # every 7th dof of the element: all four variables
        r, r_s = int(m["lids"][e_big, pos]), int(small["lids"][e_small, pos])
        big = vals[int(rowptr[r]):int(rowptr[r + 1])].cpu().numpy()
        sm = ref["crs_vals"][ref["rowptr"][r_s]:ref["rowptr"][r_s + 1]]
        assert len(big) == len(sm), (pos, len(big), len(sm))
        assert np.abs(big - sm).max() < RTOL * np.abs(ref["crs_vals"]).max(), pos
    # (2) derivative of the residual
    g = torch.Generator("cuda").manual_seed(7)
    u = torch.rand(n, dtype=torch.float64, device=dev, generator=g) * 2 - 1
    dlt = torch.rand(n, dtype=torch.float64, device=dev, generator=g) * 2 - 1
    blk.assemble_jacres(u, res, vals, path=mrhyde_amd.PATH_ROW_GATHER, overwrite=True)
    eps = 1e-5
    rp_, rm_ = torch.zeros_like(res), torch.zeros_like(res)
    blk.assemble_jacres(u + eps * dlt, rp_, None, compute_jacobian=False, path=mrhyde_amd.PATH_ROW_GATHER, overwrite=True)
    blk.assemble_jacres(u - eps * dlt, rm_, None, compute_jacobian=False, path=mrhyde_amd.PATH_ROW_GATHER, overwrite=True)
    torch.cuda.synchronize()
    rp_t, ci_t = torch.tensor(rowptr, device=dev), torch.tensor(colind, device=dev)
    Jd = torch.zeros(n, dtype=torch.float64, device=dev)
    for lo in range(0, n, 1 << 20):                                  # row chunks: the index tensors of 1.4e9 entries at once would be 30 GB
        hi = min(n, lo + (1 << 20))
        a, b = int(rowptr[lo]), int(rowptr[hi])
        counts = (rp_t[lo + 1:hi + 1] - rp_t[lo:hi]).to(torch.int64)
        rows = torch.repeat_interleave(torch.arange(hi - lo, device=dev), counts)
        Jd[lo:hi].index_add_(0, rows, vals[a:b] * dlt[ci_t[a:b].to(torch.int64)])
    fd = -(rp_ - rm_) / (2 * eps)  # the vector holds -res.val()
    assert float((fd - Jd).abs().max() / Jd.abs().max()) < 1e-7


@pytest.mark.parametrize("roe", [1, 0])
def test_config5_hdg_256_squared_blocks_are_the_derivative_of_the_residual(oracle, roe):
    """shallowwaterHybridized HDG element at BASELINE config 5 (256^2 quads, Q1 interior + HFACE-1 traces, 36 unknowns
    per element): the side blocks [E][36][36] against a central difference of the side residual [E][36] along a random
    direction in (u, lambda), element by element -- Roe-like and max-EV stabilisation."""
    torch = _torch()
    import mrhyde_amd
    H = oracle.HGRAD
    nc = 256
    m = oracle.mesh_multi(2, (nc, nc), [H, H, H], [1, 1, 1])
    blk = mrhyde_amd.Block(2, quadrature=2, physics="shallowwaterHybridized", variables=[(0, 1)] * 3)
    blk.set_mesh(m["nodes"], m["lids"], m["offsets"], m["ndof"])
    blk.set_graph()
    blk.set_physics_parameter("g", 1.0)
    blk.set_physics_parameter("Roe-like stabilization", roe)
    blk.set_physics_parameter("max EV stabilization", 1 - roe)
    rng = np.random.default_rng(6)
    E, n = m["nelem"], m["ndof"]
    # momenta 0.3 +- 0.05 in both directions: the normal velocity on the axis-aligned sides stays away from zero, where
    # the |eigenvalue| of the stabilisation has a kink and a central difference across it means nothing
    u = 0.3 + 0.05 * rng.uniform(-1, 1, n)
    hd = m["dof_var"] == 0
    u[hd] = 1.0 + 0.2 * rng.uniform(0, 1, hd.sum())          # depth stays well above zero
    lam = 0.3 + 0.05 * rng.uniform(-1, 1, (E, 3, 4, 2))
    lam[:, 0] = 1.0 + 0.2 * rng.uniform(0, 1, (E, 4, 2))
    du, dl = rng.uniform(-1, 1, n), rng.uniform(-1, 1, (E, 24))
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")
    res = torch.zeros((E, 36), dtype=torch.float64, device="cuda")
    blocks = torch.zeros((E, 36, 36), dtype=torch.float64, device="cuda")
    blk.swhdg_element_blocks(t(u), t(lam.reshape(E, 24)), res, blocks)
    eps = 1e-6
    rp_, rm_ = torch.zeros_like(res), torch.zeros_like(res)
    blk.swhdg_element_blocks(t(u + eps * du), t(lam.reshape(E, 24) + eps * dl), rp_, None)
    blk.swhdg_element_blocks(t(u - eps * du), t(lam.reshape(E, 24) - eps * dl), rm_, None)
    torch.cuda.synchronize()
    # element direction: 12 interior unknowns flattened (variable, dof), then the 24 traces
    idx = np.stack([m["lids"][:, m["offsets"][m["varptr"][v] + d]] for v in range(3) for d in range(4)], axis=1)
    de = torch.cat([t(du[idx]), t(dl)], dim=1)                 # [E][36]
    Jd = torch.einsum("erc,ec->er", blocks, de)
    fd = -(rp_ - rm_) / (2 * eps)                              # the residual arrays hold -res.val()
    assert float((fd - Jd).abs().max() / Jd.abs().max()) < 1e-6
    assert float(blocks.abs().max()) > 0.0 and bool(torch.isfinite(blocks).all())
