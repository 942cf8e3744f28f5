"""The nonlinear-solve protocol behind the C ABI (mha_newton_*, csrc/newton.{hpp,cpp}): SolverManager::nonlinearSolver
(src/managers/solverManager.cpp:1465-1709) around the GPU assembly, and the strong-Dirichlet lifting of setDirichlet
(:1876-1957).  The linear solves are the caller's: scipy here."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


def _reference_loop(assemble, u0, max_iter, tol):
    """nonlinearSolver restated: autotune residual, |res|_inf, scaled norm against the first, relative tolerance, solve,
    sol += du (solverManager.cpp:1510-1685).  assemble(u) -> (res = -R, J csr)."""
    u, hist, first, it = u0.copy(), [], None, 0
    while True:
        res, J = assemble(u)
        rn = np.abs(res).max()
        if it == 0:
            first = rn
        scaled = 1.0 if it == 0 else rn / first
        hist.append((rn, scaled))
        if scaled < tol or rn < 1e-100:
            it += 1
            break
        u = u + spla.spsolve(J.tocsc(), res)
        it += 1
        if it >= max_iter:
            break
    return u, hist, it


def test_newton_driver_on_a_nonlinear_diffusion_deck(oracle):
    """thermal with 'thermal diffusion' = "1+e*e" (deck string over the solution field), Dirichlet data lifted into the
    start vector: the driver's iteration history (residual norms, scaled norms, iteration count) and the converged
    solution equal the restated loop run on the numpy AD restatement of the assembly."""
    torch = _torch()
    import mrhyde_amd
    from test_thermal_gpu import make_block
    dim, order, qdeg, ncell = 2, 2, 4, (6, 5)
    m = oracle.mesh_structured(dim, order, ncell)
    fixed = m["boundary"]
    rowptr, colind = oracle.build_graph(m["ndof"], m["lids"])
    funcs = {"thermal source": "3*sin(pi*x)*sin(pi*y) + 0.5", "thermal diffusion": "1+e*e"}
    nd = m["ndof"]
    rng = np.random.default_rng(3)
    gD = 0.3 * rng.uniform(-1, 1, nd)  # Dirichlet values (only the fixed rows matter)
    u0 = np.where(fixed != 0, gD, 0.0)

    def assemble(u):
        o = oracle.assemble_thermal_fields(dim, order, qdeg, m["nodes"], m["lids"], m["offsets"], u, funcs, fixed=fixed,
                                           rowptr=rowptr, colind=colind)
        J = sp.csr_matrix((o["crs_vals"], colind, rowptr), shape=(nd, nd)).tolil()
        for r in np.flatnonzero(fixed):
            J[r, r] = 1.0  # dofConstraints: unit diagonal on the fixed rows
        return o["res"], J.tocsr()
    max_iter, tol = 12, 1e-9
    u_ref, hist_ref, it_ref = _reference_loop(assemble, u0, max_iter, tol)
    assert 3 <= it_ref < max_iter

    blk = make_block(m, dim, order, qdeg, fixed=fixed, graph=(rowptr, colind))
    for k, v in funcs.items():
        blk.set_function(k, v)
    u = torch.zeros(nd, dtype=torch.float64, device="cuda")
    blk.dirichlet_lift(u, torch.tensor(gD, device="cuda"))
    assert np.array_equal(u.cpu().numpy(), u0)
    res = torch.zeros(nd, dtype=torch.float64, device="cuda")
    vals = torch.zeros(len(colind), dtype=torch.float64, device="cuda")
    nw = mrhyde_amd.Newton(blk, max_iter=max_iter, nl_tol=tol)
    hist = []
    while True:
        action = nw.step(u, res, vals)
        st = nw.state()
        hist.append((st["resnorm"], st["resnorm_scaled"]))
        if action == mrhyde_amd.NEWTON_DONE:
            break
        assert action == mrhyde_amd.NEWTON_SOLVE
        torch.cuda.synchronize()
        J = sp.csr_matrix((vals.cpu().numpy(), colind, rowptr), shape=(nd, nd))
        du = spla.spsolve(J.tocsc(), res.cpu().numpy())
        nw.update(u, torch.tensor(du, device="cuda"))
    st = nw.state()
    assert st["status"] == 0 and st["iteration"] == it_ref and len(hist) == len(hist_ref)
    for (a, b), (c, d) in zip(hist, hist_ref):
        assert abs(a - c) <= 1e-8 * max(c, hist_ref[0][0] * 1e-12) + 1e-14 and abs(b - d) <= 1e-8 * max(d, 1e-12) + 1e-14
    assert np.abs(u.cpu().numpy() - u_ref).max() < 1e-10 * np.abs(u_ref).max()
    # the finer-grained calls (what a multi-rank caller interleaves with its Export / all-reduce) agree with step()
    nw2 = mrhyde_amd.Newton(blk, max_iter=max_iter, nl_tol=tol)
    u2 = torch.tensor(u0, device="cuda")
    nw2.residual(u2, res)
    rn = nw2.norm(res)
    assert abs(rn - hist_ref[0][0]) < 1e-10 * hist_ref[0][0]
    assert nw2.decide(rn, u2) == mrhyde_amd.NEWTON_SOLVE
    nw2.jacobian(u2, res, vals)
    torch.cuda.synchronize()
    r0, J0 = assemble(u0)
    assert np.abs(res.cpu().numpy() - r0).max() < 1e-12 * np.abs(r0).max()
    assert np.abs(vals.cpu().numpy() - J0.toarray()[np.repeat(np.arange(nd), np.diff(rowptr)), colind]).max() < 1e-12 * np.abs(J0).max()


def test_newton_backtracking_limit_and_lifting(oracle):
    torch = _torch()
    import mrhyde_amd
    from test_thermal_gpu import make_block
    m = oracle.mesh_structured(2, 1, (4, 3))
    blk = make_block(m, 2, 1, 2, fixed=m["boundary"])
    nd = m["ndof"]
    u = torch.zeros(nd, dtype=torch.float64, device="cuda")
    du = torch.full((nd,), 0.25, dtype=torch.float64, device="cuda")
    nw = mrhyde_amd.Newton(blk, max_iter=3, nl_tol=1e-8, allow_backtracking=True)
    assert nw.decide(1.0, u) == mrhyde_amd.NEWTON_SOLVE          # iteration 0: resnorm_first = 1
    nw.update(u, du)
    assert nw.decide(2.0, u) == mrhyde_amd.NEWTON_BACKTRACKED    # scaled norm 2 > 1.1: half a step back
    torch.cuda.synchronize()
    assert np.allclose(u.cpu().numpy(), 0.125) and nw.state()["alpha"] == 0.5
    assert nw.decide(1e-9, u) == mrhyde_amd.NEWTON_DONE and nw.state()["status"] == 0   # converged (relative)
    nw2 = mrhyde_amd.Newton(blk, max_iter=2, nl_tol=1e-8)
    assert nw2.decide(1.0, u) == mrhyde_amd.NEWTON_SOLVE
    nw2.update(u, du)
    assert nw2.decide(0.5, u) == mrhyde_amd.NEWTON_SOLVE
    nw2.update(u, du)
    res = torch.zeros(nd, dtype=torch.float64, device="cuda")
    vals = torch.zeros(len(blk.get_graph()[1]), dtype=torch.float64, device="cuda")
    assert nw2.step(u, res, vals) == mrhyde_amd.NEWTON_DONE and nw2.state()["status"] == 1  # the iteration limit
    v = torch.full((nd,), 9.0, dtype=torch.float64, device="cuda")
    blk.dirichlet_lift(v, None, 2.5)
    torch.cuda.synchronize()
    assert np.array_equal(v.cpu().numpy(), np.where(m["boundary"] != 0, 2.5, 9.0))
