"""ctypes binding of the CPU oracle (oracle/libmrhyde_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (mrhyde_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(_ORACLE_DIR, "libmrhyde_oracle.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_up = C.POINTER(C.c_ubyte)


class ThermalArgs(C.Structure):
    _fields_ = [
        ("dim", C.c_int), ("order", C.c_int), ("qdeg", C.c_int),
        ("nelem", C.c_int), ("nrows", C.c_int), ("workset_size", C.c_int),
        ("nodes", _dp), ("lids", _ip), ("offsets", _ip), ("fixed", _up), ("u", _dp),
        ("basis", _dp), ("basis_grad", _dp), ("wts", _dp), ("ip", _dp),
        ("transient", C.c_int), ("nsteps", C.c_int), ("nstages", C.c_int), ("stage", C.c_int),
        ("u_prev", _dp), ("u_stage", _dp), ("butcher_A", _dp), ("butcher_b", _dp), ("bdf", _dp),
        ("dt", C.c_double),
        ("diff", C.c_double), ("rho", C.c_double), ("cp", C.c_double),
        ("diff_ip", _dp),
        ("source_kind", C.c_int), ("source_amp", C.c_double), ("source_freq", C.c_double * 3),
        ("source_ip", _dp),
        ("compute_jacobian", C.c_int), ("num_threads", C.c_int),
        ("rowptr", _ip), ("colind", _ip),
        ("crs_vals", _dp), ("res", _dp), ("local_J", _dp), ("local_res", _dp),
        ("have_advection", C.c_int), ("adv", C.c_double * 3), ("adv_ip", _dp),
    ]


def build():
    subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_l2_error_sinprod.restype = C.c_double
    return _lib


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(_ip)


def _u(a):
    return None if a is None else a.ctypes.data_as(_up)


def gauss_line(n):
    p = np.zeros(n)
    w = np.zeros(n)
    lib().orc_gauss_line(n, _d(p), _d(w))
    return p, w


def ref_sizes(dim, order, qdeg):
    nb, nq, nn = C.c_int(), C.c_int(), C.c_int()
    rc = lib().orc_ref_sizes(dim, order, qdeg, C.byref(nb), C.byref(nq), C.byref(nn))
    assert rc == 0
    return nb.value, nq.value, nn.value


def ref_tables(dim, order, qdeg):
    nb, nq, nn = ref_sizes(dim, order, qdeg)
    t = dict(ip=np.zeros((nq, dim)), wts=np.zeros(nq), basis=np.zeros((nb, nq)),
             grad=np.zeros((nb, nq, dim)), nodeval=np.zeros((nn, nq)), nodegrad=np.zeros((nn, nq, dim)))
    rc = lib().orc_ref_tables(dim, order, qdeg, _d(t["ip"]), _d(t["wts"]), _d(t["basis"]), _d(t["grad"]),
                              _d(t["nodeval"]), _d(t["nodegrad"]))
    assert rc == 0
    return t


def mesh_structured(dim, order, ncell, lo=None, hi=None):
    ncell = np.asarray(ncell, dtype=np.int32)
    lo = np.zeros(3) if lo is None else np.asarray(lo, dtype=np.float64)
    hi = np.ones(3) if hi is None else np.asarray(hi, dtype=np.float64)
    nv, ne, nd = C.c_int(), C.c_int(), C.c_longlong()
    assert lib().orc_mesh_sizes(dim, order, _i(ncell), C.byref(nv), C.byref(ne), C.byref(nd)) == 0
    n = (order + 1) ** dim
    nn = 2 ** dim
    m = dict(verts=np.zeros((nv.value, dim)), cell2vert=np.zeros((ne.value, nn), dtype=np.int32),
             lids=np.zeros((ne.value, n), dtype=np.int32), offsets=np.zeros(n, dtype=np.int32),
             boundary=np.zeros(nd.value, dtype=np.uint8), ndof=nd.value, nelem=ne.value)
    rc = lib().orc_mesh_structured(dim, order, _i(ncell), _d(lo), _d(hi), _d(m["verts"]), _i(m["cell2vert"]),
                                   _i(m["lids"]), _i(m["offsets"]), _u(m["boundary"]))
    assert rc == 0
    m["nodes"] = np.ascontiguousarray(m["verts"][m["cell2vert"]])
    return m


def physical_basis(dim, order, qdeg, nodes, want=("basis", "basis_grad", "wts", "ip")):
    nb, nq, nn = ref_sizes(dim, order, qdeg)
    E = nodes.shape[0]
    out = dict(basis=np.zeros((E, nb, nq)) if "basis" in want else None,
               basis_grad=np.zeros((E, nb, nq, dim)) if "basis_grad" in want else None,
               wts=np.zeros((E, nq)) if "wts" in want else None,
               ip=np.zeros((E, nq, dim)) if "ip" in want else None)
    nodes = np.ascontiguousarray(nodes, dtype=np.float64)
    rc = lib().orc_physical_basis(dim, order, qdeg, E, _d(nodes), _d(out["basis"]), _d(out["basis_grad"]),
                                  _d(out["wts"]), _d(out["ip"]))
    assert rc == 0
    return out


def build_graph(nrows, lids):
    lids = np.ascontiguousarray(lids, dtype=np.int32)
    E, n = lids.shape
    rowptr = np.zeros(nrows + 1, dtype=np.int32)
    assert lib().orc_build_graph(nrows, E, n, _i(lids), _i(rowptr), None) == 0
    colind = np.zeros(rowptr[-1], dtype=np.int32)
    assert lib().orc_build_graph(nrows, E, n, _i(lids), _i(rowptr), _i(colind)) == 0
    return rowptr, colind


def ad_width(n):
    return lib().orc_ad_width(n)


def assemble_thermal(dim, order, qdeg, nodes, lids, offsets, u, *, nrows=None, fixed=None, pb=None,
                     workset_size=100, transient=None, diff=1.0, rho=1.0, cp=1.0, diff_ip=None,
                     source=("const", 0.0), compute_jacobian=True, num_threads=1,
                     rowptr=None, colind=None, want_crs=True, want_res=True, want_local=False, advection=None):
    """Run the oracle's assembleJacRes restatement.  Returns dict with crs_vals/res/local_J/local_res."""
    nodes = np.ascontiguousarray(nodes, dtype=np.float64)
    lids = np.ascontiguousarray(lids, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    u = np.ascontiguousarray(u, dtype=np.float64)
    E, n = lids.shape
    nrows = u.shape[0] if nrows is None else nrows
    if pb is None:
        pb = physical_basis(dim, order, qdeg, nodes)
    a = ThermalArgs()
    a.dim, a.order, a.qdeg = dim, order, qdeg
    a.nelem, a.nrows, a.workset_size = E, nrows, workset_size
    keep = [nodes, lids, offsets, u, pb]
    a.nodes, a.lids, a.offsets, a.u = _d(nodes), _i(lids), _i(offsets), _d(u)
    if fixed is not None:
        fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
        keep.append(fixed)
        a.fixed = _u(fixed)
    a.basis, a.basis_grad, a.wts, a.ip = _d(pb["basis"]), _d(pb["basis_grad"]), _d(pb["wts"]), _d(pb["ip"])
    if transient is not None:
        t = {k: np.ascontiguousarray(v, dtype=np.float64) if isinstance(v, np.ndarray) else v
             for k, v in transient.items()}
        keep.append(t)
        a.transient = 1
        a.nsteps, a.nstages, a.stage = t["u_prev"].shape[1], t["u_stage"].shape[1], t["stage"]
        a.u_prev, a.u_stage = _d(t["u_prev"]), _d(t["u_stage"])
        a.butcher_A, a.butcher_b, a.bdf = _d(t["butcher_A"]), _d(t["butcher_b"]), _d(t["bdf"])
        a.dt = t["dt"]
    a.diff, a.rho, a.cp = diff, rho, cp
    if diff_ip is not None:
        diff_ip = np.ascontiguousarray(diff_ip, dtype=np.float64)
        keep.append(diff_ip)
        a.diff_ip = _d(diff_ip)
    kind = source[0]
    if kind == "const":
        a.source_kind, a.source_amp = 0, float(source[1])
    elif kind == "array":
        s = np.ascontiguousarray(source[1], dtype=np.float64)
        keep.append(s)
        a.source_kind, a.source_ip = 1, _d(s)
    elif kind == "sinprod":
        a.source_kind, a.source_amp = 2, float(source[1])
        fr = list(source[2]) + [0.0] * (3 - len(source[2]))
        a.source_freq = (C.c_double * 3)(*fr)
    else:
        raise ValueError(kind)
    a.compute_jacobian, a.num_threads = int(compute_jacobian), num_threads
    if advection is not None:      # "include advection": constants (bx, by, bz) or an [E][q][dim] array
        a.have_advection = 1
        if isinstance(advection, np.ndarray) and advection.ndim == 3:
            adv = np.ascontiguousarray(advection, dtype=np.float64)
            keep.append(adv)
            a.adv_ip = _d(adv)
        else:
            bb = list(advection) + [0.0] * (3 - len(advection))
            a.adv = (C.c_double * 3)(*bb)
    out = {}
    if want_crs or want_res:
        if rowptr is None:
            rowptr, colind = build_graph(nrows, lids)
        keep += [rowptr, colind]
        a.rowptr, a.colind = _i(rowptr), _i(colind)
        out["rowptr"], out["colind"] = rowptr, colind
    if want_crs:
        out["crs_vals"] = np.zeros(rowptr[-1])
        a.crs_vals = _d(out["crs_vals"])
    if want_res:
        out["res"] = np.zeros(nrows)
        a.res = _d(out["res"])
    if want_local:
        out["local_J"] = np.zeros((E, n, n))
        out["local_res"] = np.zeros((E, n))
        a.local_J, a.local_res = _d(out["local_J"]), _d(out["local_res"])
    rc = lib().orc_assemble_thermal(C.byref(a))
    assert rc == 0, rc
    return out


def apply_dbc_diag(fixed, rowptr, colind, vals):
    fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
    assert lib().orc_apply_dbc_diag(len(fixed), _u(fixed), _i(rowptr), _i(colind), _d(vals)) == 0


def l2_error_sinprod(dim, order, qdeg, lids, offsets, pb, u, freq):
    lids = np.ascontiguousarray(lids, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    u = np.ascontiguousarray(u, dtype=np.float64)
    fr = np.zeros(3)
    fr[:dim] = freq
    return lib().orc_l2_error_sinprod(dim, order, qdeg, lids.shape[0], _i(lids), _i(offsets), _d(pb["basis"]),
                                      _d(pb["wts"]), _d(pb["ip"]), _d(u), _d(fr))


# ---- boundary (side) terms ------------------------------------------------------------------------------------
class ThermalBndArgs(C.Structure):
    _fields_ = [
        ("dim", C.c_int), ("order", C.c_int), ("qdeg", C.c_int), ("nrows", C.c_int),
        ("nodes", _dp), ("lids", _ip), ("offsets", _ip), ("fixed", _up), ("u", _dp),
        ("transient", C.c_int), ("nsteps", C.c_int), ("nstages", C.c_int), ("stage", C.c_int),
        ("u_prev", _dp), ("u_stage", _dp), ("butcher_A", _dp), ("butcher_b", _dp), ("bdf", _dp),
        ("dt", C.c_double),
        ("nb", C.c_int), ("belem", _ip), ("bside", _ip),
        ("bc_type", C.c_int), ("data_kind", C.c_int), ("data_amp", C.c_double), ("data_freq", C.c_double * 3),
        ("data_ip", _dp),
        ("diff", C.c_double), ("form_param", C.c_double),
        ("compute_jacobian", C.c_int),
        ("rowptr", _ip), ("colind", _ip), ("crs_vals", _dp), ("res", _dp),
    ]


BC_NEUMANN, BC_WEAK_DIRICHLET = 1, 2


def side_sizes(dim, qdeg):
    ns, nqs = C.c_int(), C.c_int()
    assert lib().orc_side_sizes(dim, qdeg, C.byref(ns), C.byref(nqs)) == 0
    return ns.value, nqs.value


def side_tables(dim, order, qdeg):
    nb, _, nn = ref_sizes(dim, order, qdeg)
    ns, nqs = side_sizes(dim, qdeg)
    t = dict(sip=np.zeros((ns, nqs, dim)), swts=np.zeros(nqs), tanU=np.zeros((ns, dim)), tanV=np.zeros((ns, dim)),
             sbasis=np.zeros((ns, nb, nqs)), sgrad=np.zeros((ns, nb, nqs, dim)), snodeval=np.zeros((ns, nn, nqs)),
             snodegrad=np.zeros((ns, nn, nqs, dim)))
    rc = lib().orc_side_tables(dim, order, qdeg, _d(t["sip"]), _d(t["swts"]), _d(t["tanU"]), _d(t["tanV"]),
                               _d(t["sbasis"]), _d(t["sgrad"]), _d(t["snodeval"]), _d(t["snodegrad"]))
    assert rc == 0
    return t


def physical_side_basis(dim, order, qdeg, nodes, belem, bside):
    nb, _, _ = ref_sizes(dim, order, qdeg)
    _, nqs = side_sizes(dim, qdeg)
    nodes = np.ascontiguousarray(nodes, dtype=np.float64)
    belem = np.ascontiguousarray(belem, dtype=np.int32)
    bside = np.ascontiguousarray(bside, dtype=np.int32)
    k = len(belem)
    out = dict(wts=np.zeros((k, nqs)), normals=np.zeros((k, nqs, dim)), ip=np.zeros((k, nqs, dim)),
               basis=np.zeros((k, nb, nqs)), basis_grad=np.zeros((k, nb, nqs, dim)))
    rc = lib().orc_physical_side_basis(dim, order, qdeg, k, _d(nodes), _i(belem), _i(bside), _d(out["wts"]),
                                       _d(out["normals"]), _d(out["ip"]), _d(out["basis"]), _d(out["basis_grad"]))
    assert rc == 0
    return out


def boundary_sides(dim, ncell, which):
    """(element ids, local side ids) of a named side of the structured mesh: left/right (x), bottom/top (y),
    back/front (z, 3-D) -- the panzer SquareQuad/CubeHex side-set names the reference's decks use.  Elements are
    numbered x-fastest; local side ids are shards' (see oracle header)."""
    nc = list(ncell) + [1] * (3 - len(ncell))
    ix, iy, iz = np.meshgrid(np.arange(nc[0]), np.arange(nc[1]), np.arange(nc[2]), indexing="ij")
    eid = (ix + nc[0] * (iy + nc[1] * iz)).astype(np.int32)
    if dim == 2:
        sel = {"bottom": (iy == 0, 0), "right": (ix == nc[0] - 1, 1), "top": (iy == nc[1] - 1, 2), "left": (ix == 0, 3)}
    else:
        sel = {"bottom": (iy == 0, 0), "right": (ix == nc[0] - 1, 1), "top": (iy == nc[1] - 1, 2), "left": (ix == 0, 3),
               "back": (iz == 0, 4), "front": (iz == nc[2] - 1, 5)}
    mask, side = sel[which]
    e = np.sort(eid[mask])
    return e, np.full(len(e), side, dtype=np.int32)


def assemble_thermal_boundary(dim, order, qdeg, nodes, lids, offsets, u, belem, bside, bc_type, data, *, rowptr, colind,
                              crs_vals=None, res=None, fixed=None, transient=None, diff=1.0, form_param=1.0,
                              compute_jacobian=True):
    """Accumulates the boundary contribution of one boundary group into crs_vals / res (both optional)."""
    nodes = np.ascontiguousarray(nodes, dtype=np.float64)
    lids = np.ascontiguousarray(lids, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    u = np.ascontiguousarray(u, dtype=np.float64)
    belem = np.ascontiguousarray(belem, dtype=np.int32)
    bside = np.ascontiguousarray(bside, dtype=np.int32)
    a = ThermalBndArgs()
    a.dim, a.order, a.qdeg, a.nrows = dim, order, qdeg, u.shape[0]
    keep = [nodes, lids, offsets, u, belem, bside]
    a.nodes, a.lids, a.offsets, a.u = _d(nodes), _i(lids), _i(offsets), _d(u)
    if fixed is not None:
        fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
        keep.append(fixed)
        a.fixed = _u(fixed)
    if transient is not None:
        t = {k: np.ascontiguousarray(v, dtype=np.float64) if isinstance(v, np.ndarray) else v
             for k, v in transient.items()}
        keep.append(t)
        a.transient = 1
        a.nsteps, a.nstages, a.stage = t["u_prev"].shape[1], t["u_stage"].shape[1], t["stage"]
        a.u_prev, a.u_stage = _d(t["u_prev"]), _d(t["u_stage"])
        a.butcher_A, a.butcher_b, a.bdf = _d(t["butcher_A"]), _d(t["butcher_b"]), _d(t["bdf"])
        a.dt = t["dt"]
    a.nb, a.belem, a.bside, a.bc_type = len(belem), _i(belem), _i(bside), bc_type
    kind = data[0]
    if kind == "const":
        a.data_kind, a.data_amp = 0, float(data[1])
    elif kind == "array":
        s = np.ascontiguousarray(data[1], dtype=np.float64)
        keep.append(s)
        a.data_kind, a.data_ip = 1, _d(s)
    elif kind == "sinprod":
        a.data_kind, a.data_amp = 2, float(data[1])
        fr = list(data[2]) + [0.0] * (3 - len(data[2]))
        a.data_freq = (C.c_double * 3)(*fr)
    else:
        raise ValueError(kind)
    a.diff, a.form_param, a.compute_jacobian = diff, form_param, int(compute_jacobian)
    a.rowptr, a.colind, a.crs_vals, a.res = _i(rowptr), _i(colind), _d(crs_vals), _d(res)
    rc = lib().orc_assemble_thermal_boundary(C.byref(a))
    assert rc == 0, rc


# ---- multi-variable blocks ----------------------------------------------------------------------------------------
HGRAD, HVOL, HDIV = 0, 1, 2
PHYS_THERMAL, PHYS_POROUS_MIXED, PHYS_NAVIERSTOKES = 1, 2, 3
_sp = C.POINTER(C.c_byte)


class Func(C.Structure):
    _fields_ = [("kind", C.c_int), ("amp", C.c_double), ("freq", C.c_double * 3), ("ip", _dp), ("expr", C.c_char_p),
                ("t", C.c_double)]


class BlockArgs(C.Structure):
    _fields_ = [
        ("dim", C.c_int), ("qdeg", C.c_int), ("nvars", C.c_int),
        ("types", C.c_int * 8), ("orders", C.c_int * 8), ("physics", C.c_int),
        ("nelem", C.c_int), ("nrows", C.c_int),
        ("nodes", _dp), ("lids", _ip), ("offsets", _ip), ("orient", _sp), ("fixed", _up), ("u", _dp),
        ("transient", C.c_int), ("nsteps", C.c_int), ("nstages", C.c_int), ("stage", C.c_int),
        ("u_prev", _dp), ("u_stage", _dp), ("butcher_A", _dp), ("butcher_b", _dp), ("bdf", _dp),
        ("dt", C.c_double),
        ("funcs", Func * 8), ("params", C.c_double * 8),
        ("compute_jacobian", C.c_int),
        ("rowptr", _ip), ("colind", _ip), ("crs_vals", _dp), ("res", _dp), ("local_J", _dp), ("local_res", _dp),
        ("nb", C.c_int), ("belem", _ip), ("bside", _ip), ("bc_type", C.c_int), ("bdata", Func),
        ("aux_ip", _dp), ("farfield_ip", _dp),
    ]


def basis_card(dim, typ, order):
    return lib().orc_basis_card(dim, typ, order)


def mesh_multi(dim, ncell, types, orders, lo=None, hi=None):
    ncell = np.asarray(ncell, dtype=np.int32)
    lo = np.zeros(3) if lo is None else np.asarray(lo, dtype=np.float64)
    hi = np.ones(3) if hi is None else np.asarray(hi, dtype=np.float64)
    types, orders = np.asarray(types, dtype=np.int32), np.asarray(orders, dtype=np.int32)
    nv, ne, nt, nd = C.c_int(), C.c_int(), C.c_int(), C.c_longlong()
    rc = lib().orc_mesh_multi_sizes(dim, _i(ncell), len(types), _i(types), _i(orders), C.byref(nv), C.byref(ne),
                                    C.byref(nt), C.byref(nd))
    assert rc == 0, rc
    nn = 2 ** dim
    m = dict(verts=np.zeros((nv.value, dim)), cell2vert=np.zeros((ne.value, nn), np.int32),
             lids=np.zeros((ne.value, nt.value), np.int32), offsets=np.zeros(nt.value, np.int32),
             orient=np.ones((ne.value, nt.value), np.int8), side_mask=np.zeros(nd.value, np.uint8),
             dof_var=np.zeros(nd.value, np.int32), ndof=nd.value, nelem=ne.value, n_tot=nt.value,
             types=types, orders=orders, dim=dim, ncell=tuple(int(c) for c in ncell))
    rc = lib().orc_mesh_multi(dim, _i(ncell), _d(lo), _d(hi), len(types), _i(types), _i(orders), _d(m["verts"]),
                              _i(m["cell2vert"]), _i(m["lids"]), _i(m["offsets"]), m["orient"].ctypes.data_as(_sp),
                              _u(m["side_mask"]), _i(m["dof_var"]))
    assert rc == 0, rc
    m["nodes"] = np.ascontiguousarray(m["verts"][m["cell2vert"]])
    cards = [basis_card(dim, int(t), int(o)) for t, o in zip(types, orders)]
    m["varptr"] = np.concatenate([[0], np.cumsum(cards)]).astype(np.int32)
    return m


def physical_basis_var(dim, typ, order, qdeg, nodes, orient=None):
    """orient: int8 [E][n] signs of this variable's dofs (HDIV) or None."""
    _, nq, _ = ref_sizes(dim, 1, qdeg)
    n = basis_card(dim, typ, order)
    nodes = np.ascontiguousarray(nodes, dtype=np.float64)
    E = nodes.shape[0]
    nc = dim if typ == HDIV else 1
    out = dict(basis=np.zeros((E, n, nq, nc)), grad=np.zeros((E, n, nq, dim)), div=np.zeros((E, n, nq)),
               wts=np.zeros((E, nq)), ip=np.zeros((E, nq, dim)))
    op, stride = None, 0
    if orient is not None:
        orient = np.ascontiguousarray(orient, dtype=np.int8)
        op, stride = orient.ctypes.data_as(_sp), orient.shape[1]
    rc = lib().orc_physical_basis_var(dim, typ, order, qdeg, E, _d(nodes), op, stride, 0, _d(out["basis"]),
                                      _d(out["grad"]), _d(out["div"]), _d(out["wts"]), _d(out["ip"]))
    assert rc == 0, rc
    return out


def eval_expression(expr, xyz, t=0.0, nrm=None, h=0.0):
    f = lib().orc_eval_expression
    f.restype = C.c_double
    f.argtypes = [C.c_char_p, _dp, C.c_double, _dp, C.c_double, C.POINTER(C.c_int)]
    x = np.zeros(3)
    x[:len(xyz)] = xyz
    n3 = None
    if nrm is not None:
        n3 = np.zeros(3)
        n3[:len(nrm)] = nrm
    err = C.c_int()
    v = f(expr.encode(), _d(x), float(t), _d(n3), float(h), C.byref(err))
    if err.value:
        raise ValueError("bad expression: " + expr)
    return v


def _func(spec, keep):
    f = Func()
    if isinstance(spec, str):  # deck string at time 0
        spec = ("expr", spec, 0.0)
    if isinstance(spec, tuple) and spec[0] == "expr":
        b = spec[1].encode()
        keep.append(b)
        f.kind, f.expr, f.t = 3, b, float(spec[2]) if len(spec) > 2 else 0.0
        return f
    if isinstance(spec, (int, float)):
        f.kind, f.amp = 0, float(spec)
    elif spec[0] == "const":
        f.kind, f.amp = 0, float(spec[1])
    elif spec[0] == "array":
        arr = np.ascontiguousarray(spec[1], dtype=np.float64)
        keep.append(arr)
        f.kind, f.ip = 1, _d(arr)
    elif spec[0] == "sinprod":
        f.kind, f.amp = 2, float(spec[1])
        fr = list(spec[2]) + [0.0] * (3 - len(spec[2]))
        f.freq = (C.c_double * 3)(*fr)
    else:
        raise ValueError(spec)
    return f


PHYS_FUNCS = {
    PHYS_THERMAL: ["thermal source", "thermal diffusion", "specific heat", "density"],
    PHYS_POROUS_MIXED: ["source", "Kinv_xx", "Kinv_yy", "Kinv_zz", "total_mobility"],
    PHYS_NAVIERSTOKES: ["source ux", "source pr", "source uy", "source uz", "density", "viscosity"],
}
PHYS_DEFAULTS = {
    PHYS_THERMAL: [0.0, 1.0, 1.0, 1.0],
    PHYS_POROUS_MIXED: [0.0, 1.0, 1.0, 1.0, 1.0],
    PHYS_NAVIERSTOKES: [0.0, 0.0, 0.0, 0.0, 1.0, 1.0],
}


def _block_args(m, physics, qdeg, u, funcs, params, fixed, transient, compute_jacobian, keep):
    a = BlockArgs()
    a.dim, a.qdeg, a.nvars, a.physics = m["dim"], qdeg, len(m["types"]), physics
    for v, (t, o) in enumerate(zip(m["types"], m["orders"])):
        a.types[v], a.orders[v] = int(t), int(o)
    nodes = np.ascontiguousarray(m["nodes"], dtype=np.float64)
    lids = np.ascontiguousarray(m["lids"], dtype=np.int32)
    offsets = np.ascontiguousarray(m["offsets"], dtype=np.int32)
    orient = np.ascontiguousarray(m["orient"], dtype=np.int8)
    u = np.ascontiguousarray(u, dtype=np.float64)
    keep += [nodes, lids, offsets, orient, u]
    a.nelem, a.nrows = lids.shape[0], u.shape[0]
    a.nodes, a.lids, a.offsets, a.u = _d(nodes), _i(lids), _i(offsets), _d(u)
    a.orient = orient.ctypes.data_as(_sp)
    if fixed is not None:
        fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
        keep.append(fixed)
        a.fixed = _u(fixed)
    if transient is not None:
        t = {k: np.ascontiguousarray(v, dtype=np.float64) if isinstance(v, np.ndarray) else v
             for k, v in transient.items()}
        keep.append(t)
        a.transient = 1
        a.nsteps, a.nstages, a.stage = t["u_prev"].shape[1], t["u_stage"].shape[1], t["stage"]
        a.u_prev, a.u_stage = _d(t["u_prev"]), _d(t["u_stage"])
        a.butcher_A, a.butcher_b, a.bdf = _d(t["butcher_A"]), _d(t["butcher_b"]), _d(t["bdf"])
        a.dt = t["dt"]
    else:
        a.dt = 1.0
    names, defaults = PHYS_FUNCS[physics], PHYS_DEFAULTS[physics]
    funcs = funcs or {}
    assert set(funcs) <= set(names), set(funcs) - set(names)
    for k, name in enumerate(names):
        a.funcs[k] = _func(funcs.get(name, defaults[k]), keep)
    for k, p in enumerate(params or []):
        a.params[k] = float(p)
    a.compute_jacobian = int(compute_jacobian)
    return a


def assemble_block(m, physics, qdeg, u, *, funcs=None, params=None, fixed=None, transient=None, compute_jacobian=True,
                   rowptr=None, colind=None, want_local=False):
    keep = []
    a = _block_args(m, physics, qdeg, u, funcs, params, fixed, transient, compute_jacobian, keep)
    if rowptr is None:
        rowptr, colind = build_graph(a.nrows, m["lids"])
    out = dict(rowptr=rowptr, colind=colind, crs_vals=np.zeros(rowptr[-1]), res=np.zeros(a.nrows))
    a.rowptr, a.colind, a.crs_vals, a.res = _i(rowptr), _i(colind), _d(out["crs_vals"]), _d(out["res"])
    if want_local:
        E, n = m["lids"].shape
        out["local_J"], out["local_res"] = np.zeros((E, n, n)), np.zeros((E, n))
        a.local_J, a.local_res = _d(out["local_J"]), _d(out["local_res"])
    rc = lib().orc_assemble_block(C.byref(a))
    assert rc == 0, rc
    return out


def get_mass(m, qdeg, masswts=None):
    """Dense element mass matrices [E][n_tot][n_tot] (getWeightedMass)."""
    a = BlockArgs()
    a.dim, a.qdeg, a.nvars = m["dim"], qdeg, len(m["types"])
    for v, (t, o) in enumerate(zip(m["types"], m["orders"])):
        a.types[v], a.orders[v] = int(t), int(o)
    nodes = np.ascontiguousarray(m["nodes"], dtype=np.float64)
    offsets = np.ascontiguousarray(m["offsets"], dtype=np.int32)
    orient = np.ascontiguousarray(m["orient"], dtype=np.int8)
    a.nelem, a.nodes, a.offsets, a.orient = nodes.shape[0], _d(nodes), _i(offsets), orient.ctypes.data_as(_sp)
    E, n = m["lids"].shape
    mass = np.zeros((E, n, n))
    w = None if masswts is None else np.ascontiguousarray(masswts, dtype=np.float64)
    rc = lib().orc_get_mass(C.byref(a), _d(w), _d(mass))
    assert rc == 0, rc
    return mass


def _geom_args(m, qdeg, keep):
    a = BlockArgs()
    a.dim, a.qdeg, a.nvars = m["dim"], qdeg, len(m["types"])
    for v, (t, o) in enumerate(zip(m["types"], m["orders"])):
        a.types[v], a.orders[v] = int(t), int(o)
    nodes = np.ascontiguousarray(m["nodes"], dtype=np.float64)
    lids = np.ascontiguousarray(m["lids"], dtype=np.int32)
    offsets = np.ascontiguousarray(m["offsets"], dtype=np.int32)
    orient = np.ascontiguousarray(m["orient"], dtype=np.int8)
    keep += [nodes, lids, offsets, orient]
    a.nelem, a.nodes, a.lids, a.offsets = nodes.shape[0], _d(nodes), _i(lids), _i(offsets)
    a.orient = orient.ctypes.data_as(_sp)
    return a


def apply_mass_matrix_free(m, qdeg, x, y, masswts=None):
    """applyMassMatrixFree, !storeMass branch: y += M x with the basis recomputed per element."""
    keep = []
    a = _geom_args(m, qdeg, keep)
    x = np.ascontiguousarray(x, dtype=np.float64)
    w = None if masswts is None else np.ascontiguousarray(masswts, dtype=np.float64)
    assert lib().orc_apply_mass_matrix_free(C.byref(a), _d(w), _d(x), _d(y)) == 0


def apply_mass_stored(m, mass, x, y, index=None):
    lids = np.ascontiguousarray(m["lids"], dtype=np.int32)
    offsets = np.ascontiguousarray(m["offsets"], dtype=np.int32)
    varptr = np.ascontiguousarray(m["varptr"], dtype=np.int32)
    mass, x = np.ascontiguousarray(mass, dtype=np.float64), np.ascontiguousarray(x, dtype=np.float64)
    idx = None if index is None else np.ascontiguousarray(index, dtype=np.int32)
    E, n = lids.shape
    assert lib().orc_apply_mass_stored(E, n, len(varptr) - 1, _i(varptr), _i(offsets), _i(lids),
                                       None if idx is None else _i(idx), _d(mass), _d(x), _d(y)) == 0


def sparse3d(dense, tol):
    """Sparse3DView(denseview, tol) -> (values, columns, nnz_row, maxent)."""
    dense = np.ascontiguousarray(dense, dtype=np.float64)
    E, n, _ = dense.shape
    me = C.c_int()
    nnz = np.zeros((E, n), np.int32)
    assert lib().orc_sparse3d(E, n, _d(dense), C.c_double(tol), C.byref(me), _i(nnz), None, None) == 0
    vals, cols = np.zeros((E, n, max(me.value, 1))), np.zeros((E, n, max(me.value, 1)), np.int32)
    if me.value:
        assert lib().orc_sparse3d(E, n, _d(dense), C.c_double(tol), C.byref(me), _i(nnz), _d(vals), _i(cols)) == 0
    return vals[:, :, :me.value], cols[:, :, :me.value], nnz, me.value


def apply_mass_sparse(m, values, columns, nnz_row, x, y, index=None):
    lids = np.ascontiguousarray(m["lids"], dtype=np.int32)
    offsets = np.ascontiguousarray(m["offsets"], dtype=np.int32)
    varptr = np.ascontiguousarray(m["varptr"], dtype=np.int32)
    values, columns = np.ascontiguousarray(values, dtype=np.float64), np.ascontiguousarray(columns, dtype=np.int32)
    nnz_row, x = np.ascontiguousarray(nnz_row, dtype=np.int32), np.ascontiguousarray(x, dtype=np.float64)
    idx = None if index is None else np.ascontiguousarray(index, dtype=np.int32)
    E, n = lids.shape
    assert lib().orc_apply_mass_sparse(E, n, len(varptr) - 1, _i(varptr), _i(offsets), _i(lids),
                                       None if idx is None else _i(idx), values.shape[2], _i(nnz_row), _d(values), _i(columns),
                                       _d(x), _d(y)) == 0


def identify_database(m, qdeg, tol=1e-10):
    """identifyVolumetricDatabase: (basis_index[E], first_users[U]) by the reference's first-match scan."""
    keep = []
    a = _geom_args(m, qdeg, keep)
    E = a.nelem
    idx, fu = np.zeros(E, np.int32), np.zeros(E, np.int32)
    nu = lib().orc_identify_database(C.byref(a), C.c_double(tol), _i(idx), _i(fu))
    assert nu > 0
    return idx, fu[:nu].copy()


def assemble_block_boundary(m, physics, qdeg, u, belem, bside, bc_type, data, *, rowptr, colind, crs_vals, res,
                            funcs=None, params=None, fixed=None, transient=None, compute_jacobian=True, aux=None,
                            farfield=None):
    keep = []
    a = _block_args(m, physics, qdeg, u, funcs, params, fixed, transient, compute_jacobian, keep)
    belem = np.ascontiguousarray(belem, dtype=np.int32)
    bside = np.ascontiguousarray(bside, dtype=np.int32)
    a.nb, a.belem, a.bside, a.bc_type = len(belem), _i(belem), _i(bside), bc_type
    a.bdata = _func(data, keep)
    if aux is not None:
        aux = np.ascontiguousarray(aux, dtype=np.float64)
        keep.append(aux)
        a.aux_ip = _d(aux)
    if farfield is not None:
        farfield = np.ascontiguousarray(farfield, dtype=np.float64)
        keep.append(farfield)
        a.farfield_ip = _d(farfield)
    a.rowptr, a.colind, a.crs_vals, a.res = _i(rowptr), _i(colind), _d(crs_vals), _d(res)
    rc = lib().orc_assemble_block_boundary(C.byref(a))
    assert rc == 0, rc


def physical_side_basis_hdiv(dim, qdeg, nodes, belem, bside, orient=None):
    _, nqs = side_sizes(dim, qdeg)
    nodes = np.ascontiguousarray(nodes, dtype=np.float64)
    belem = np.ascontiguousarray(belem, dtype=np.int32)
    bside = np.ascontiguousarray(bside, dtype=np.int32)
    out = np.zeros((len(belem), 2 * dim, nqs, dim))
    op, stride = None, 0
    if orient is not None:
        orient = np.ascontiguousarray(orient, dtype=np.int8)
        op, stride = orient.ctypes.data_as(_sp), orient.shape[1]
    rc = lib().orc_physical_side_basis_hdiv(dim, qdeg, len(belem), _d(nodes), _i(belem), _i(bside), op, stride, 0,
                                            _d(out))
    assert rc == 0, rc
    return out


def flux_condition(belem, lids, off, flux, wts, basis, res, fixed=None):
    """PhysicsInterface::fluxConditions + scatterRes for one variable; basis[nb][card][nqs][ncomp]; accumulates into res."""
    belem = np.ascontiguousarray(belem, dtype=np.int32)
    lids = np.ascontiguousarray(lids, dtype=np.int32)
    off = np.ascontiguousarray(off, dtype=np.int32)
    flux, wts = np.ascontiguousarray(flux, dtype=np.float64), np.ascontiguousarray(wts, dtype=np.float64)
    basis = np.ascontiguousarray(basis, dtype=np.float64)
    nb, card, nqs, ncomp = basis.shape
    fx = None if fixed is None else np.ascontiguousarray(fixed, dtype=np.uint8)
    rc = lib().orc_flux_condition(nb, card, nqs, ncomp, _i(belem), _i(lids), lids.shape[1], _i(off),
                                  None if fx is None else _u(fx), _d(flux), _d(wts), _d(basis), _d(res))
    assert rc == 0, rc


# ---- L2-projection systems (setInitial / setDirichlet) ----------------------------------------------------------------
def _ci(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _cd(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def project_rhs(lids, off, data, basis, wts, rhs):
    """getInitial(project) + setInitial's vector loop, one variable; data[E][nq][ncomp], basis[E][card][nq][ncomp]."""
    lids, off, data, basis, wts = _ci(lids), _ci(off), _cd(data), _cd(basis), _cd(wts)
    ne, card, nq, ncomp = basis.shape
    rc = lib().orc_project_rhs(ne, card, nq, ncomp, _i(lids), lids.shape[1], _i(off), _d(data), _d(basis), _d(wts), _d(rhs))
    assert rc == 0, rc


def set_initial_mass(lids, mass, lump, rowptr, colind, vals):
    """setInitial's matrix loop over dense element mass matrices + fix_zero_rows."""
    lids, mass, rowptr, colind = _ci(lids), _cd(mass), _ci(rowptr), _ci(colind)
    rc = lib().orc_set_initial_mass(lids.shape[0], lids.shape[1], _i(lids), _d(mass), int(lump), len(rowptr) - 1, _i(rowptr),
                                    _i(colind), _d(vals))
    assert rc == 0, rc


def set_initial_nodal(lids, off, nodal, initial, vert_of_dof=None):
    """vert_of_dof[k] = vertex of order-1 dof k; default: the single-variable mesh of this oracle, whose LID list starts
    with the vertex sites in vertex order, so that dof k (LID position off[k]) sits on vertex off[k]."""
    lids, off, nodal = _ci(lids), _ci(off), _cd(nodal)
    vod = _ci(off[:nodal.shape[1]] if vert_of_dof is None else vert_of_dof)
    rc = lib().orc_set_initial_nodal(lids.shape[0], nodal.shape[1], _i(lids), lids.shape[1], _i(off), _i(vod), _d(nodal),
                                     _d(initial))
    assert rc == 0, rc


def dirichlet_boundary(n_tot, off, dip, basis, wts, normals=None):
    """getDirichletBoundary + getMassBoundary of one variable: (dvals[nb][n_tot], mass[nb][n_tot][n_tot])."""
    off, dip, basis, wts = _ci(off), _cd(dip), _cd(basis), _cd(wts)
    nb, card, nqs, ncomp = basis.shape
    dvals, mass = np.zeros((nb, n_tot)), np.zeros((nb, n_tot, n_tot))
    nr = None if normals is None else _cd(normals)
    rc = lib().orc_dirichlet_boundary(nb, card, nqs, ncomp, n_tot, _i(off), 0 if nr is None else 1, _d(dip), _d(basis),
                                      _d(wts), None if nr is None else _d(nr), _d(dvals), _d(mass))
    assert rc == 0, rc
    return dvals, mass


def set_dirichlet_group(belem, lids, fixed, dvals, mass, lump, rowptr, colind, vals, rhs):
    belem, lids, rowptr, colind = _ci(belem), _ci(lids), _ci(rowptr), _ci(colind)
    fx = None if fixed is None else np.ascontiguousarray(fixed, dtype=np.uint8)
    rc = lib().orc_set_dirichlet_group(len(belem), lids.shape[1], _i(belem), _i(lids), None if fx is None else _u(fx),
                                       _d(_cd(dvals)), _d(_cd(mass)), int(lump), _i(rowptr), _i(colind), _d(vals), _d(rhs))
    assert rc == 0, rc


def set_dirichlet_identity(lids, fixed, rowptr, colind, vals):
    lids, rowptr, colind = _ci(lids), _ci(rowptr), _ci(colind)
    fx = None if fixed is None else np.ascontiguousarray(fixed, dtype=np.uint8)
    rc = lib().orc_set_dirichlet_identity(lids.shape[0], lids.shape[1], _i(lids), None if fx is None else _u(fx), _i(rowptr),
                                          _i(colind), _d(vals))
    assert rc == 0, rc


# ---- shallowwaterHybridized, point level ---------------------------------------------------------------------------
PHYS_SHALLOWWATER_HYBRIDIZED = 4
PHYS_FUNCS[PHYS_SHALLOWWATER_HYBRIDIZED] = ["source H", "source Hux", "source Huy"]
PHYS_DEFAULTS[PHYS_SHALLOWWATER_HYBRIDIZED] = [0.0, 0.0, 0.0]


def swh_eigendecomp(dim, Shat, nrm, g=9.81):
    nv = dim + 1
    L, lam, R = np.zeros((nv, nv)), np.zeros(nv), np.zeros((nv, nv))
    Shat, nrm = np.ascontiguousarray(Shat, dtype=np.float64), np.ascontiguousarray(nrm, dtype=np.float64)
    f = lib().orc_swh_eigendecomp
    f.restype = None
    f.argtypes = [C.c_int, _dp, _dp, C.c_double, _dp, _dp, _dp]
    f(dim, _d(Shat), _d(nrm), g, _d(L), _d(lam), _d(R))
    return L, lam, R


def swh_flux_vector(dim, S, g=9.81):
    S = np.ascontiguousarray(S, dtype=np.float64)
    F = np.zeros((dim + 1, dim))
    f = lib().orc_swh_flux_vector
    f.restype = None
    f.argtypes = [C.c_int, _dp, C.c_double, _dp]
    f(dim, _d(S), g, _d(F))
    return F


def swh_stab_term(dim, S, Shat, nrm, g=9.81, roe=True):
    S, Shat, nrm = (np.ascontiguousarray(a, dtype=np.float64) for a in (S, Shat, nrm))
    out = np.zeros(dim + 1)
    f = lib().orc_swh_stab_term
    f.restype = None
    f.argtypes = [C.c_int, _dp, _dp, _dp, C.c_double, C.c_int, _dp]
    f(dim, _d(S), _d(Shat), _d(nrm), g, int(roe), _d(out))
    return out


def swh_boundary_term(dim, btype, S, Shat, Sinf, nrm, g=9.81):
    S, Shat, Sinf, nrm = (np.ascontiguousarray(a, dtype=np.float64) for a in (S, Shat, Sinf, nrm))
    out = np.zeros(dim + 1)
    f = lib().orc_swh_boundary_term
    f.restype = None
    f.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_double, _dp]
    f(dim, btype, _d(S), _d(Shat), _d(Sinf), _d(nrm), g, _d(out))
    return out


def swh_interface_flux(dim, side_type, roe, S, Shat, Sinf, nrm, g=9.81):
    S, Shat, Sinf, nrm = (np.ascontiguousarray(a, dtype=np.float64) for a in (S, Shat, Sinf, nrm))
    out = np.zeros(dim + 1)
    f = lib().orc_swh_interface_flux
    f.restype = None
    f.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_double, _dp]
    f(dim, side_type, int(roe), _d(S), _d(Shat), _d(Sinf), _d(nrm), g, _d(out))
    return out


def swh_matvec(A, x):
    A, x = np.ascontiguousarray(A, dtype=np.float64), np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros(len(x))
    f = lib().orc_swh_matvec
    f.restype = None
    f.argtypes = [C.c_int, _dp, _dp, _dp]
    f(len(x), _d(A), _d(x), _d(y))
    return y


def swh_hdg_element(m, qdeg, u, lam, side_types, farfield, g=9.81, roe=True, transient=None):
    """HDG element, side part: -> (res [E][36], blocks [E][36][36]) (orc_swh_hdg_element)."""
    keep = []
    a = _block_args(m, PHYS_SHALLOWWATER_HYBRIDIZED, qdeg, u, None, [g, 1.0 if roe else 0.0], None, transient, True, keep)
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    st = np.ascontiguousarray(side_types, dtype=np.uint8)
    ff = np.ascontiguousarray(farfield, dtype=np.float64)
    E = m["nelem"]
    res, blocks = np.zeros((E, 36)), np.zeros((E, 36, 36))
    rc = lib().orc_swh_hdg_element(C.byref(a), _d(lam), _u(st), _d(ff), _d(blocks), _d(res))
    assert rc == 0, rc
    return res, blocks


def subgrid_nonlinear_solver(m, qdeg, u, lam, side_types, farfield, max_iter, tol, g=9.81, roe=True, transient=None):
    """SubGridDtN_Solver::nonlinearSolver (src/subgrid/subgridDtN_solver.cpp:909-1041) restated for subgrids of one HDG
    element each (element-local interior unknowns), one independent loop per element:
        while iter < sub_maxNLiter and resnorm_scaled > sub_NLtol:
            assembleJacobianResidual (side blocks + volume block); iter 0: resnorm_initial = |res|_inf, scaled = 1 (0 if 0);
            else scaled = |res|_inf / resnorm_initial; if scaled > tol: solve J du = res, sol += du; iter += 1
    -> (u, iters[E], scaled[E]).  The element blocks come from the oracle's C restatements; the solve is numpy's."""
    u = np.array(u, dtype=np.float64)
    E = m["nelem"]
    off = m["offsets"]
    rows = m["lids"][:, off]                              # [E][12] flattened (variable, dof) -> global row
    iters, scaled, rn0 = np.zeros(E, np.int32), np.full(E, 10.0 * tol), np.zeros(E)
    inloop = np.ones(E, bool)
    for it in range(max_iter):
        if not inloop.any():
            break
        res, blk = swh_hdg_element(m, qdeg, u, lam, side_types, farfield, g=g, roe=roe, transient=transient)
        vol = assemble_block(m, PHYS_SHALLOWWATER_HYBRIDIZED, qdeg, u, params=[g], transient=transient, want_local=True)
        A = blk[:, :12, :12] + vol["local_J"][:, off][:, :, off]
        r = res[:, :12] + vol["local_res"][:, off]
        nrm = np.abs(r).max(axis=1)
        for e in np.flatnonzero(inloop):
            if it == 0:
                rn0[e] = nrm[e]
                scaled[e] = 1.0 if nrm[e] > 0.0 else 0.0
            else:
                scaled[e] = nrm[e] / rn0[e]
            if scaled[e] > tol:
                u[rows[e]] += np.linalg.solve(A[e], r[e])
            iters[e] += 1
            inloop[e] = scaled[e] > tol
    return u, iters, scaled


# ---------------------------------------------------------------------------------------------------------------------
# FunctionManager<AD>::evaluate over (elem, pt) views with Sacado-style derivative arrays (reference:
# src/managers/functionManager.cpp:95-540 decomposeFunctions, :543-860 evaluate op by op; src/tools/vista.hpp:22-132),
# restated with numpy: deck strings that read solution fields ("1+e*e") and other named functions of the deck.
# Test infrastructure (the checker), small cases only.
# ---------------------------------------------------------------------------------------------------------------------

class ADView:
    """A View<AD**>(elem, pt): val [E][q], dx [E][q][W] (W = dofs per element, the seeded slots)."""

    def __init__(self, val, dx=None, W=0):
        self.val = np.asarray(val, dtype=np.float64)
        self.dx = np.zeros(self.val.shape + (W,)) if dx is None else dx

    @staticmethod
    def _lift(o, like):
        return o if isinstance(o, ADView) else ADView(np.broadcast_to(np.float64(o), like.val.shape).copy(), W=like.dx.shape[-1])

    def __add__(s, o):
        o = ADView._lift(o, s)
        return ADView(s.val + o.val, s.dx + o.dx)
    __radd__ = __add__

    def __sub__(s, o):
        o = ADView._lift(o, s)
        return ADView(s.val - o.val, s.dx - o.dx)

    def __rsub__(s, o):
        return ADView._lift(o, s) - s

    def __neg__(s):
        return ADView(-s.val, -s.dx)

    def __mul__(s, o):
        o = ADView._lift(o, s)
        return ADView(s.val * o.val, s.dx * o.val[..., None] + o.dx * s.val[..., None])
    __rmul__ = __mul__

    def __truediv__(s, o):
        o = ADView._lift(o, s)
        q = s.val / o.val
        return ADView(q, (s.dx - o.dx * q[..., None]) / o.val[..., None])

    def __rtruediv__(s, o):
        return ADView._lift(o, s) / s

    def __pow__(s, o):
        o = ADView._lift(o, s)
        p = s.val ** o.val
        dx = (o.val * s.val ** (o.val - 1.0))[..., None] * s.dx
        if np.any(o.dx != 0.0):
            dx = dx + (p * np.log(s.val))[..., None] * o.dx
        return ADView(p, dx)

    def __rpow__(s, o):
        return ADView._lift(o, s) ** s

    def _cmp(s, o, op):
        o = ADView._lift(o, s)
        return ADView(op(s.val, o.val).astype(np.float64), W=s.dx.shape[-1])

    def __lt__(s, o): return s._cmp(o, np.less)
    def __gt__(s, o): return s._cmp(o, np.greater)
    def __le__(s, o): return s._cmp(o, np.less_equal)
    def __ge__(s, o): return s._cmp(o, np.greater_equal)


def _ad_unary(f, df):
    def g(a):
        if not isinstance(a, ADView):
            return f(a)
        return ADView(f(a.val), df(a.val)[..., None] * a.dx)
    return g


_AD_FUNCS = dict(sin=_ad_unary(np.sin, np.cos), cos=_ad_unary(np.cos, lambda v: -np.sin(v)),
                 tan=_ad_unary(np.tan, lambda v: 1.0 + np.tan(v) ** 2), exp=_ad_unary(np.exp, np.exp),
                 log=_ad_unary(np.log, lambda v: 1.0 / v), sqrt=_ad_unary(np.sqrt, lambda v: 0.5 / np.sqrt(v)),
                 sinh=_ad_unary(np.sinh, np.cosh), cosh=_ad_unary(np.cosh, np.sinh),
                 abs=_ad_unary(np.abs, np.sign))


def deck_eval_ad(text, fields, functions=None, _open=()):
    """Evaluate a deck string op by op on AD views.  fields: name ('e', 'grad(e)[x]', 'x', 't', ...) -> ADView or
    array; functions: the deck's other named strings (resolved into sub-trees as decomposeFunctions does)."""
    import re
    functions = functions or {}
    env = dict(_AD_FUNCS)
    env["pi"] = np.pi
    expr = text.replace("^", "**")
    names = sorted(set(list(fields) + list(functions)), key=len, reverse=True)
    for k, nm in enumerate(names):  # longest first: 'grad(e)[x]' before 'e'
        tag = "_v%d_" % k
        pat = re.escape(nm) if not nm.isidentifier() else r"(?<![A-Za-z0-9_])" + re.escape(nm) + r"(?![A-Za-z0-9_(\[])"
        if not re.search(pat, expr):
            continue
        if nm in fields:
            env[tag] = fields[nm]
        else:
            assert nm not in _open, "function '%s' refers to itself" % nm
            env[tag] = deck_eval_ad(functions[nm], fields, functions, _open + (nm,))
        expr = re.sub(pat, tag, expr)
    return eval(expr, {"__builtins__": {}}, env)


def assemble_thermal_fields(dim, order, qdeg, nodes, lids, offsets, u, funcs, *, fixed=None, rowptr=None, colind=None,
                            functions=None):
    """thermal::volumeResidual (src/physics/thermal.cpp:71-165, steady) with its functions given as deck strings that may
    read the solution fields, AD arrays of width n seeded at off(j) (workset.cpp:823-859), then the scatter of
    assemblyManager.cpp:4031-4145 (-res.val(), +res.dx(col), fixed rows skipped).  funcs: 'thermal source',
    'thermal diffusion' -> string or number."""
    lids = np.ascontiguousarray(lids, dtype=np.int32)
    E, n = lids.shape
    pb = physical_basis(dim, order, qdeg, nodes)
    B, G, w, ip = pb["basis"], pb["basis_grad"], pb["wts"], pb["ip"]      # [E][n][q], [E][n][q][d], [E][q], [E][q][d]
    off = np.asarray(offsets)
    ue = u[lids[:, off]]                                                    # value of basis function j: LID position off[j]
    nq = w.shape[1]
    fields = {}
    val = np.einsum("ej,ejq->eq", ue, B)
    dx = np.zeros((E, nq, n))
    dx[:, :, off] = np.transpose(B, (0, 2, 1))                              # d e / d u_(slot off[j]) = N_j
    fields["e"] = ADView(val, dx)
    for d, c in enumerate("xyz"[:dim]):
        gd = np.zeros((E, nq, n))
        gd[:, :, off] = np.transpose(G[..., d], (0, 2, 1))
        fields["grad(e)[%s]" % c] = ADView(np.einsum("ej,ejq->eq", ue, G[..., d]), gd)
        fields[c] = ADView(ip[..., d], W=n)
    fields["t"] = ADView(np.zeros((E, nq)), W=n)

    def fn(name, default):
        v = funcs.get(name, default)
        r = deck_eval_ad(v, fields, functions) if isinstance(v, str) else v
        return ADView._lift(r, fields["e"])
    src, kap = fn("thermal source", 0.0), fn("thermal diffusion", 1.0)
    if rowptr is None:
        rowptr, colind = build_graph(len(u), lids)
    vals, res = np.zeros(rowptr[-1]), np.zeros(len(u))
    for j in range(n):  # residual row of basis function j: res(e, off[j])
        r = ADView(np.zeros((E, nq)), W=n)
        r = r - src * B[:, j, :] * w
        for d, c in enumerate("xyz"[:dim]):
            r = r + kap * fields["grad(e)[%s]" % c] * (G[:, j, :, d] * w)
        rv, rdx = r.val.sum(axis=1), r.dx.sum(axis=1)                       # [E], [E][n] (slot order)
        rows = lids[:, off[j]]
        for e in range(E):
            row = rows[e]
            if fixed is not None and fixed[row]:
                continue
            res[row] -= rv[e]
            lo, hi = rowptr[row], rowptr[row + 1]
            vals[lo + np.searchsorted(colind[lo:hi], lids[e])] += rdx[e]
    return dict(rowptr=rowptr, colind=colind, crs_vals=vals, res=res)
