"""ctypes binding of include/mrhyde_amd.h (one Python method per C entry point)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "lib", "libmrhyde_amd.so")

PATH_AUTO, PATH_ELEMENT_ATOMIC, PATH_ROW_OWNER, PATH_LOCAL_THEN_SCATTER = 0, 1, 2, 3
FUNC_CONSTANT, FUNC_IP_ARRAY, FUNC_SINPROD = 0, 1, 2
TOPO_QUAD4, TOPO_HEX8 = 4, 8
PHYSICS_THERMAL = 1
MAX_VARS = 8

# every symbol include/mrhyde_amd.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "mha_last_error", "mha_version", "mha_device_count", "mha_block_create", "mha_block_destroy", "mha_set_stream",
    "mha_set_mesh", "mha_set_graph", "mha_get_graph_sizes", "mha_get_graph", "mha_physics_select",
    "mha_set_function", "mha_set_time_integration", "mha_assemble_jacres", "mha_compute_local_jacres",
    "mha_scatter_local", "mha_apply_dbc_diag", "mha_gather", "mha_num_worksets", "mha_workset_update",
    "mha_workset_view", "mha_mesh_sizes", "mha_mesh_structured", "mha_mesh_multi_sizes", "mha_mesh_structured_multi", "mha_swhdg_condensed_element", "mha_compute_flux", "mha_newton_create", "mha_newton_destroy", "mha_newton_reset", "mha_newton_residual", "mha_newton_norm", "mha_newton_decide", "mha_newton_jacobian", "mha_newton_update", "mha_newton_step", "mha_newton_state", "mha_dirichlet_lift", "mha_export_plan_create", "mha_export_plan_destroy", "mha_export_pack", "mha_export_unpack_add", "mha_export_buffers", "mha_export_bytes_on_wire", "mha_comm_unique_id", "mha_comm_create", "mha_comm_destroy", "mha_export_add", "mha_get_info", "mha_set_timing",
    "mha_get_last_kernel_ms", "mha_row_partition_build", "mha_row_partition_sizes", "mha_row_partition_get",
    "mha_row_partition_destroy", "mha_scatter_plan_create", "mha_scatter_plan_nnz",
    "mha_scatter_plan_graph", "mha_scatter_plan_apply", "mha_scatter_plan_destroy", "mha_add_boundary_group", "mha_clear_boundary_groups", "mha_num_boundary_groups",
    "mha_assemble_boundary", "mha_boundary_update", "mha_boundary_view", "mha_set_physics_parameter",
    "mha_set_orientation", "mha_swhdg_side_terms", "mha_swhdg_eigendecomp", "mha_get_mass", "mha_swhdg_element_blocks", "mha_batched_condense", "mha_set_function_expression", "mha_set_time", "mha_check_expression",
    "mha_add_flux_group", "mha_add_dirichlet_group", "mha_set_initial", "mha_set_initial_nodal", "mha_set_dirichlet",
    "mha_workset_compute_solution", "mha_workset_compute_residual",
    "mha_sparse3d_create", "mha_sparse3d_views", "mha_sparse3d_size", "mha_sparse3d_destroy", "mha_database_build",
    "mha_database_get", "mha_apply_mass_matrix_free", "mha_swhdg_subgrid_workspace_bytes", "mha_swhdg_subgrid_solve",
]
MASS_ON_THE_FLY, MASS_LOCAL, MASS_DATABASE, MASS_DATABASE_SPARSE = 0, 1, 2, 3
SWH_INTERFACE, SWH_FARFIELD, SWH_SLIP = 0, 1, 2
BASIS_HGRAD, BASIS_HVOL, BASIS_HDIV = 0, 1, 2
PHYSICS_IDS = {"thermal": 1, "porousMixed": 2, "navierstokes": 3, "shallowwaterHybridized": 4}
PATH_POINT_ENGINE = 4
PATH_ROW_GATHER = 5
BC_NEUMANN, BC_WEAK_DIRICHLET, BC_FLUX = 1, 2, 3
BC_INTERFACE = 5  # thermal: weak-Dirichlet terms with the trace "aux e <side>" as data
BC_SWH_INTERFACE, BC_SWH_FARFIELD, BC_SWH_SLIP = 10, 11, 12


class MhaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mrhyde_amd error %d: %s" % (code, msg))
        self.code = code


class BlockDesc(C.Structure):
    _fields_ = [("dimension", C.c_int), ("topology", C.c_int), ("num_vars", C.c_int),
                ("basis_type", C.c_int * MAX_VARS), ("basis_order", C.c_int * MAX_VARS),
                ("quadrature_degree", C.c_int), ("workset_size", C.c_int), ("device", C.c_int)]


_lib = None


def lib_path():
    return _LIB


def load_library():
    """Load the HIP shared library; fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise ImportError("%s is missing: run `make -C mrhyde_amd/csrc` (or __graft_entry__.build()); "
                              "there is no fallback path" % _LIB)
        # PyTorch bundles its own libamdhip64 (same soname as /opt/rocm's): load torch first so the process
        # holds exactly one HIP runtime, the one that owns torch's device memory and streams.
        import torch  # noqa: F401
        _lib = C.CDLL(_LIB)
        _lib.mha_last_error.restype = C.c_char_p
        _lib.mha_version.restype = C.c_char_p
        _lib.mha_block_destroy.restype = None
        _lib.mha_set_function.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        _lib.mha_set_time_integration.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                                  C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mha_assemble_jacres.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5
        _lib.mha_compute_local_jacres.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        _lib.mha_scatter_local.argtypes = [C.c_void_p] * 5
        _lib.mha_apply_dbc_diag.argtypes = [C.c_void_p, C.c_void_p]
        _lib.mha_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mha_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        _lib.mha_set_mesh.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib.mha_set_graph.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mha_get_graph.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mha_get_graph_sizes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mha_physics_select.argtypes = [C.c_void_p, C.c_int]
        _lib.mha_num_worksets.argtypes = [C.c_void_p]
        _lib.mha_workset_update.argtypes = [C.c_void_p, C.c_int]
        _lib.mha_workset_view.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mha_get_info.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        _lib.mha_set_timing.argtypes = [C.c_void_p, C.c_int]
        _lib.mha_get_last_kernel_ms.argtypes = [C.c_void_p, C.c_void_p]
        _lib.mha_block_destroy.argtypes = [C.c_void_p]
        _lib.mha_add_boundary_group.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_void_p]
        _lib.mha_add_flux_group.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mha_add_dirichlet_group.argtypes = _lib.mha_add_flux_group.argtypes
        _lib.mha_set_initial.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _lib.mha_set_initial_nodal.argtypes = [C.c_void_p, C.c_void_p]
        _lib.mha_set_dirichlet.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _lib.mha_workset_compute_solution.argtypes = [C.c_void_p] * 4
        _lib.mha_workset_compute_residual.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 3
        _lib.mha_clear_boundary_groups.argtypes = [C.c_void_p]
        _lib.mha_sparse3d_create.argtypes = [C.c_int64, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
        _lib.mha_sparse3d_views.argtypes = [C.c_void_p] * 5
        _lib.mha_sparse3d_size.argtypes = [C.c_void_p, C.c_void_p]
        _lib.mha_sparse3d_destroy.argtypes = [C.c_void_p]
        _lib.mha_sparse3d_destroy.restype = None
        _lib.mha_database_build.argtypes = [C.c_void_p, C.c_void_p]
        _lib.mha_swhdg_subgrid_workspace_bytes.argtypes = [C.c_void_p, C.c_void_p]
        _lib.mha_swhdg_subgrid_solve.argtypes = ([C.c_void_p] * 7 + [C.c_int, C.c_double, C.c_void_p, C.c_int64] + [C.c_void_p] * 5)
        _lib.mha_database_get.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mha_apply_mass_matrix_free.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        _lib.mha_num_boundary_groups.argtypes = [C.c_void_p]
        _lib.mha_assemble_boundary.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        _lib.mha_boundary_update.argtypes = [C.c_void_p, C.c_int]
        _lib.mha_boundary_view.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mha_set_physics_parameter.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        _lib.mha_set_orientation.argtypes = [C.c_void_p, C.c_void_p]
        _lib.mha_get_mass.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mha_swhdg_element_blocks.argtypes = [C.c_void_p] * 9
        _lib.mha_set_function_expression.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        _lib.mha_set_time.argtypes = [C.c_void_p, C.c_double]
        _lib.mha_batched_condense.argtypes = [C.c_int, C.c_int, C.c_int64] + [C.c_void_p] * 7
        _lib.mha_swhdg_side_terms.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int64] + [C.c_void_p] * 10
        _lib.mha_swhdg_eigendecomp.argtypes = [C.c_double, C.c_int64] + [C.c_void_p] * 6
    return _lib


def _check(rc):
    if rc != 0:
        raise MhaError(rc, load_library().mha_last_error().decode())


def version():
    return load_library().mha_version().decode()


def device_count():
    return load_library().mha_device_count()


def _np(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device arguments must be contiguous CUDA tensors"
    return C.c_void_p(t.data_ptr())


def swhdg_side_terms(side_type, roe, g, S, Shat, normals, Sinf=None, want=("fluxvec", "term", "iflux", "d_dS", "d_dShat")):
    """shallowwaterHybridized side terms at the points of CUDA tensors S, Shat [npts,3], normals [npts,2] (mha_swhdg_side_terms).
    -> dict of CUDA tensors."""
    import torch
    npts = S.shape[0]
    shapes = dict(fluxvec=(npts, 3, 2), term=(npts, 3), iflux=(npts, 3), d_dS=(npts, 3, 3), d_dShat=(npts, 3, 3))
    out = {k: torch.zeros(shapes[k], dtype=torch.float64, device=S.device) for k in want}
    _check(load_library().mha_swhdg_side_terms(side_type, int(roe), float(g), npts, _ptr(S), _ptr(Shat), _ptr(normals),
                                               _ptr(Sinf), _ptr(out.get("fluxvec")), _ptr(out.get("term")),
                                               _ptr(out.get("iflux")), _ptr(out.get("d_dS")), _ptr(out.get("d_dShat")),
                                               None))
    return out


def batched_condense(n_int, n_trace, blocks, res, want_du=True):
    """Static condensation of element blocks [E][n][n] / [E][n] (CUDA tensors) -> (schur, gvec, du, num_singular)."""
    import torch
    E = blocks.shape[0]
    schur = torch.zeros((E, n_trace, n_trace), dtype=torch.float64, device=blocks.device)
    gvec = torch.zeros((E, n_trace), dtype=torch.float64, device=blocks.device)
    du = torch.zeros((E, n_int), dtype=torch.float64, device=blocks.device) if want_du else None
    ns = C.c_int()
    _check(load_library().mha_batched_condense(n_int, n_trace, E, _ptr(blocks), _ptr(res), _ptr(schur), _ptr(gvec), _ptr(du),
                                               C.byref(ns), None))
    return schur, gvec, du, ns.value


NEWTON_SOLVE, NEWTON_BACKTRACKED, NEWTON_DONE = 1, 2, 3


class Newton:
    """The nonlinear-solve protocol of SolverManager::nonlinearSolver behind the C ABI (mha_newton_*): the caller supplies
    the linear solve between step() and update()."""

    def __init__(self, blk, max_iter=10, nl_tol=1e-6, nl_abs_tol=1e-6, use_relative=True, use_absolute=False,
                 allow_backtracking=False, autotune=True):
        lib = load_library()
        self._blk, self._h = blk, C.c_void_p()
        lib.mha_newton_create.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _check(lib.mha_newton_create(blk._h, int(max_iter), float(nl_tol), float(nl_abs_tol), int(use_relative), int(use_absolute),
                                     int(allow_backtracking), int(autotune), C.byref(self._h)))

    def step(self, u, res, vals, u_prev=None, u_stage=None):
        lib = load_library()
        a = C.c_int()
        lib.mha_newton_step.argtypes = [C.c_void_p] * 7
        _check(lib.mha_newton_step(self._h, _ptr(u), _ptr(u_prev), _ptr(u_stage), _ptr(res), _ptr(vals), C.byref(a)))
        return a.value

    def residual(self, u, res, u_prev=None, u_stage=None):
        lib = load_library()
        lib.mha_newton_residual.argtypes = [C.c_void_p] * 5
        _check(lib.mha_newton_residual(self._h, _ptr(u), _ptr(u_prev), _ptr(u_stage), _ptr(res)))

    def norm(self, res):
        lib = load_library()
        v = C.c_double()
        lib.mha_newton_norm.argtypes = [C.c_void_p] * 3
        _check(lib.mha_newton_norm(self._h, _ptr(res), C.byref(v)))
        return v.value

    def decide(self, resnorm, u):
        lib = load_library()
        a = C.c_int()
        lib.mha_newton_decide.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
        _check(lib.mha_newton_decide(self._h, float(resnorm), _ptr(u), C.byref(a)))
        return a.value

    def jacobian(self, u, res, vals, u_prev=None, u_stage=None):
        lib = load_library()
        lib.mha_newton_jacobian.argtypes = [C.c_void_p] * 6
        _check(lib.mha_newton_jacobian(self._h, _ptr(u), _ptr(u_prev), _ptr(u_stage), _ptr(res), _ptr(vals)))

    def update(self, u, du):
        lib = load_library()
        lib.mha_newton_update.argtypes = [C.c_void_p] * 3
        _check(lib.mha_newton_update(self._h, _ptr(u), _ptr(du)))

    def state(self):
        lib = load_library()
        it, st = C.c_int(), C.c_int()
        rn, rs, rf, al = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        lib.mha_newton_state.argtypes = [C.c_void_p] * 7
        _check(lib.mha_newton_state(self._h, C.byref(it), C.byref(rn), C.byref(rs), C.byref(rf), C.byref(al), C.byref(st)))
        return dict(iteration=it.value, resnorm=rn.value, resnorm_scaled=rs.value, resnorm_first=rf.value, alpha=al.value,
                    status=st.value)

    def close(self):
        if getattr(self, "_h", None):
            lib = load_library()
            lib.mha_newton_destroy.argtypes = [C.c_void_p]
            lib.mha_newton_destroy.restype = None
            lib.mha_newton_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ScatterPlan:
    """Scatter of dense element blocks through an arbitrary LID map (mha_scatter_plan_*): the flux -> trace scatter of
    the HDG caller.  lids [E][n] int32 (host); graph given or built (every dof of an element couples)."""

    def __init__(self, lids, nrows, rowptr=None, colind=None, fixed=None):
        lib = load_library()
        lids = _np(lids, np.int32)
        self.n, self.nrows = lids.shape[1], int(nrows)
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        rp = None if rowptr is None else _np(rowptr, np.int32)
        ci = None if colind is None else _np(colind, np.int32)
        fx = None if fixed is None else _np(fixed, np.uint8)
        self._h = C.c_void_p()
        lib.mha_scatter_plan_create.argtypes = [C.c_int, C.c_int64, C.c_int64] + [C.c_void_p] * 5
        _check(lib.mha_scatter_plan_create(self.n, lids.shape[0], self.nrows, vp(lids), vp(rp), vp(ci), vp(fx),
                                           C.byref(self._h)))
        nnz = C.c_int64()
        lib.mha_scatter_plan_nnz.argtypes = [C.c_void_p, C.c_void_p]
        _check(lib.mha_scatter_plan_nnz(self._h, C.byref(nnz)))
        self.nnz = nnz.value

    def graph(self):
        lib = load_library()
        rowptr, colind = np.zeros(self.nrows + 1, np.int32), np.zeros(self.nnz, np.int32)
        lib.mha_scatter_plan_graph.argtypes = [C.c_void_p] * 3
        _check(lib.mha_scatter_plan_graph(self._h, rowptr.ctypes.data_as(C.c_void_p), colind.ctypes.data_as(C.c_void_p)))
        return rowptr, colind

    def apply(self, blocks=None, vec=None, res=None, vals=None, overwrite=False, stream=None):
        lib = load_library()
        lib.mha_scatter_plan_apply.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p]
        _check(lib.mha_scatter_plan_apply(self._h, _ptr(blocks), _ptr(vec), _ptr(res), _ptr(vals), int(overwrite),
                                          None if stream is None else C.c_void_p(stream)))

    def close(self):
        if getattr(self, "_h", None):
            lib = load_library()
            lib.mha_scatter_plan_destroy.argtypes = [C.c_void_p]
            lib.mha_scatter_plan_destroy.restype = None
            lib.mha_scatter_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Sparse3D:
    """Sparse3DView (src/tools/sparse3DView.hpp) of a CUDA tensor dense[E][n][n]; device-resident."""

    def __init__(self, dense, tol):
        E, n, n2 = dense.shape
        assert n == n2 and dense.is_contiguous()
        self._h = C.c_void_p()
        self.num_elems, self.n = E, n
        _check(load_library().mha_sparse3d_create(E, n, _ptr(dense), float(tol), None, C.byref(self._h)))
        me, v, c, z = C.c_int(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(load_library().mha_sparse3d_views(self._h, C.byref(me), C.byref(v), C.byref(c), C.byref(z)))
        self.maxent, self._values, self._columns, self._nnz = me.value, v.value, c.value, z.value

    def size(self):
        t = C.c_int64()
        _check(load_library().mha_sparse3d_size(self._h, C.byref(t)))
        return t.value

    def numpy(self):
        """(values[E][n][maxent], columns[E][n][maxent], nnz_row[E][n]) copied to the host."""
        import torch
        torch.cuda.synchronize()
        E, n, me = self.num_elems, self.n, max(self.maxent, 1)
        vals, cols, nnz = np.zeros((E, n, me)), np.zeros((E, n, me), np.int32), np.zeros((E, n), np.int32)
        for arr, ptr in ((vals, self._values), (cols, self._columns), (nnz, self._nnz)):
            assert _hip_memcpy_dtoh(arr.ctypes.data, ptr, arr.size * arr.itemsize) == 0
        return vals[:, :, :self.maxent], cols[:, :, :self.maxent], nnz

    def close(self):
        if self._h:
            load_library().mha_sparse3d_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def check_expression(text):
    """Raises MhaError if the deck string is not in the supported grammar (host only)."""
    lib = load_library()
    lib.mha_check_expression.argtypes = [C.c_char_p]
    _check(lib.mha_check_expression(text.encode()))


def swhdg_eigendecomp(g, Shat, normals):
    import torch
    npts = Shat.shape[0]
    L = torch.zeros((npts, 3, 3), dtype=torch.float64, device=Shat.device)
    lam = torch.zeros((npts, 3), dtype=torch.float64, device=Shat.device)
    R = torch.zeros_like(L)
    _check(load_library().mha_swhdg_eigendecomp(float(g), npts, _ptr(Shat), _ptr(normals), _ptr(L), _ptr(lam), _ptr(R), None))
    return L, lam, R


def mesh_structured(dim, order, ncell, lo=None, hi=None):
    """Structured quad/hex mesh + HGRAD dof map (mha_mesh_structured).  Host numpy arrays."""
    lib = load_library()
    ncell = _np(ncell, np.int32)
    lo = _np(np.zeros(3) if lo is None else lo, np.float64)
    hi = _np(np.ones(3) if hi is None else hi, np.float64)
    nv, ne, nd = C.c_int(), C.c_int(), C.c_int64()
    _check(lib.mha_mesh_sizes(dim, order, ncell.ctypes.data_as(C.c_void_p), C.byref(nv), C.byref(ne), C.byref(nd)))
    n, nn = (order + 1) ** dim, 2 ** dim
    m = dict(verts=np.zeros((nv.value, dim)), cell2vert=np.zeros((ne.value, nn), np.int32),
             lids=np.zeros((ne.value, n), np.int32), offsets=np.zeros(n, np.int32),
             boundary=np.zeros(nd.value, np.uint8), ndof=nd.value, nelem=ne.value, dim=dim, order=order)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    _check(lib.mha_mesh_structured(dim, order, vp(ncell), vp(lo), vp(hi), vp(m["verts"]), vp(m["cell2vert"]),
                                   vp(m["lids"]), vp(m["offsets"]), vp(m["boundary"])))
    m["nodes"] = np.ascontiguousarray(m["verts"][m["cell2vert"]])
    return m


def mesh_multi(dim, ncell, types, orders, lo=None, hi=None):
    """Structured mesh + subcell-major dof map of a block of several variables (mha_mesh_structured_multi).
    -> dict(verts, cell2vert, nodes, lids, offsets, orient, side_mask, dof_var, ndof, nelem, n_tot)."""
    lib = load_library()
    ncell = np.ascontiguousarray(ncell, dtype=np.int32)
    lo = np.zeros(3) if lo is None else np.ascontiguousarray(lo, dtype=np.float64)
    hi = np.ones(3) if hi is None else np.ascontiguousarray(hi, dtype=np.float64)
    types, orders = np.ascontiguousarray(types, dtype=np.int32), np.ascontiguousarray(orders, dtype=np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    nv, ne, nt, nd = C.c_int(), C.c_int(), C.c_int(), C.c_int64()
    lib.mha_mesh_multi_sizes.argtypes = [C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 6
    _check(lib.mha_mesh_multi_sizes(dim, vp(ncell), len(types), vp(types), vp(orders), C.byref(nv), C.byref(ne), C.byref(nt), C.byref(nd)))
    nn = 2 ** dim
    m = dict(verts=np.zeros((nv.value, dim)), cell2vert=np.zeros((ne.value, nn), np.int32),
             lids=np.zeros((ne.value, nt.value), np.int32), offsets=np.zeros(nt.value, np.int32),
             orient=np.ones((ne.value, nt.value), np.int8), side_mask=np.zeros(nd.value, np.uint8),
             dof_var=np.zeros(nd.value, np.int32), ndof=nd.value, nelem=ne.value, n_tot=nt.value, dim=dim)
    lib.mha_mesh_structured_multi.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 9
    _check(lib.mha_mesh_structured_multi(dim, vp(ncell), vp(lo), vp(hi), len(types), vp(types), vp(orders), vp(m["verts"]),
                                         vp(m["cell2vert"]), vp(m["lids"]), vp(m["offsets"]), vp(m["orient"]),
                                         vp(m["side_mask"]), vp(m["dof_var"])))
    m["nodes"] = np.ascontiguousarray(m["verts"][m["cell2vert"]])
    return m


def block_patterns_host_apply(dim, nodes, lids, nrows, rowptr, colind, khat, factors, fixed=None, scale_u=1.0,
                              scale_t=1.0, chunk_elems=16, num_cus=8, max_patterns=256):
    """Host-only test hook (csrc/test_hooks.h, not part of the boundary): the block-pattern plan of the matrix-core
    row-owner Jacobian, walked as the kernel walks it.  -> (vals [nnz], (patterns, roles, workgroups, parts))."""
    lib = load_library()
    nodes, lids, rowptr, colind = _np(nodes, np.float64), _np(lids, np.int32), _np(rowptr, np.int32), _np(colind, np.int32)
    khat, factors = _np(khat, np.float64), _np(factors, np.float64)
    fx = None if fixed is None else _np(fixed, np.uint8)
    vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    nsym = khat.shape[0] - 1
    vals = np.full(len(colind), np.nan)
    counts = (C.c_int * 4)()
    f = lib.mha_test_block_patterns_host_apply
    f.argtypes = [C.c_int] * 6 + [C.c_void_p] * 7 + [C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    _check(f(int(dim), int(nrows), lids.shape[0], nodes.shape[1], lids.shape[1], nsym, vp(nodes), vp(lids), vp(rowptr),
             vp(colind), vp(fx), vp(khat), vp(factors), float(scale_u), float(scale_t), int(chunk_elems), int(num_cus),
             int(max_patterns), vp(vals), counts))
    return vals, tuple(counts)


def row_partition(dim, nodes, lids, nrows, rowptr, caps=None):
    """Host-only: the row-owner partition (mha_row_partition_*).  -> dict(row_ptr, rows, elem_ptr, elems, max_*)."""
    lib = load_library()
    nodes, lids, rowptr = _np(nodes, np.float64), _np(lids, np.int32), _np(rowptr, np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    h = C.c_void_p()
    cp = None if caps is None else _np(caps, np.int32)
    lib.mha_row_partition_build.argtypes = [C.c_int] * 4 + [C.c_void_p] * 5
    _check(lib.mha_row_partition_build(dim, lids.shape[0], lids.shape[1], int(nrows), vp(nodes), vp(lids), vp(rowptr),
                                       None if cp is None else vp(cp), C.byref(h)))
    try:
        nb, nr, ne = C.c_int(), C.c_int64(), C.c_int64()
        mr, me, ma = C.c_int(), C.c_int(), C.c_int()
        lib.mha_row_partition_sizes.argtypes = [C.c_void_p] * 7
        _check(lib.mha_row_partition_sizes(h, C.byref(nb), C.byref(nr), C.byref(ne), C.byref(mr), C.byref(me),
                                           C.byref(ma)))
        out = dict(row_ptr=np.zeros(nb.value + 1, np.int32), rows=np.zeros(nr.value, np.int32),
                   elem_ptr=np.zeros(nb.value + 1, np.int32), elems=np.zeros(ne.value, np.int32),
                   max_rows=mr.value, max_elems=me.value, max_entries=ma.value, num_blocks=nb.value)
        lib.mha_row_partition_get.argtypes = [C.c_void_p] * 5
        _check(lib.mha_row_partition_get(h, vp(out["row_ptr"]), vp(out["rows"]), vp(out["elem_ptr"]), vp(out["elems"])))
    finally:
        lib.mha_row_partition_destroy.argtypes = [C.c_void_p]
        lib.mha_row_partition_destroy.restype = None
        lib.mha_row_partition_destroy(h)
    return out


class Block:
    """One element block on one GPU (mha_context)."""

    def __init__(self, dim, order=1, quadrature=0, workset_size=100, device=0, physics="thermal", variables=None):
        """variables: list of (basis type, order) in the module's myvars order; default one HGRAD variable of `order`."""
        lib = load_library()
        d = BlockDesc()
        variables = [(BASIS_HGRAD, order)] if variables is None else list(variables)
        d.dimension, d.topology, d.num_vars = dim, (TOPO_QUAD4 if dim == 2 else TOPO_HEX8), len(variables)
        for v, (t, o) in enumerate(variables):
            d.basis_type[v], d.basis_order[v] = int(t), int(o)
        d.quadrature_degree, d.workset_size, d.device = quadrature, workset_size, device
        self._h = C.c_void_p()
        _check(lib.mha_block_create(C.byref(d), C.byref(self._h)))
        self.dim, self.order = dim, order
        self._keep = []
        if physics is not None:
            if physics not in PHYSICS_IDS:
                raise ValueError(physics)
            _check(lib.mha_physics_select(self._h, PHYSICS_IDS[physics]))

    def close(self):
        if getattr(self, "_h", None):
            load_library().mha_block_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- setup ------------------------------------------------------------
    def set_stream(self, cuda_stream_handle):
        _check(load_library().mha_set_stream(self._h, C.c_void_p(cuda_stream_handle)))

    def set_mesh(self, nodes, lids, offsets, nrows, fixed=None):
        nodes, lids, offsets = _np(nodes, np.float64), _np(lids, np.int32), _np(offsets, np.int32)
        fixed = None if fixed is None else _np(fixed, np.uint8)
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        _check(load_library().mha_set_mesh(self._h, lids.shape[0], vp(nodes), vp(lids), vp(offsets), int(nrows),
                                           vp(fixed)))
        self.nelem, self.n, self.nrows = lids.shape[0], lids.shape[1], int(nrows)

    def set_orientation(self, signs):
        signs = None if signs is None else _np(signs, np.int8)
        _check(load_library().mha_set_orientation(self._h, None if signs is None else signs.ctypes.data_as(C.c_void_p)))

    def set_graph(self, rowptr=None, colind=None):
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        if rowptr is not None:
            rowptr, colind = _np(rowptr, np.int32), _np(colind, np.int32)
        _check(load_library().mha_set_graph(self._h, vp(rowptr), vp(colind)))

    def get_graph(self):
        nr, nnz = C.c_int(), C.c_int64()
        _check(load_library().mha_get_graph_sizes(self._h, C.byref(nr), C.byref(nnz)))
        rowptr, colind = np.zeros(nr.value + 1, np.int32), np.zeros(nnz.value, np.int32)
        _check(load_library().mha_get_graph(self._h, rowptr.ctypes.data_as(C.c_void_p),
                                            colind.ctypes.data_as(C.c_void_p)))
        return rowptr, colind

    def set_function(self, name, value):
        """value: float | deck string ("8*pi*pi*sin(2*pi*x)") | ("sinprod", amp, freq[dim]) | CUDA tensor [E, numip]."""
        lib = load_library()
        if isinstance(value, str):
            _check(lib.mha_set_function_expression(self._h, name.encode(), value.encode()))
        elif isinstance(value, (int, float)):
            _check(lib.mha_set_function(self._h, name.encode(), FUNC_CONSTANT, float(value), None, None))
        elif isinstance(value, tuple) and value[0] == "sinprod":
            fr = np.zeros(3)
            fr[:len(value[2])] = value[2]
            _check(lib.mha_set_function(self._h, name.encode(), FUNC_SINPROD, float(value[1]),
                                        fr.ctypes.data_as(C.c_void_p), None))
        else:
            self._keep.append(value)
            _check(lib.mha_set_function(self._h, name.encode(), FUNC_IP_ARRAY, 0.0, None, _ptr(value)))

    def set_time(self, t):
        _check(load_library().mha_set_time(self._h, float(t)))

    def set_time_integration(self, transient, nsteps=0, nstages=0, stage=0, dt=1.0, butcher_A=None, butcher_b=None,
                             bdf=None):
        vp = lambda a: None if a is None else _np(a, np.float64).ctypes.data_as(C.c_void_p)
        A, b, w = (None if x is None else _np(x, np.float64) for x in (butcher_A, butcher_b, bdf))
        _check(load_library().mha_set_time_integration(self._h, int(transient), nsteps, nstages, stage, float(dt),
                                                       vp(A), vp(b), vp(w)))

    # -- assembly -----------------------------------------------------------
    def assemble_jacres(self, u, res, crs_vals=None, compute_jacobian=True, path=PATH_AUTO, u_prev=None,
                        u_stage=None, overwrite=False, adjoint=False, lump_mass=False, deterministic=False):
        """adjoint / lump_mass: the scatter options isAdjoint_ / lump_mass_ of the reference (MHA_ASSEMBLE_ADJOINT, _LUMP_MASS);
        deterministic: MHA_ASSEMBLE_DETERMINISTIC (bit-reproducible results on the affine row-owner path)."""
        flags = ((1 if compute_jacobian else 0) | (2 if overwrite else 0) | (4 if adjoint else 0) | (8 if lump_mass else 0) |
                 (16 if deterministic else 0))
        _check(load_library().mha_assemble_jacres(self._h, flags, path, _ptr(u), _ptr(u_prev),
                                                  _ptr(u_stage), _ptr(res), _ptr(crs_vals)))

    def compute_local_jacres(self, u, local_J, local_res, compute_jacobian=True, u_prev=None, u_stage=None):
        _check(load_library().mha_compute_local_jacres(self._h, int(compute_jacobian), _ptr(u), _ptr(u_prev),
                                                       _ptr(u_stage), _ptr(local_J), _ptr(local_res)))

    def get_mass(self, local_mass, masswts=None):
        w = None if masswts is None else _np(masswts, np.float64)
        _check(load_library().mha_get_mass(self._h, None if w is None else w.ctypes.data_as(C.c_void_p), _ptr(local_mass)))

    def swhdg_element_blocks(self, u, lam, res, blocks, side_types=None, farfield=None, u_prev=None, u_stage=None):
        ff = None if farfield is None else _np(farfield, np.float64)
        _check(load_library().mha_swhdg_element_blocks(self._h, _ptr(u), _ptr(u_prev), _ptr(u_stage), _ptr(lam),
                                                       _ptr(side_types), None if ff is None else ff.ctypes.data_as(C.c_void_p),
                                                       _ptr(res), _ptr(blocks)))

    def database_build(self):
        """identifyVolumetricDatabase with exact matching -> (basis_index[E], first_users[U])."""
        nu = C.c_int()
        _check(load_library().mha_database_build(self._h, C.byref(nu)))
        ne = self.info("num_elems")
        idx, fu = np.zeros(ne, np.int32), np.zeros(nu.value, np.int32)
        _check(load_library().mha_database_get(self._h, idx.ctypes.data_as(C.c_void_p), fu.ctypes.data_as(C.c_void_p)))
        return idx, fu

    def apply_mass_matrix_free(self, x, y, mode=MASS_ON_THE_FLY, masswts=None, mass=None, sparse=None):
        """y += M x (AssemblyManager::applyMassMatrixFree); mass: CUDA tensor of dense element / database masses, sparse: Sparse3D."""
        w = None if masswts is None else _np(masswts, np.float64)
        _check(load_library().mha_apply_mass_matrix_free(self._h, mode, None if w is None else w.ctypes.data_as(C.c_void_p),
                                                         _ptr(mass), None if sparse is None else sparse._h, _ptr(x), _ptr(y)))

    def swhdg_condensed_element(self, u, lam, schur=None, gvec=None, du=None, num_singular=None, side_types=None,
                                farfield=None, u_prev=None, u_stage=None):
        """Side + volume assembly + static condensation of every HDG element in one kernel (mha_swhdg_condensed_element):
        fills the given CUDA tensors schur [E][24][24], gvec [E][24], du [E][12]; nothing else leaves the chip."""
        ff = None if farfield is None else _np(farfield, np.float64)
        lib = load_library()
        lib.mha_swhdg_condensed_element.argtypes = [C.c_void_p] * 11
        _check(lib.mha_swhdg_condensed_element(self._h, _ptr(u), _ptr(u_prev), _ptr(u_stage), _ptr(lam), _ptr(side_types),
                                               None if ff is None else ff.ctypes.data_as(C.c_void_p), _ptr(schur), _ptr(gvec),
                                               _ptr(du), _ptr(num_singular)))

    def swhdg_subgrid_solve(self, u, lam, max_iter, tol, side_types=None, farfield=None, u_prev=None, u_stage=None,
                            want_condensed=True):
        """SubGridDtN_Solver::nonlinearSolver on the device (u updated in place) -> dict(iters, resnorm, schur, gvec, num_singular).
        No host synchronisation happens inside the call; the tensors are ready once the stream is."""
        import torch
        nb = C.c_int64()
        _check(load_library().mha_swhdg_subgrid_workspace_bytes(self._h, C.byref(nb)))
        E = lam.shape[0]
        # one workspace per Block, reallocated only when the size changes (a new mesh): a time loop that calls this once
        # per macro step must not grow the heap (0.8 GB per call at 256^2 elements)
        ws = getattr(self, "_subgrid_ws", None)
        if ws is None or ws.numel() != nb.value or ws.device != u.device:
            ws = torch.empty(nb.value, dtype=torch.uint8, device=u.device)
            self._subgrid_ws = ws
        out = dict(iters=torch.zeros(E, dtype=torch.int32, device=u.device), resnorm=torch.zeros(E, dtype=torch.float64, device=u.device),
                   num_singular=torch.zeros(1, dtype=torch.int32, device=u.device))
        if want_condensed:
            out["schur"] = torch.zeros((E, 24, 24), dtype=torch.float64, device=u.device)
            out["gvec"] = torch.zeros((E, 24), dtype=torch.float64, device=u.device)
        ff = None if farfield is None else _np(farfield, np.float64)
        _check(load_library().mha_swhdg_subgrid_solve(self._h, _ptr(u), _ptr(u_prev), _ptr(u_stage), _ptr(lam), _ptr(side_types),
                                                      None if ff is None else ff.ctypes.data_as(C.c_void_p), int(max_iter),
                                                      float(tol), _ptr(ws), nb.value, _ptr(out.get("schur")), _ptr(out.get("gvec")),
                                                      _ptr(out["iters"]), _ptr(out["resnorm"]), _ptr(out["num_singular"])))
        return out

    def scatter_local(self, local_J, local_res, res, crs_vals):
        _check(load_library().mha_scatter_local(self._h, _ptr(local_J), _ptr(local_res), _ptr(res), _ptr(crs_vals)))

    def dirichlet_lift(self, u, fixed_soln=None, scalar=0.0):
        """u[row] = fixed_soln[row] (or scalar) on the fixed rows (SolverManager::setDirichlet)."""
        lib = load_library()
        lib.mha_dirichlet_lift.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        _check(lib.mha_dirichlet_lift(self._h, _ptr(u), _ptr(fixed_soln), float(scalar)))

    def apply_dbc_diag(self, crs_vals):
        _check(load_library().mha_apply_dbc_diag(self._h, _ptr(crs_vals)))

    def gather(self, vec, out):
        _check(load_library().mha_gather(self._h, _ptr(vec), _ptr(out)))

    # -- boundary groups -------------------------------------------------------
    def add_boundary_group(self, sidename, bc_type, elem_ids, side_ids):
        e, s_ = _np(elem_ids, np.int32), _np(side_ids, np.int32)
        gid = C.c_int()
        _check(load_library().mha_add_boundary_group(self._h, sidename.encode(), bc_type, len(e),
                                                     e.ctypes.data_as(C.c_void_p), s_.ctypes.data_as(C.c_void_p),
                                                     C.byref(gid)))
        return gid.value

    def compute_flux(self, group_id, u, flux, dflux_du=None, dflux_daux=None, u_prev=None, u_stage=None):
        """<module>::computeFlux on one boundary group (mha_compute_flux): flux [num][nqs] (+ derivative arrays)."""
        lib = load_library()
        lib.mha_compute_flux.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6
        _check(lib.mha_compute_flux(self._h, int(group_id), _ptr(u), _ptr(u_prev), _ptr(u_stage), _ptr(flux), _ptr(dflux_du),
                                    _ptr(dflux_daux)))

    def add_flux_group(self, sidename, varname, elem_ids, side_ids):
        """The "Flux" condition of PhysicsInterface::fluxConditions for one variable; data = function "Flux <var> <side>"."""
        e, s_ = _np(elem_ids, np.int32), _np(side_ids, np.int32)
        gid = C.c_int()
        _check(load_library().mha_add_flux_group(self._h, sidename.encode(), varname.encode(), len(e),
                                                 e.ctypes.data_as(C.c_void_p), s_.ctypes.data_as(C.c_void_p),
                                                 C.byref(gid)))
        return gid.value

    def add_dirichlet_group(self, sidename, varname, elem_ids, side_ids):
        """Strong Dirichlet entries of one variable (setDirichlet); data = function "Dirichlet <var> <side>"."""
        e, s_ = _np(elem_ids, np.int32), _np(side_ids, np.int32)
        gid = C.c_int()
        _check(load_library().mha_add_dirichlet_group(self._h, sidename.encode(), varname.encode(), len(e),
                                                      e.ctypes.data_as(C.c_void_p), s_.ctypes.data_as(C.c_void_p),
                                                      C.byref(gid)))
        return gid.value

    def set_initial(self, rhs, mass_vals, lump_mass=False):
        """AssemblyManager::setInitial: rhs += (initial, basis), mass_vals += mass matrix (CRS), zero rows -> identity."""
        _check(load_library().mha_set_initial(self._h, int(lump_mass), _ptr(rhs), _ptr(mass_vals)))

    def set_initial_nodal(self, initial):
        """AssemblyManager::setInitial(initial): values of "initial <var>" at the vertices (HGRAD order 1)."""
        _check(load_library().mha_set_initial_nodal(self._h, _ptr(initial)))

    def set_dirichlet(self, rhs, mass_vals, lump_mass=False):
        """AssemblyManager::setDirichlet: boundary mass + data on the fixed rows, identity on the others."""
        _check(load_library().mha_set_dirichlet(self._h, int(lump_mass), _ptr(rhs), _ptr(mass_vals)))

    def clear_boundary_groups(self):
        _check(load_library().mha_clear_boundary_groups(self._h))

    def num_boundary_groups(self):
        return load_library().mha_num_boundary_groups(self._h)

    def assemble_boundary(self, u, res, crs_vals=None, compute_jacobian=True, u_prev=None, u_stage=None, flags=None):
        flags = (1 if compute_jacobian else 0) if flags is None else flags
        _check(load_library().mha_assemble_boundary(self._h, flags, _ptr(u), _ptr(u_prev), _ptr(u_stage), _ptr(res),
                                                    _ptr(crs_vals)))

    def boundary_update(self, gid):
        _check(load_library().mha_boundary_update(self._h, gid))

    def boundary_view(self, gid, name):
        p, ext, rank = C.c_void_p(), (C.c_int64 * 4)(), C.c_int()
        _check(load_library().mha_boundary_view(self._h, gid, name.encode(), C.byref(p), ext, C.byref(rank)))
        return p.value, tuple(ext[k] for k in range(rank.value))

    def boundary_view_numpy(self, gid, name):
        import torch
        ptr, shape = self.boundary_view(gid, name)
        out = np.zeros(shape, np.float64)
        torch.cuda.synchronize()
        rc = _hip_memcpy_dtoh(out.ctypes.data, ptr, out.size * out.itemsize)
        assert rc == 0, rc
        return out

    def set_physics_parameter(self, name, value):
        _check(load_library().mha_set_physics_parameter(self._h, name.encode(), float(value)))

    # -- workset views ---------------------------------------------------------
    def num_worksets(self):
        return load_library().mha_num_worksets(self._h)

    def workset_update(self, index):
        _check(load_library().mha_workset_update(self._h, index))

    def workset_compute_solution(self, u, u_prev=None, u_stage=None):
        """Workset::computeSoln on the current workset; the fields become views ("e", "grad(e)[x]", "u[x]", "div(u)", ...)."""
        _check(load_library().mha_workset_compute_solution(self._h, _ptr(u), _ptr(u_prev), _ptr(u_stage)))

    def workset_compute_residual(self, u, compute_jacobian=True, u_prev=None, u_stage=None):
        """resetResidual + volumeResidual on the current workset; views "res" and "res.dx"."""
        _check(load_library().mha_workset_compute_residual(self._h, int(compute_jacobian), _ptr(u), _ptr(u_prev),
                                                           _ptr(u_stage)))

    def workset_view(self, name):
        """-> (device pointer, shape tuple)."""
        p, ext, rank = C.c_void_p(), (C.c_int64 * 4)(), C.c_int()
        _check(load_library().mha_workset_view(self._h, name.encode(), C.byref(p), ext, C.byref(rank)))
        return p.value, tuple(ext[k] for k in range(rank.value))

    def workset_view_numpy(self, name):
        """Copy a workset view to the host (tests)."""
        import torch
        ptr, shape = self.workset_view(name)
        is_int = name in ("LIDs", "offsets")
        count = int(np.prod(shape))
        out = np.zeros(shape, np.int32 if is_int else np.float64)
        torch.cuda.synchronize()
        rc = _hip_memcpy_dtoh(out.ctypes.data, ptr, count * out.itemsize)
        assert rc == 0, rc
        return out

    # -- introspection --------------------------------------------------------------
    def info(self, key):
        v = C.c_int64()
        _check(load_library().mha_get_info(self._h, key.encode(), C.byref(v)))
        return v.value

    def set_timing(self, on=True):
        _check(load_library().mha_set_timing(self._h, int(on)))

    def last_kernel_ms(self):
        v = C.c_double()
        _check(load_library().mha_get_last_kernel_ms(self._h, C.byref(v)))
        return v.value


_hip = None


def _hip_memcpy_dtoh(dst_host, src_dev, nbytes):
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return _hip.hipMemcpy(C.c_void_p(dst_host), C.c_void_p(src_dev), nbytes, 2)
