"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header declares,
its host-only entry points agree with the oracle bit for bit, and it fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import mrhyde_amd
from mrhyde_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mrhyde_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mha_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = mrhyde_amd.load_library()
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "libmrhyde_amd.so does not export %s" % s
    assert sorted(api.EXPORTS) == syms, "api.EXPORTS out of sync with include/mrhyde_amd.h"
    assert "gfx950" in mrhyde_amd.version()


@pytest.mark.parametrize("dim,order,ncell", [(2, 1, (5, 3)), (2, 1, (40, 40)), (2, 4, (3, 2)), (3, 1, (3, 2, 4)),
                                              (3, 2, (3, 4, 2)), (2, 2, (1, 1)), (3, 2, (1, 1, 1))])
def test_mesh_and_dof_map_bit_exact_vs_oracle(oracle, dim, order, ncell):
    a = mrhyde_amd.mesh_structured(dim, order, ncell)
    b = oracle.mesh_structured(dim, order, ncell)
    for k in ("verts", "cell2vert", "lids", "offsets", "boundary", "nodes"):
        assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, k
        assert np.array_equal(a[k], b[k]), k  # integer maps bit-exact; coordinates bit-exact too
    assert a["ndof"] == b["ndof"] and a["nelem"] == b["nelem"]


def test_simplemesh_reference_map():
    """reference: src/tools/simplemeshmanager.hpp:659-675, offsets {0,1,3,2} discretizationInterface.cpp:302"""
    m = mrhyde_amd.mesh_structured(2, 1, (4, 3))
    assert m["offsets"].tolist() == [0, 1, 3, 2]
    nx = 4
    assert m["lids"][0].tolist() == [0, 1, nx + 2, nx + 1]
    assert m["lids"][5].tolist() == [1 * (nx + 1) + 1, 1 * (nx + 1) + 2, 2 * (nx + 1) + 2, 2 * (nx + 1) + 1]


def test_bad_arguments_report_errors():
    lib = mrhyde_amd.load_library()
    nv, ne, nd = C.c_int(), C.c_int(), C.c_int64()
    nc = np.array([0, 2], np.int32)
    rc = lib.mha_mesh_sizes(2, 1, nc.ctypes.data_as(C.c_void_p), C.byref(nv), C.byref(ne), C.byref(nd))
    assert rc == 1 and b"positive" in lib.mha_last_error()
    rc = lib.mha_mesh_sizes(4, 1, nc.ctypes.data_as(C.c_void_p), C.byref(nv), C.byref(ne), C.byref(nd))
    assert rc == 1 and b"dimension" in lib.mha_last_error()
    assert lib.mha_num_worksets(None) == 0


def test_no_gpu_means_loud_failure_not_fallback():
    if mrhyde_amd.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(mrhyde_amd.MhaError) as ei:
        mrhyde_amd.Block(2, 1)
    assert ei.value.code == 3 and "no CPU fallback" in str(ei.value)


def test_export_plan_rejects_bad_lists():
    """mha_export_plan_create range-checks every index against the array sizes and refuses duplicate receive targets of
    one neighbour (the unpack kernel adds with plain read-modify-write) -- before anything touches a device."""
    import ctypes as C
    import numpy as np
    import mrhyde_amd
    lib = mrhyde_amd.load_library()
    lib.mha_export_plan_create.argtypes = [C.c_int] + [C.c_void_p] * 9 + [C.c_int64, C.c_int64, C.c_void_p]
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    ranks = np.array([1], np.int32)
    one, zero = np.array([0, 1], np.int64), np.array([0, 0], np.int64)
    two = np.array([0, 2], np.int64)

    def create(sv, sr, rv, rr, nnz, nrows, svp=one, srp=one, rvp=one, rrp=one):
        h = C.c_void_p()
        a = lambda x: np.array(x, np.int32)
        rc = lib.mha_export_plan_create(1, vp(ranks), vp(svp), vp(a(sv)), vp(srp), vp(a(sr)), vp(rvp), vp(a(rv)), vp(rrp),
                                        vp(a(rr)), nnz, nrows, C.byref(h))
        return rc, lib.mha_last_error().decode() if rc else ""

    lib.mha_last_error.restype = C.c_char_p
    rc, msg = create([10], [0], [0], [0], 10, 4)
    assert rc == 1 and "outside the value array" in msg
    rc, msg = create([0], [4], [0], [0], 10, 4)
    assert rc == 1 and "outside the residual" in msg
    rc, msg = create([0], [0], [-2], [0], 10, 4)
    assert rc == 1 and "receive target" in msg
    rc, msg = create([0, 1], [0], [3, 3], [0], 10, 4, svp=two, rvp=two)
    assert rc == 1 and "twice" in msg
