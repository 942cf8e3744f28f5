"""Host logic of the row-owner partition (no GPU): every row owned exactly once, every incident
element of an owned row is in the block's element list, caps respected, aligned chunks on tensor meshes."""
import numpy as np
import pytest

import mrhyde_amd


def _check_partition(m, rowptr, part, caps=None):
    nrows = m["ndof"]
    rows = part["rows"]
    assert len(rows) == nrows and np.array_equal(np.sort(rows), np.arange(nrows)), "each row owned exactly once"
    inc = [[] for _ in range(nrows)]
    for e, l in enumerate(m["lids"]):
        for r in l:
            inc[r].append(e)
    for b in range(part["num_blocks"]):
        br = rows[part["row_ptr"][b]:part["row_ptr"][b + 1]]
        be = part["elems"][part["elem_ptr"][b]:part["elem_ptr"][b + 1]]
        assert np.all(np.diff(br) > 0) and np.all(np.diff(be) > 0)
        need = sorted(set(e for r in br for e in inc[r]))
        assert need == be.tolist(), "block %d element list" % b
        acc = int(sum(rowptr[r + 1] - rowptr[r] for r in br))
        assert acc <= part["max_entries"] and len(br) <= part["max_rows"] and len(be) <= part["max_elems"]
        if caps is not None:
            assert acc <= caps[1] and len(br) <= caps[2] and len(be) <= caps[3]


@pytest.mark.parametrize("dim,order,ncell", [(3, 2, (4, 4, 4)), (3, 2, (5, 3, 2)), (3, 1, (4, 4, 4)), (2, 1, (9, 7)),
                                              (2, 4, (4, 4)), (2, 2, (8, 8))])
def test_partition_covers_rows_and_elements(oracle, dim, order, ncell):
    m = mrhyde_amd.mesh_structured(dim, order, ncell)
    rowptr, _ = oracle.build_graph(m["ndof"], m["lids"])
    part = mrhyde_amd.row_partition(dim, m["nodes"], m["lids"], m["ndof"], rowptr)
    _check_partition(m, rowptr, part)


def test_partition_aligned_chunks_on_tensor_mesh(oracle):
    """64^3-style case in small: interior 2x2x2 element chunks own 64 rows / 4096 entries / touch 27 elements."""
    m = mrhyde_amd.mesh_structured(3, 2, (8, 8, 8))
    rowptr, _ = oracle.build_graph(m["ndof"], m["lids"])
    part = mrhyde_amd.row_partition(3, m["nodes"], m["lids"], m["ndof"], rowptr, caps=[8, 4224, 128, 32])
    _check_partition(m, rowptr, part, caps=[8, 4224, 128, 32])
    nr = np.diff(part["row_ptr"])
    ne = np.diff(part["elem_ptr"])
    assert part["max_elems"] == 27
    interior = (nr == 64) & (ne == 27)
    assert interior.sum() == 8  # chunks touching neither the low nor the high faces: 2 of 4 per direction


def test_partition_perturbed_and_shuffled_elements(oracle):
    """Element numbering and vertex perturbation must not matter for validity."""
    rng = np.random.default_rng(4)
    m = mrhyde_amd.mesh_structured(3, 1, (5, 4, 3))
    v = m["verts"] + 0.02 * rng.uniform(-1, 1, m["verts"].shape)
    perm = rng.permutation(m["nelem"])
    m["lids"] = np.ascontiguousarray(m["lids"][perm])
    m["nodes"] = np.ascontiguousarray(v[m["cell2vert"][perm]])
    rowptr, _ = oracle.build_graph(m["ndof"], m["lids"])
    part = mrhyde_amd.row_partition(3, m["nodes"], m["lids"], m["ndof"], rowptr, caps=[8, 300, 16, 20])
    _check_partition(m, rowptr, part, caps=[8, 300, 16, 20])


def test_partition_rejects_impossible_caps(oracle):
    m = mrhyde_amd.mesh_structured(2, 1, (3, 3))
    rowptr, _ = oracle.build_graph(m["ndof"], m["lids"])
    with pytest.raises(mrhyde_amd.MhaError) as ei:
        mrhyde_amd.row_partition(2, m["nodes"], m["lids"], m["ndof"], rowptr, caps=[4, 4, 8, 8])
    assert ei.value.code == 1 and "exceeds" in str(ei.value)


# ---- block patterns of the matrix-core row-owner Jacobian (csrc/block_pattern.{hpp,cpp}), walked on the host as the
# ---- kernel walks them (workgroup -> wavefront -> part -> block -> MFMA panel)

def _pattern_reference(m, rowptr, colind, khat, factors, fixed=None, scale=None):
    """vals[row(e,si)][col(e,sj)] += sum_c scale_c factors[e][c] * khat[c][si][sj], fixed rows zero: plain element loop."""
    n = m["lids"].shape[1]
    vals = np.zeros(len(colind))
    ncomp = khat.shape[0]
    sc = np.ones(ncomp) if scale is None else np.asarray(scale)
    for e, L in enumerate(m["lids"]):
        K = np.tensordot(factors[e, :ncomp] * sc, khat, axes=(0, 0)).reshape(n, n)
        for si in range(n):
            r = L[si]
            if fixed is not None and fixed[r]:
                continue
            lo, hi = rowptr[r], rowptr[r + 1]
            vals[lo + np.searchsorted(colind[lo:hi], L)] += K[si]
    return vals


@pytest.mark.parametrize("dim,order,ncell,chunk", [(3, 2, (8, 4, 4), 16), (3, 2, (4, 3, 3), 16), (3, 1, (4, 4, 3), 16),
                                                    (2, 1, (7, 5), 16), (2, 2, (8, 8), 16), (2, 4, (3, 3), 4),
                                                    (3, 2, (4, 4, 4), 8)])
def test_block_patterns_reproduce_the_element_scatter(oracle, dim, order, ncell, chunk):
    m = mrhyde_amd.mesh_structured(dim, order, ncell)
    nrows = m["ndof"]
    rowptr, colind = oracle.build_graph(nrows, m["lids"])
    n = m["lids"].shape[1]
    nsym = dim * (dim + 1) // 2
    rng = np.random.default_rng(31)
    khat = rng.uniform(-1, 1, (nsym + 1, n * n))      # any tables: the grouping does not look at their values
    factors = rng.uniform(0.5, 2.0, (m["nelem"], nsym + 1))
    fixed = (rng.uniform(size=nrows) < 0.1).astype(np.uint8)
    su, st = 1.7, -0.3
    vals, counts = mrhyde_amd.block_patterns_host_apply(dim, m["nodes"], m["lids"], nrows, rowptr, colind, khat, factors,
                                                        fixed, scale_u=su, scale_t=st, chunk_elems=chunk,
                                                        max_patterns=4096)
    ref = _pattern_reference(m, rowptr, colind, khat, factors, fixed, scale=[su] * nsym + [st])
    assert not np.any(np.isnan(vals)), "every CRS entry written exactly once"
    assert np.max(np.abs(vals - ref)) <= 1e-12 * np.max(np.abs(ref))
    assert counts[0] >= 1 and counts[1] >= counts[0] and counts[2] == 8 and counts[3] >= counts[1]  # one workgroup per CU


def test_block_patterns_of_a_structured_mesh_are_few(oracle):
    """64^3-style case in small: 4x2x2-element chunks; per direction a chunk is first / interior / last."""
    m = mrhyde_amd.mesh_structured(3, 2, (16, 8, 8))
    nrows = m["ndof"]
    rowptr, colind = oracle.build_graph(nrows, m["lids"])
    n = m["lids"].shape[1]
    rng = np.random.default_rng(33)
    khat = rng.uniform(-1, 1, (7, n * n))
    factors = rng.uniform(0.5, 2.0, (m["nelem"], 7))
    vals, counts = mrhyde_amd.block_patterns_host_apply(3, m["nodes"], m["lids"], nrows, rowptr, colind, khat, factors,
                                                        m["boundary"], num_cus=32)
    ref = _pattern_reference(m, rowptr, colind, khat, factors, m["boundary"])
    assert np.max(np.abs(vals - ref)) <= 1e-12 * np.max(np.abs(ref))
    assert counts[0] == 27 and counts[1] == 27


def test_block_patterns_reject_unstructured_numbering(oracle):
    """A random renumbering of the dofs gives (nearly) every block its own pattern: the grouping must refuse
    (the caller then keeps the row-block kernel) rather than build one set of tables per block."""
    m = mrhyde_amd.mesh_structured(3, 2, (8, 8, 8))
    rng = np.random.default_rng(32)
    perm = rng.permutation(m["ndof"]).astype(np.int32)
    lids = perm[m["lids"]]
    rowptr, colind = oracle.build_graph(m["ndof"], lids)
    n = lids.shape[1]
    khat = rng.uniform(-1, 1, (7, n * n))
    factors = rng.uniform(0.5, 2.0, (lids.shape[0], 7))
    with pytest.raises(mrhyde_amd.MhaError, match="patterns"):
        mrhyde_amd.block_patterns_host_apply(3, m["nodes"], lids, m["ndof"], rowptr, colind, khat, factors, max_patterns=16)
