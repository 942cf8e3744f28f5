"""Export(ADD) of shared-DOF rows between element-block shards: the caller side of mha_export_* (include/mrhyde_amd.h).

Reference semantics (src/interfaces/linearAlgebraInterface.hpp:296-337): every rank assembles into its OVERLAPPED
residual / CRS matrix (owned + ghost rows); `doExport(..., Tpetra::ADD)` then sums the ghost-row contributions into
the owning rank's rows.  Which rows are shared, who owns them (the lowest rank that has the dof) and in which order
their entries travel is agreed ONCE, at setup, from the global ids of every rank's rows -- any partition: HGRAD planes
between slabs, HDIV face dofs, HDG trace rows, unequal slabs.  Per assembly:

    pack (HIP kernel, C ABI)  ->  point-to-point send / recv between neighbours  ->  unpack-add (HIP kernel, C ABI)

The transport is torch.distributed P2P (backend "nccl" = RCCL: ncclSend / ncclRecv in one group; neighbouring slabs
have a direct xGMI link), or, with MHA_EXPORT_TRANSPORT=rccl, the library's own RCCL communicator (mha_export_add).
Entries of a shared row whose column the owner does not have (couplings to the sender's interior dofs: off-rank
columns of an owned row) stay in the receive buffer: `remote_vals(p)`.

On CPU tensors (the world-size-2 gloo tests, no GPU in the build container) pack / unpack are plain torch indexing of
the same lists; on CUDA tensors they are the library's kernels -- there is no silent fallback on a GPU box.
"""
import ctypes as C
import os

import numpy as np


def _ranges(starts, lengths):
    """Concatenation of arange(s, s + l) for every (s, l)."""
    starts, lengths = np.asarray(starts, dtype=np.int64), np.asarray(lengths, dtype=np.int64)
    ends = np.cumsum(lengths)
    total = int(ends[-1]) if len(ends) else 0
    return np.repeat(starts - (ends - lengths), lengths) + np.arange(total, dtype=np.int64)


class _DevView:
    """A device buffer owned by the library, seen by torch without a copy."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class SharedRowExport:
    """Setup + per-assembly exchange for one rank.

    gid [nrows] int64: global id of every local row (= of every local column: the overlapped matrix is square in the
    local numbering); rowptr / colind: the rank's local CRS graph (colind ascending inside a row)."""

    def __init__(self, gid, rowptr, colind, rank, world, device):
        import torch
        import torch.distributed as dist
        self.rank, self.world, self.device = rank, world, device
        gid = np.ascontiguousarray(gid, dtype=np.int64)
        rowptr = np.asarray(rowptr, dtype=np.int64)
        colind = np.asarray(colind)
        n = len(gid)
        assert len(rowptr) == n + 1 and len(np.unique(gid)) == n, "one global id per local row"
        self._nrows, self._nnz = n, int(rowptr[-1])
        self.neighbors, self.send_val, self.send_row, self.recv_val, self.recv_row = [], {}, {}, {}, {}
        self._plan = None
        if world == 1:
            return
        # ---- who else has my dofs, who owns them (the lowest rank that has the dof) ----
        gids_all = [None] * world
        dist.all_gather_object(gids_all, gid)
        owner = np.full(n, rank, dtype=np.int64)
        for p in range(rank - 1, -1, -1):
            owner[np.isin(gid, gids_all[p])] = p
        order = np.argsort(gid, kind="stable")
        # ---- what I send: whole rows, grouped by owner, ascending global id; the owner needs their column ids ----
        payload = {}
        for o in np.unique(owner):
            if o == rank:
                continue
            rows = order[owner[order] == o]
            lens = rowptr[rows + 1] - rowptr[rows]
            idx = _ranges(rowptr[rows], lens)
            self.send_val[int(o)] = idx
            self.send_row[int(o)] = rows.astype(np.int64)
            payload[int(o)] = (gid[rows], lens, gid[colind[idx]])
        payloads = [None] * world
        dist.all_gather_object(payloads, payload)
        # ---- what I receive: every entry's place in my arrays (or -1: a column I do not have) ----
        gsorted = gid[order]
        for p in range(world):
            if p == rank or not payloads[p] or rank not in payloads[p]:
                continue
            rg, lens, cg = payloads[p][rank]
            pos = np.searchsorted(gsorted, rg)
            assert np.all(pos < n) and np.array_equal(gsorted[pos], rg), "a sender lists a row this rank does not have"
            rows = order[pos]
            cpos = np.minimum(np.searchsorted(gsorted, cg), n - 1)
            have = gsorted[cpos] == cg
            cols = np.where(have, order[cpos], -1)
            rrep = np.repeat(rows, lens)
            # position of (row, col) in my CRS: binary search inside each row
            tgt = np.full(len(cg), -1, dtype=np.int64)
            lo, hi = rowptr[rrep].copy(), rowptr[rrep + 1].copy()
            active = have.copy()
            while np.any(active & (lo < hi)):
                a = active & (lo < hi)
                mid = (lo + hi) // 2
                c = colind[np.minimum(mid, len(colind) - 1)]
                less = a & (c < cols)
                more = a & (c > cols)
                hit = a & (c == cols)
                tgt[hit] = mid[hit]
                active &= ~hit
                lo[less] = mid[less] + 1
                hi[more] = mid[more]
            self.recv_val[p] = tgt
            self.recv_row[p] = rows.astype(np.int64)
        self.neighbors = sorted(set(self.send_val) | set(self.recv_val))
        e = np.zeros(0, np.int64)
        for k in self.neighbors:
            self.send_val.setdefault(k, e)
            self.send_row.setdefault(k, e)
            self.recv_val.setdefault(k, e)
            self.recv_row.setdefault(k, e)
        self._on_gpu = torch.device(device).type == "cuda"
        self._comm = None
        if self._on_gpu:
            self._create_plan()
            if os.environ.get("MHA_EXPORT_TRANSPORT") == "rccl":
                self._create_comm()
        else:
            t = lambda a: torch.tensor(np.ascontiguousarray(a, dtype=np.int64))
            self._t = {k: (t(self.send_val[k]), t(self.send_row[k]), t(self.recv_val[k]), t(self.recv_row[k])) for k in self.neighbors}
            self._send = {k: torch.zeros(len(self.send_val[k]) + len(self.send_row[k]), dtype=torch.float64) for k in self.neighbors}
            self._recv = {k: torch.zeros(len(self.recv_val[k]) + len(self.recv_row[k]), dtype=torch.float64) for k in self.neighbors}

    # ---- device side: the library's plan ----
    def _create_plan(self):
        import torch
        from .api import _check, load_library
        lib = load_library()
        nb = self.neighbors

        def csr(d, dtype=np.int32):
            ptr = np.zeros(len(nb) + 1, np.int64)
            for i, k in enumerate(nb):
                ptr[i + 1] = ptr[i] + len(d[k])
            flat = np.concatenate([d[k] for k in nb]) if nb else np.zeros(0)
            assert flat.size == 0 or flat.max() < 2 ** 31
            return ptr, np.ascontiguousarray(flat, dtype=dtype)
        self._lists = [csr(self.send_val), csr(self.send_row), csr(self.recv_val), csr(self.recv_row)]
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        ranks = np.ascontiguousarray(nb, dtype=np.int32)
        self._plan = C.c_void_p()
        lib.mha_export_plan_create.argtypes = [C.c_int] + [C.c_void_p] * 9 + [C.c_int64, C.c_int64, C.c_void_p]
        args = [vp(ranks)]
        for ptr, flat in self._lists:
            args += [vp(ptr), vp(flat)]
        _check(lib.mha_export_plan_create(len(nb), *args, self._nnz, self._nrows, C.byref(self._plan)))
        self._send, self._recv = {}, {}
        lib.mha_export_buffers.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4
        for i, k in enumerate(nb):
            sp, rp, sc, rc = C.c_void_p(), C.c_void_p(), C.c_int64(), C.c_int64()
            _check(lib.mha_export_buffers(self._plan, i, C.byref(sp), C.byref(sc), C.byref(rp), C.byref(rc)))
            mk = lambda p, c: (torch.as_tensor(_DevView(p.value, c.value), device=self.device) if c.value else
                               torch.zeros(0, dtype=torch.float64, device=self.device))
            self._send[k], self._recv[k] = mk(sp, sc), mk(rp, rc)
        lib.mha_export_pack.argtypes = [C.c_void_p] * 4
        lib.mha_export_unpack_add.argtypes = [C.c_void_p] * 4

    def _create_comm(self):
        import torch
        import torch.distributed as dist
        from .api import _check, load_library
        lib = load_library()
        buf = (C.c_char * 128)()
        if self.rank == 0:
            _check(lib.mha_comm_unique_id(buf))
        t = torch.tensor(list(bytes(buf)), dtype=torch.uint8, device=self.device)
        dist.broadcast(t, 0)
        idb = (C.c_char * 128).from_buffer_copy(bytes(t.cpu().numpy().tobytes()))
        self._comm = C.c_void_p()
        _check(lib.mha_comm_create(self.world, self.rank, idb, C.byref(self._comm)))
        lib.mha_export_add.argtypes = [C.c_void_p] * 5

    def bytes_on_wire(self):
        return 8 * sum(len(self.send_val[k]) + len(self.send_row[k]) for k in self.neighbors)

    def remote_vals(self, p):
        """Values received from rank p whose column this rank does not have (off-rank columns of owned rows), and their
        positions in p's message."""
        idx = np.flatnonzero(self.recv_val[p] < 0)
        return self._recv[p][:len(self.recv_val[p])][idx], idx

    def export_add(self, res, vals, stream=None):
        """Sum ghost-row contributions into the owner.  res / vals: this rank's overlapped arrays (vals may be None)."""
        import torch
        import torch.distributed as dist
        if self.world == 1 or not self.neighbors:
            return
        if self._on_gpu:
            from .api import _check, load_library
            lib = load_library()
            assert res.is_cuda and (vals is None or vals.is_cuda), "the plan lives on the GPU"
            cur = torch.cuda.current_stream().cuda_stream
            # the staging copies below run on torch's current stream: a foreign stream would race them against the pack /
            # unpack kernels
            assert stream is None or int(stream) == int(cur), "export_add: pass torch's current stream (or none)"
            s = C.c_void_p(cur)
            vptr = C.c_void_p(vals.data_ptr()) if vals is not None else None
            if self._comm is not None:  # the library's own RCCL communicator: pack + ncclSend / ncclRecv + unpack
                _check(lib.mha_export_add(self._plan, self._comm, vptr, C.c_void_p(res.data_ptr()), s))
                return
            _check(lib.mha_export_pack(self._plan, vptr, C.c_void_p(res.data_ptr()), s))
        else:
            for k in self.neighbors:
                sv, sr, _, _ = self._t[k]
                if vals is not None:
                    self._send[k][:len(sv)] = vals[sv]
                self._send[k][len(sv):] = res[sr]
        # The plan's buffers are the library's (seen by torch through __cuda_array_interface__).  gloo (rehearsal of N > 1
        # on one GPU) moves host copies; with RCCL the buffers go on the wire as they are (MHA_EXPORT_STAGED=1 restores the
        # torch-owned device copies of round 2).  A residual-only exchange (vals is None) moves the residual halves only.
        gloo = self._on_gpu and dist.get_backend() == "gloo"
        staged = self._on_gpu and (gloo or os.environ.get("MHA_EXPORT_STAGED", "0") == "1")
        ops, keep = [], []
        for k in self.neighbors:
            s0 = 0 if vals is not None else len(self.send_val[k])
            r0 = 0 if vals is not None else len(self.recv_val[k])
            sb, rb = self._send[k][s0:], self._recv[k][r0:]
            if len(sb):
                b = (sb.cpu() if gloo else sb.clone()) if staged else sb
                keep.append(b)
                ops.append(dist.P2POp(dist.isend, b, k))
            if len(rb):
                b = (rb.cpu() if gloo else torch.empty_like(rb)) if staged else rb
                keep.append((rb, b))
                ops.append(dist.P2POp(dist.irecv, b, k))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if staged:
            for item in keep:
                if isinstance(item, tuple):
                    item[0].copy_(item[1])
        if self._on_gpu:
            _check(lib.mha_export_unpack_add(self._plan, vptr, C.c_void_p(res.data_ptr()), s))
        else:
            for k in self.neighbors:  # ascending rank: the same order as the library's unpack
                _, _, rv, rr = self._t[k]
                buf = self._recv[k]
                if vals is not None and len(rv):
                    ok = rv >= 0
                    vals.index_add_(0, rv[ok], buf[:len(rv)][ok])
                res.index_add_(0, rr, buf[len(rv):])

    def close(self):
        from .api import load_library
        lib = load_library()
        if getattr(self, "_comm", None):
            lib.mha_comm_destroy(self._comm)
            self._comm = None
        if getattr(self, "_plan", None):
            lib.mha_export_plan_destroy(self._plan)
            self._plan = None


def slab_gids(nrows, plane_rows, rank):
    """Global ids of a z-slab's rows when every slab has the same dof layout and the shared plane is the first /
    last `plane_rows` rows of a slab (lexicographic node numbering, z slowest; any set of HGRAD variables numbered
    lattice-site-major): the slabs' numberings overlap by one plane."""
    return np.arange(nrows, dtype=np.int64) + rank * (nrows - plane_rows)


def porous_slab_gids(nxy, nz, rank):
    """porousMixed (HVOL p + HDIV u) on z-slabs of nz[rank] layers of nxy x nxy hexes: global ids (numbering of the
    single-domain mesh_multi: cells, x-faces, y-faces, z-faces) of a slab's dofs, in the slab's local order.  The shared
    dofs are the z-faces between two slabs: not a leading / trailing run of rows."""
    nzt, k0, nzl = int(sum(nz)), int(sum(nz[:rank])), int(nz[rank])
    ne_g = nxy * nxy * nzt
    cells = np.arange(nxy * nxy * nzl, dtype=np.int64) + nxy * nxy * k0
    fx = ne_g + np.arange((nxy + 1) * nxy * nzl, dtype=np.int64) + (nxy + 1) * nxy * k0
    fy = ne_g + (nxy + 1) * nxy * nzt + np.arange(nxy * (nxy + 1) * nzl, dtype=np.int64) + nxy * (nxy + 1) * k0
    fz = ne_g + 2 * (nxy + 1) * nxy * nzt + np.arange(nxy * nxy * (nzl + 1), dtype=np.int64) + nxy * nxy * k0
    return np.concatenate([cells, fx, fy, fz])


def hdg_trace_lids(ncx, ncy, nvar=3, per_edge=2):
    """Trace LIDs [E][4 * nvar * per_edge] of an ncx x ncy quad mesh with HFACE traces: edges numbered vertical ones first
    (row by row), then the horizontal ones; local edge order left, bottom, right, top (Intrepid2_HFACE_QUAD_In_FEM);
    dof = ((edge * nvar + var) * per_edge + f), element-local slot ((var * 4 + k) * per_edge + f).  Returns (lids, nrows)."""
    nvert = (ncx + 1) * ncy
    ii, jj = np.meshgrid(np.arange(ncx), np.arange(ncy), indexing="xy")
    ii, jj = ii.ravel(), jj.ravel()
    edges = np.stack([jj * (ncx + 1) + ii, nvert + jj * ncx + ii, jj * (ncx + 1) + ii + 1, nvert + (jj + 1) * ncx + ii], axis=1)
    lids = np.zeros((ncx * ncy, 4 * nvar * per_edge), np.int32)
    for v in range(nvar):
        for k in range(4):
            for f in range(per_edge):
                lids[:, (v * 4 + k) * per_edge + f] = (edges[:, k] * nvar + v) * per_edge + f
    return lids, (nvert + ncx * (ncy + 1)) * nvar * per_edge


def hdg_strip_gids(ncx, ncy, rank, nvar=3, per_edge=2):
    """Strips of ncy[rank] element rows stacked in y (config 5: 8 strips of 32 rows): global id of every local trace
    row of strip `rank`, numbering of hdg_trace_lids on the whole ncx x sum(ncy) mesh.  Neighbouring strips share the
    horizontal edges between them (the HDG trace rows of SubGridDtN_Solver::updateFlux, subgridDtN_solver.cpp:1542-1616)."""
    nyt, j0, nyl = int(sum(ncy)), int(sum(ncy[:rank])), int(ncy[rank])
    nvert_l, nvert_g = (ncx + 1) * nyl, (ncx + 1) * nyt
    e_loc = np.arange(nvert_l + ncx * (nyl + 1), dtype=np.int64)
    vert = e_loc < nvert_l
    ge = np.where(vert, e_loc + j0 * (ncx + 1), nvert_g + (e_loc - nvert_l) + j0 * ncx)
    d = np.arange(nvar * per_edge, dtype=np.int64)
    return (ge[:, None] * (nvar * per_edge) + d[None, :]).ravel()


class SlabExchange(SharedRowExport):
    """z-slab decomposition with identical slabs (bench.py configs 2 and 4): SharedRowExport with slab_gids."""

    def __init__(self, rowptr, colind, plane_rows, nrows, rank, world, device):
        super().__init__(slab_gids(nrows, plane_rows, rank), rowptr, colind, rank, world, device)
        self.P = int(plane_rows)
