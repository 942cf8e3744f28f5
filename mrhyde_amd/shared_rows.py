"""Export(ADD) of shared-DOF rows between element-block shards, over torch.distributed (RCCL on GPU).

Reference semantics (src/interfaces/linearAlgebraInterface.hpp:296-337): every rank assembles into
its OVERLAPPED residual / CRS matrix (owned + ghost rows); `doExport(..., Tpetra::ADD)` then sums the
ghost-row contributions into the owning rank's rows.  Here the element block is cut into z-slabs of a
structured mesh (contiguous runs of worksets in the reference's sequential order); the only shared rows
are the dofs on the plane between two slabs, owned by the LOWER slab.  Because the plane's rows are
contiguous at the start of the upper slab's CRS arrays, the message is a plain slice (values only, no
indices on the wire: both sides derive the same sparsity from the mesh).  The owner adds the entries
whose columns lie on the shared plane into its local CRS and keeps the rest -- couplings to the upper
slab's interior dofs, i.e. the off-rank columns of its owned rows -- in `remote_vals`.

Point-to-point send/recv between slab neighbours (each pair has a direct xGMI link) instead of a ring
collective.
"""
import numpy as np


class SlabExchange:
    """Setup + per-assembly exchange for one rank of a z-slab decomposition with identical slabs."""

    def __init__(self, rowptr, colind, plane_rows, nrows, rank, world, device):
        import torch
        self.rank, self.world = rank, world
        self.P = int(plane_rows)
        self.nsend = int(rowptr[self.P])          # entries of my bottom-plane rows [0, P)
        self.has_lower = rank > 0                 # I send my bottom plane to rank-1 (the owner)
        self.has_upper = rank < world - 1         # I own my top plane and receive from rank+1
        self.top0 = nrows - self.P                # first row of my top plane
        if self.has_upper:
            # the upper neighbour's bottom rows have the structure of MY bottom rows (identical slabs)
            rows = np.repeat(np.arange(self.P, dtype=np.int64), np.diff(rowptr[:self.P + 1]))
            cols = colind[:self.nsend].astype(np.int64)
            inplane = cols < self.P
            trow = rows[inplane] + self.top0
            tcol = cols[inplane] + self.top0
            # position of (trow, tcol) in my CRS: entries are sorted by (row, col), so one searchsorted
            seg = int(rowptr[self.top0])
            top_rows = np.repeat(np.arange(self.top0, nrows, dtype=np.int64), np.diff(rowptr[self.top0:nrows + 1]))
            keys = top_rows * nrows + colind[seg:].astype(np.int64)
            pos = seg + np.searchsorted(keys, trow * nrows + tcol)
            assert np.array_equal(colind[pos], tcol)
            self.src_idx = torch.tensor(np.flatnonzero(inplane), device=device)
            self.dst_pos = torch.tensor(pos, device=device)
            self.remote_idx = torch.tensor(np.flatnonzero(~inplane), device=device)
            self.recv_vals = torch.zeros(self.nsend, dtype=torch.float64, device=device)
            self.recv_res = torch.zeros(self.P, dtype=torch.float64, device=device)
            self.remote_vals = torch.zeros(int((~inplane).sum()), dtype=torch.float64, device=device)

    def _export_add_staged(self, res, vals):
        """Same exchange with host staging (gloo cannot move device buffers); used only to rehearse N>1 on one GPU."""
        import torch.distributed as dist
        ops, keep = [], []
        if self.has_lower:
            a, b = vals[:self.nsend].cpu(), res[:self.P].cpu()
            keep += [a, b]
            ops += [dist.P2POp(dist.isend, a, self.rank - 1), dist.P2POp(dist.isend, b, self.rank - 1)]
        if self.has_upper:
            rv, rr = self.recv_vals.cpu(), self.recv_res.cpu()
            ops += [dist.P2POp(dist.irecv, rv, self.rank + 1), dist.P2POp(dist.irecv, rr, self.rank + 1)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if self.has_upper:
            self.recv_vals.copy_(rv)
            self.recv_res.copy_(rr)
            vals.index_add_(0, self.dst_pos, self.recv_vals[self.src_idx])
            res[self.top0:] += self.recv_res
            self.remote_vals.copy_(self.recv_vals[self.remote_idx])

    def bytes_on_wire(self):
        return (self.nsend + self.P) * 8

    def export_add(self, res, vals, fixed_top=None):
        """Sum ghost-row contributions into the owner.  res/vals: this rank's overlapped arrays."""
        import torch.distributed as dist
        if self.world == 1:
            return
        if dist.get_backend() == "gloo" and vals.is_cuda:
            return self._export_add_staged(res, vals)  # rehearsal on one GPU: gloo moves host buffers
        ops = []
        if self.has_lower:
            ops.append(dist.P2POp(dist.isend, vals[:self.nsend], self.rank - 1))
            ops.append(dist.P2POp(dist.isend, res[:self.P], self.rank - 1))
        if self.has_upper:
            ops.append(dist.P2POp(dist.irecv, self.recv_vals, self.rank + 1))
            ops.append(dist.P2POp(dist.irecv, self.recv_res, self.rank + 1))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if self.has_upper:
            vals.index_add_(0, self.dst_pos, self.recv_vals[self.src_idx])
            res[self.top0:] += self.recv_res
            self.remote_vals.copy_(self.recv_vals[self.remote_idx])
