// export_plan.hpp -- Export(ADD) of shared-DOF rows between element-block shards (one process per GPU).
//
// Reference semantics (src/interfaces/linearAlgebraInterface.hpp:296-337): every rank assembles into its OVERLAPPED
// residual / CRS matrix (owned + ghost rows); `doExport(..., Tpetra::ADD)` then sums the ghost-row contributions into the
// owning rank's rows.  Here the shared rows are given as explicit lists agreed at setup (any partition: HGRAD planes
// between slabs, HDIV face dofs, HDG trace rows, unequal slabs):
//   per neighbour: which entries of my value array / residual go on the wire (I am not the owner of those rows), and
//   for what arrives from it, where each entry is added (target index into my arrays, -1 = an off-rank column of one of
//   my owned rows: kept in the receive buffer for the caller).
// No indices travel: both sides derive the same order from the global ids.  pack / unpack are HIP kernels; the
// transport is RCCL point-to-point (ncclSend / ncclRecv in one group: neighbouring slabs have a direct xGMI link) --
// either inside mha_export_add (this library's own communicator) or by the caller between mha_export_pack and
// mha_export_unpack_add (e.g. torch.distributed, whose "nccl" backend is RCCL).
#pragma once
#include <cstdint>
#include <vector>

#include "common.hpp"

namespace mha {

class Comm;  // RCCL communicator (export_plan.cpp), loaded on first use

class ExportPlan {
 public:
  // CSR-like per-neighbour lists (host): *_ptr have num_neighbors + 1 entries
  // nnz / nrows: sizes of the value array and the residual the lists index (every index is checked against them, and
  // the receive targets of one neighbour must be distinct: unpackAdd adds with plain read-modify-write)
  ExportPlan(int num_neighbors, const int32_t *ranks, const int64_t *send_val_ptr, const int32_t *send_val_index,
             const int64_t *send_row_ptr, const int32_t *send_row_index, const int64_t *recv_val_ptr,
             const int32_t *recv_val_target, const int64_t *recv_row_ptr, const int32_t *recv_row_target, int64_t nnz,
             int64_t nrows);
  int numNeighbors() const { return static_cast<int>(ranks_.size()); }
  int rank(int k) const { return ranks_[k]; }
  // my non-owned rows -> send buffers.  vals == nullptr: a residual-only exchange (the value segments are skipped here,
  // in unpackAdd and on the wire of exportAdd); res == nullptr likewise
  void pack(const double *vals, const double *res, hipStream_t stream) const;
  // receive buffers -> my owned rows (+=), neighbours in the order given (deterministic)
  void unpackAdd(double *vals, double *res, hipStream_t stream) const;
  // buffer of neighbour k: [values | residual entries]
  double *sendBuffer(int k, int64_t *count) const;
  double *recvBuffer(int k, int64_t *count) const;
  int64_t bytesOnWire() const;
  int64_t sendValues(int k) const { return sv_ptr_[k + 1] - sv_ptr_[k]; }
  int64_t recvValues(int k) const { return rv_ptr_[k + 1] - rv_ptr_[k]; }
  // pack + ncclSend / ncclRecv with every neighbour in one group + unpackAdd
  void exportAdd(Comm &comm, double *vals, double *res, hipStream_t stream) const;

 private:
  std::vector<int32_t> ranks_;
  std::vector<int64_t> sv_ptr_, sr_ptr_, rv_ptr_, rr_ptr_;
  DeviceBuffer<int32_t> sv_idx_, sr_idx_, rv_tgt_, rr_tgt_;
  std::vector<DeviceBuffer<double>> send_, recv_;
};

// RCCL communicator of this library (librccl.so is loaded on first use: a single-GPU process never needs it)
class Comm {
 public:
  static void uniqueId(char id[128]);
  Comm(int nranks, int rank, const char id[128]);
  ~Comm();
  void sendRecv(const ExportPlan &plan, hipStream_t stream, bool values = true, bool residual = true);
  int rank() const { return rank_; }
  int size() const { return nranks_; }

 private:
  void *comm_ = nullptr;
  int rank_ = 0, nranks_ = 0;
};

}  // namespace mha
