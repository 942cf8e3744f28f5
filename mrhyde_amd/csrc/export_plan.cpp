// export_plan.cpp -- see export_plan.hpp.
#include "export_plan.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <mutex>

#include "kernels/launch.hpp"

namespace mha {

ExportPlan::ExportPlan(int nnb, const int32_t *ranks, const int64_t *sv_ptr, const int32_t *sv_idx, const int64_t *sr_ptr,
                       const int32_t *sr_idx, const int64_t *rv_ptr, const int32_t *rv_tgt, const int64_t *rr_ptr,
                       const int32_t *rr_tgt, int64_t nnz, int64_t nrows) {
  MHA_REQUIRE(nnz >= 0 && nrows >= 0, MHA_ERR_INVALID, "export plan: negative array size");
  MHA_REQUIRE(nnb >= 0, MHA_ERR_INVALID, "export plan: negative neighbour count");
  if (nnb > 0)
    MHA_REQUIRE(ranks && sv_ptr && sr_ptr && rv_ptr && rr_ptr, MHA_ERR_INVALID, "export plan: null list");
  ranks_.assign(ranks, ranks + nnb);
  auto take = [&](const int64_t *p, std::vector<int64_t> &dst) {
    dst.assign(1, 0);
    if (nnb > 0) dst.assign(p, p + nnb + 1);
    MHA_REQUIRE(dst[0] == 0, MHA_ERR_INVALID, "export plan: list offsets must start at 0");
    for (int k = 0; k < nnb; ++k) MHA_REQUIRE(dst[k + 1] >= dst[k], MHA_ERR_INVALID, "export plan: list offsets must not decrease");
  };
  take(sv_ptr, sv_ptr_);
  take(sr_ptr, sr_ptr_);
  take(rv_ptr, rv_ptr_);
  take(rr_ptr, rr_ptr_);
  MHA_REQUIRE((sv_ptr_.back() == 0 || sv_idx) && (sr_ptr_.back() == 0 || sr_idx) && (rv_ptr_.back() == 0 || rv_tgt) &&
                  (rr_ptr_.back() == 0 || rr_tgt), MHA_ERR_INVALID, "export plan: null index list");
  for (int64_t i = 0; i < sv_ptr_.back(); ++i)
    MHA_REQUIRE(sv_idx[i] >= 0 && sv_idx[i] < nnz, MHA_ERR_INVALID, "export plan: send index " << sv_idx[i] << " outside the value array (" << nnz << " entries)");
  for (int64_t i = 0; i < sr_ptr_.back(); ++i)
    MHA_REQUIRE(sr_idx[i] >= 0 && sr_idx[i] < nrows, MHA_ERR_INVALID, "export plan: send row " << sr_idx[i] << " outside the residual (" << nrows << " rows)");
  for (int64_t i = 0; i < rv_ptr_.back(); ++i)
    MHA_REQUIRE(rv_tgt[i] >= -1 && rv_tgt[i] < nnz, MHA_ERR_INVALID, "export plan: receive target " << rv_tgt[i] << " outside the value array (" << nnz << " entries)");
  for (int64_t i = 0; i < rr_ptr_.back(); ++i)
    MHA_REQUIRE(rr_tgt[i] >= 0 && rr_tgt[i] < nrows, MHA_ERR_INVALID, "export plan: receive row " << rr_tgt[i] << " outside the residual (" << nrows << " rows)");
  {  // the targets of ONE neighbour are added by one kernel launch with plain read-modify-write: they must be distinct
    std::vector<int32_t> t;
    for (int k = 0; k < nnb; ++k) {
      t.assign(rv_tgt + rv_ptr_[k], rv_tgt + rv_ptr_[k + 1]);
      t.erase(std::remove(t.begin(), t.end(), -1), t.end());
      std::sort(t.begin(), t.end());
      MHA_REQUIRE(std::adjacent_find(t.begin(), t.end()) == t.end(), MHA_ERR_INVALID, "export plan: neighbour " << ranks[k] << " lists a value target twice");
      t.assign(rr_tgt + rr_ptr_[k], rr_tgt + rr_ptr_[k + 1]);
      std::sort(t.begin(), t.end());
      MHA_REQUIRE(std::adjacent_find(t.begin(), t.end()) == t.end(), MHA_ERR_INVALID, "export plan: neighbour " << ranks[k] << " lists a residual row twice");
    }
  }
  sv_idx_.upload(sv_idx, static_cast<size_t>(sv_ptr_.back()));
  sr_idx_.upload(sr_idx, static_cast<size_t>(sr_ptr_.back()));
  rv_tgt_.upload(rv_tgt, static_cast<size_t>(rv_ptr_.back()));
  rr_tgt_.upload(rr_tgt, static_cast<size_t>(rr_ptr_.back()));
  send_.resize(nnb);
  recv_.resize(nnb);
  for (int k = 0; k < nnb; ++k) {
    send_[k].resize(static_cast<size_t>(sv_ptr_[k + 1] - sv_ptr_[k] + sr_ptr_[k + 1] - sr_ptr_[k]));
    recv_[k].resize(static_cast<size_t>(rv_ptr_[k + 1] - rv_ptr_[k] + rr_ptr_[k + 1] - rr_ptr_[k]));
  }
}

double *ExportPlan::sendBuffer(int k, int64_t *count) const {
  MHA_REQUIRE(k >= 0 && k < numNeighbors(), MHA_ERR_INVALID, "export plan: neighbour index out of range");
  if (count) *count = static_cast<int64_t>(send_[k].size());
  return send_[k].data();
}
double *ExportPlan::recvBuffer(int k, int64_t *count) const {
  MHA_REQUIRE(k >= 0 && k < numNeighbors(), MHA_ERR_INVALID, "export plan: neighbour index out of range");
  if (count) *count = static_cast<int64_t>(recv_[k].size());
  return recv_[k].data();
}
int64_t ExportPlan::bytesOnWire() const {
  int64_t n = 0;
  for (const auto &b : send_) n += static_cast<int64_t>(b.size());
  return 8 * n;
}

void ExportPlan::pack(const double *vals, const double *res, hipStream_t stream) const {
  for (int k = 0; k < numNeighbors(); ++k) {
    const int64_t nv = sv_ptr_[k + 1] - sv_ptr_[k], nr = sr_ptr_[k + 1] - sr_ptr_[k];
    if (nv > 0 && vals) launch_export_pack(vals, sv_idx_.data() + sv_ptr_[k], nv, send_[k].data(), stream);
    if (nr > 0 && res) launch_export_pack(res, sr_idx_.data() + sr_ptr_[k], nr, send_[k].data() + nv, stream);
  }
}

void ExportPlan::unpackAdd(double *vals, double *res, hipStream_t stream) const {
  for (int k = 0; k < numNeighbors(); ++k) {  // one neighbour after the other: two may add into the same row
    const int64_t nv = rv_ptr_[k + 1] - rv_ptr_[k], nr = rr_ptr_[k + 1] - rr_ptr_[k];
    if (nv > 0 && vals) launch_export_unpack_add(recv_[k].data(), rv_tgt_.data() + rv_ptr_[k], nv, vals, stream);
    if (nr > 0 && res) launch_export_unpack_add(recv_[k].data() + nv, rr_tgt_.data() + rr_ptr_[k], nr, res, stream);
  }
}

void ExportPlan::exportAdd(Comm &comm, double *vals, double *res, hipStream_t stream) const {
  MHA_REQUIRE(vals || res, MHA_ERR_INVALID, "export plan: nothing to exchange (no value array, no residual)");
  pack(vals, res, stream);
  comm.sendRecv(*this, stream, vals != nullptr, res != nullptr);
  unpackAdd(vals, res, stream);
}

// ---- RCCL, bound at run time ----------------------------------------------------------------------------------------
namespace {
struct Rccl {
  void *lib = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(void **, int, char[128], int) = nullptr;  // ncclUniqueId is passed by value: 128 bytes
  int (*CommDestroy)(void *) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
struct UniqueId { char b[128]; };

Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
    r.GetUniqueId = reinterpret_cast<int (*)(void *)>(dlsym(r.lib, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<int (*)(void **, int, char[128], int)>(dlsym(r.lib, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<int (*)(void *)>(dlsym(r.lib, "ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<int (*)()>(dlsym(r.lib, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<int (*)()>(dlsym(r.lib, "ncclGroupEnd"));
    r.Send = reinterpret_cast<int (*)(const void *, size_t, int, int, void *, hipStream_t)>(dlsym(r.lib, "ncclSend"));
    r.Recv = reinterpret_cast<int (*)(void *, size_t, int, int, void *, hipStream_t)>(dlsym(r.lib, "ncclRecv"));
    r.GetErrorString = reinterpret_cast<const char *(*)(int)>(dlsym(r.lib, "ncclGetErrorString"));
  });
  MHA_REQUIRE(r.lib && r.GetUniqueId && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv, MHA_ERR_DEVICE,
              "librccl.so could not be loaded: " << (dlerror() ? dlerror() : "symbol missing"));
  return r;
}
void check(int rc, const char *what) {
  if (rc == 0) return;
  Rccl &r = rccl();
  throw Error(MHA_ERR_DEVICE, std::string(what) + " failed: " + (r.GetErrorString ? r.GetErrorString(rc) : "rccl error"));
}
constexpr int kNcclFloat64 = 8;  // ncclDouble
}  // namespace

void Comm::uniqueId(char id[128]) {
  UniqueId u;
  std::memset(&u, 0, sizeof u);
  check(rccl().GetUniqueId(&u), "ncclGetUniqueId");
  std::memcpy(id, u.b, 128);
}

Comm::Comm(int nranks, int rank, const char id[128]) : rank_(rank), nranks_(nranks) {
  MHA_REQUIRE(nranks > 0 && rank >= 0 && rank < nranks && id, MHA_ERR_INVALID, "communicator: bad rank / size");
  // ncclCommInitRank(ncclComm_t *, int, ncclUniqueId (by value: a 128-byte struct, passed in memory), int)
  using InitFn = int (*)(void **, int, UniqueId, int);
  InitFn init = reinterpret_cast<InitFn>(dlsym(rccl().lib, "ncclCommInitRank"));
  MHA_REQUIRE(init != nullptr, MHA_ERR_DEVICE, "ncclCommInitRank not found in librccl.so");
  UniqueId u;
  std::memcpy(u.b, id, 128);
  check(init(&comm_, nranks, u, rank), "ncclCommInitRank");
}

Comm::~Comm() {
  if (comm_) (void)rccl().CommDestroy(comm_);
}

void Comm::sendRecv(const ExportPlan &plan, hipStream_t stream, bool values, bool residual) {
  Rccl &r = rccl();
  check(r.GroupStart(), "ncclGroupStart");
  for (int k = 0; k < plan.numNeighbors(); ++k) {
    int64_t ns = 0, nr = 0;
    double *s = plan.sendBuffer(k, &ns), *v = plan.recvBuffer(k, &nr);
    // the buffers are [values | residual entries]: a residual-only (or values-only) exchange moves its half
    const int64_t sv = plan.sendValues(k), rv = plan.recvValues(k);
    const int64_t s0 = values ? 0 : sv, s1 = residual ? ns : sv, r0 = values ? 0 : rv, r1 = residual ? nr : rv;
    if (s1 > s0) check(r.Send(s + s0, static_cast<size_t>(s1 - s0), kNcclFloat64, plan.rank(k), comm_, stream), "ncclSend");
    if (r1 > r0) check(r.Recv(v + r0, static_cast<size_t>(r1 - r0), kNcclFloat64, plan.rank(k), comm_, stream), "ncclRecv");
  }
  check(r.GroupEnd(), "ncclGroupEnd");
}

}  // namespace mha
