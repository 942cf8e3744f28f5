// common.hpp -- error plumbing and device-buffer RAII for the mrhyde_amd host layer.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdlib>
#include <cstdint>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mrhyde_amd.h"

namespace mha {

// Internal exception; translated to a status code + mha_last_error() at the C ABI.
struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define MHA_REQUIRE(cond, code, msg)                                                   \
  do {                                                                                 \
    if (!(cond)) {                                                                     \
      std::ostringstream os_;                                                          \
      os_ << msg;                                                                      \
      throw ::mha::Error((code), os_.str());                                           \
    }                                                                                  \
  } while (0)

#define MHA_HIP(call)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      std::ostringstream os_;                                                          \
      os_ << #call << " failed: " << hipGetErrorString(e_) << " (" << __FILE__ << ":"  \
          << __LINE__ << ")";                                                          \
      throw ::mha::Error(MHA_ERR_DEVICE, os_.str());                                   \
    }                                                                                  \
  } while (0)

// Owning device allocation (hipMalloc); move-only.
template <class T>
class DeviceBuffer {
 public:
  DeviceBuffer() = default;
  explicit DeviceBuffer(size_t n) { resize(n); }
  ~DeviceBuffer() { release(); }
  DeviceBuffer(const DeviceBuffer &) = delete;
  DeviceBuffer &operator=(const DeviceBuffer &) = delete;
  DeviceBuffer(DeviceBuffer &&o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
  DeviceBuffer &operator=(DeviceBuffer &&o) noexcept {
    if (this != &o) { release(); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; }
    return *this;
  }
  void resize(size_t n) {
    if (n == n_) return;
    release();
    if (n) MHA_HIP(hipMalloc(reinterpret_cast<void **>(&p_), n * sizeof(T)));
    n_ = n;
  }
  void upload(const T *host, size_t n) {
    resize(n);
    if (n) MHA_HIP(hipMemcpy(p_, host, n * sizeof(T), hipMemcpyHostToDevice));
  }
  void upload(const std::vector<T> &v) { upload(v.data(), v.size()); }
  void download(T *host) const {
    if (n_) MHA_HIP(hipMemcpy(host, p_, n_ * sizeof(T), hipMemcpyDeviceToHost));
  }
  T *data() const { return p_; }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }

 private:
  void release() {
    if (p_) (void)hipFree(p_);
    p_ = nullptr;
    n_ = 0;
  }
  T *p_ = nullptr;
  size_t n_ = 0;
};

// Makes `device` the current HIP device for the lifetime of the object and restores the caller's device afterwards:
// every entry point of the C ABI that works on a context runs under one (a context is bound to the device it was
// created on, mha_block_desc.device; the caller's current device is none of our business).
class DeviceGuard {
 public:
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
    if (device >= 0 && device != prev_) { MHA_HIP(hipSetDevice(device)); changed_ = true; }
  }
  ~DeviceGuard() {
    if (changed_ && prev_ >= 0) (void)hipSetDevice(prev_);
  }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;

 private:
  int prev_ = -1;
  bool changed_ = false;
};

// CU count of the CURRENT device (cached per ordinal: launchers size their persistent grids with it).
inline int current_device_num_cus() {
  static int cache[64] = {0};
  int dev = 0;
  MHA_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) { int n = 0; MHA_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev)); return n; }
  if (!cache[dev]) MHA_HIP(hipDeviceGetAttribute(&cache[dev], hipDeviceAttributeMultiprocessorCount, dev));
  return cache[dev];
}

// A kernel whose register spills need a large private (scratch) segment makes the runtime carve per-wave x all-wave-slots
// bytes out of device memory at launch; when that fails the HSA runtime abort()s the process -- nothing the C ABI can
// catch (the round-1 `MHA_ENGINE_MINW` build variants of the point engine spilled up to 359 registers = 1.4 KB per lane,
// 0.75 GB for the chip, and their porousMixed run died with SIGABRT).  Launchers of kernels that CAN spill ask here
// first: more than the limit (default 1.25 KB per lane; the deck-string instantiations, which carry the interpreter's
// stack and spill around it, need 0.15-0.55 KB since the interpreter is inlined -- 1.0-1.3 KB when it was a call --, env
// MHA_MAX_SCRATCH_BYTES) is refused with MHA_ERR_DEVICE.
template <class Kernel>
inline void require_modest_scratch(Kernel kern, const char *what) {
  hipFuncAttributes attr;
  MHA_HIP(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(kern)));
  size_t limit = 1280;
  if (const char *e = std::getenv("MHA_MAX_SCRATCH_BYTES")) limit = static_cast<size_t>(std::atoll(e));
  MHA_REQUIRE(attr.localSizeBytes <= limit, MHA_ERR_DEVICE,
              what << ": the kernel needs " << attr.localSizeBytes << " B of scratch per lane (limit " << limit
                   << "): refusing a launch the runtime may not be able to back");
}

inline int ipow(int b, int e) { int r = 1; while (e-- > 0) r *= b; return r; }

}  // namespace mha
