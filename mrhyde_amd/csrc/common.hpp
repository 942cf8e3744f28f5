// common.hpp -- error plumbing and device-buffer RAII for the mrhyde_amd host layer.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
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

inline int ipow(int b, int e) { int r = 1; while (e-- > 0) r *= b; return r; }

}  // namespace mha
