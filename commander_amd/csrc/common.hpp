// Common host-side helpers for libcmdr_hip: error propagation to the C ABI, device buffers, host threads.
#pragma once
#include <hip/hip_runtime.h>

#include "common_host.hpp"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace cmdr {

#define CMDR_HIP_CHECK(expr)                                                                          \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            throw ::cmdr::Error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" +     \
                                __FILE__ + ":" + std::to_string(__LINE__) + ")");                     \
    } while (0)

// RAII device buffer (hipMalloc); move-only.
template <typename T>
class DevBuf {
  public:
    DevBuf() = default;
    explicit DevBuf(size_t n) { alloc(n); }
    ~DevBuf() { release(); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; }
        return *this;
    }
    void alloc(size_t n) {
        release();
        n_ = n;
        if (n) CMDR_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p_), n * sizeof(T)));
    }
    void ensure(size_t n) { if (n > n_) alloc(n); }
    void release() {
        if (p_) (void)hipFree(p_);
        p_ = nullptr; n_ = 0;
    }
    void upload(const T* h, size_t n, hipStream_t s = nullptr) {
        ensure(n);
        if (n) CMDR_HIP_CHECK(hipMemcpyAsync(p_, h, n * sizeof(T), hipMemcpyHostToDevice, s));
        if (n) CMDR_HIP_CHECK(hipStreamSynchronize(s));
    }
    void upload(const std::vector<T>& h, hipStream_t s = nullptr) { upload(h.data(), h.size(), s); }
    void zero(hipStream_t s = nullptr) { if (n_) CMDR_HIP_CHECK(hipMemsetAsync(p_, 0, n_ * sizeof(T), s)); }
    T* get() const { return p_; }
    size_t size() const { return n_; }

  private:
    T* p_ = nullptr;
    size_t n_ = 0;
};

}  // namespace cmdr
