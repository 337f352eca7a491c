// HIP-free host helpers (shared by the product library and the host-emulation test library).
#pragma once
#include <algorithm>
#include <cstdint>
#include <exception>
#include <functional>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace cmdr {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define CMDR_REQUIRE(cond, msg)                                              \
    do {                                                                     \
        if (!(cond)) throw ::cmdr::Error(std::string(msg) + " [" #cond "]"); \
    } while (0)

void set_last_error(const char* msg);   // c_api.cpp: the message cmdr_last_error() returns on this thread

// Strided parallel for on host threads (plan construction only; never on the per-iteration path).
inline void host_parallel_for(int n, const std::function<void(int)>& fn, int nthreads = 0) {
    if (nthreads <= 0) nthreads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
    nthreads = std::min(nthreads, std::max(1, n));
    if (nthreads == 1) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
    }
    std::vector<std::thread> th;
    std::exception_ptr err = nullptr;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t]() {
            try {
                for (int i = t; i < n; i += nthreads) fn(i);
            } catch (...) {
                err = std::current_exception();
            }
        });
    for (auto& x : th) x.join();
    if (err) std::rethrow_exception(err);
}

}  // namespace cmdr
