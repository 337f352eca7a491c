// TEST INFRASTRUCTURE ONLY: minimal stand-in for <hip/hip_runtime.h> so the *host* logic of libcmdr_hip
// (plan tables, CR orchestration, C ABI) can be compiled with g++ and exercised on a GPU-less box together with
// tests/host_emul/emul_launch.cpp (single-thread loops over the kernel bodies).  Never part of the product build.
#pragma once
#include <cstddef>
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
static const hipError_t hipSuccess = 0;
typedef void* hipStream_t;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };

inline const char* hipGetErrorString(hipError_t) { return "host-emulation error"; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = *t = (size_t)1 << 40; return 0; }
inline hipError_t hipFree(void* p) { std::free(p); return 0; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return 0; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return 0; }
constexpr unsigned hipHostRegisterDefault = 0;
inline hipError_t hipHostRegister(void*, size_t, unsigned) { return 0; }
inline hipError_t hipHostUnregister(void*) { return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return 0; }
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
typedef void* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return 0; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
constexpr unsigned hipEventDisableTiming = 2;
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
