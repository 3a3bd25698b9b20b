// hip_stubs.h — host-only stand-ins for the handful of HIP runtime entry points that the host-side sources reference
// through context.h, so that imread.cpp / homography.cpp / host_pool.h can be compiled with g++ and run under
// AddressSanitizer / UndefinedBehaviorSanitizer / ThreadSanitizer on a machine without a GPU (tests/test_cpu_sanitize.py).
// CPU build only: GPU ASan is not available on the pool. Nothing here is part of the product.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

extern "C" {
hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipGetLastError(void) { return hipSuccess; }
}
