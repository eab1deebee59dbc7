// Shared helpers for the SAT HIP library (gfx950 / MI355X only).
#pragma once
#include <cstdio>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

namespace sat {

enum : int { SAT_OK = 0, SAT_EINVAL = 1, SAT_EHIP = 2, SAT_EUNSUPPORTED = 3 };

char* last_error_buf();          // thread-local, 512 bytes
int fail(int code, const char* fmt, ...);

#define SAT_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) return ::sat::fail(::sat::SAT_EHIP, "%s:%d %s -> %s",      \
                                                 __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
    } while (0)

#define SAT_REQUIRE(cond, ...)                                            \
    do {                                                                  \
        if (!(cond)) return ::sat::fail(::sat::SAT_EINVAL, __VA_ARGS__);  \
    } while (0)

#define SAT_TRY(expr)                \
    do {                             \
        int _rc = (expr);            \
        if (_rc != 0) return _rc;    \
    } while (0)

int& trace_launches();      // api.hip: sat_debug_trace_launches
inline int launch_ok(const char* what) {
    if (trace_launches()) {        // dev: wait for the kernel and name it, so that a device fault points at the launch after the last line
        hipError_t s = hipDeviceSynchronize();
        fprintf(stderr, "[sat] %s: %s\n", what, hipGetErrorString(s));
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SAT_EHIP, "launch %s -> %s", what, hipGetErrorString(e));
    return SAT_OK;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// devmem.hip: clears / device-to-device copies as kernel launches (the train path enqueues kernels only: capturable, see the file header)
int dev_fill_bytes(hipStream_t st, void* ptr, int byte, size_t nbytes);
int dev_copy_bytes(hipStream_t st, void* dst, const void* src, size_t nbytes);

// ---- device helpers ----------------------------------------------------
__device__ __forceinline__ float fast_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }
// tanh through one exp: |err| ~1e-7 absolute, saturates correctly for large |x|
__device__ __forceinline__ float fast_tanh(float x) {
    float e = __expf(2.0f * x);
    return 1.0f - 2.0f / (e + 1.0f);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace sat
