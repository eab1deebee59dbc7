// Clears and device-to-device copies as KERNELS.  The train path enqueues nothing but kernel launches (no hipMemsetAsync / hipMemcpyAsync
// nodes, no host-memory reads at execution time), so that a captured stream (hipGraph) is a plain chain of kernel nodes: memset / memcpy
// nodes were seen to run out of order with the kernels around them inside a replayed graph (hipGraph replay).
#include "common.h"

namespace sat {

__global__ void dev_fill16_kernel(uint4* __restrict__ p, long n16, unsigned v) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) p[i] = make_uint4(v, v, v, v);
}
__global__ void dev_fill1_kernel(unsigned char* __restrict__ p, long n, unsigned char v) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void dev_copy16_kernel(uint4* __restrict__ d, const uint4* __restrict__ s, long n16) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) d[i] = s[i];
}
__global__ void dev_copy4_kernel(unsigned* __restrict__ d, const unsigned* __restrict__ s, long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) d[i] = s[i];
}
__global__ void dev_copy1_kernel(unsigned char* __restrict__ d, const unsigned char* __restrict__ s, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = s[i];
}

// `byte` replicated over nbytes (what hipMemsetAsync does); 16 bytes per lane over the aligned body, bytes at the ragged ends
int dev_fill_bytes(hipStream_t st, void* ptr, int byte, size_t nbytes) {
    if (nbytes == 0) return SAT_OK;
    unsigned char* p = reinterpret_cast<unsigned char*>(ptr);
    const unsigned char b = (unsigned char)byte;
    const unsigned w = 0x01010101u * b;
    const size_t head = (16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15;
    if (head >= nbytes) {
        hipLaunchKernelGGL(dev_fill1_kernel, dim3(cdiv((long)nbytes, 256)), dim3(256), 0, st, p, (long)nbytes, b);
        return launch_ok("dev_fill (bytes)");
    }
    const size_t body = (nbytes - head) / 16, tail = nbytes - head - body * 16;
    if (head) hipLaunchKernelGGL(dev_fill1_kernel, dim3(1), dim3(64), 0, st, p, (long)head, b);
    if (body) hipLaunchKernelGGL(dev_fill16_kernel, dim3(cdiv((long)body, 256)), dim3(256), 0, st, reinterpret_cast<uint4*>(p + head), (long)body, w);
    if (tail) hipLaunchKernelGGL(dev_fill1_kernel, dim3(1), dim3(64), 0, st, p + head + body * 16, (long)tail, b);
    return launch_ok("dev_fill");
}

// non-overlapping device-to-device copy; 16 bytes per lane when both ends are 16-byte aligned
int dev_copy_bytes(hipStream_t st, void* dst, const void* src, size_t nbytes) {
    if (nbytes == 0 || dst == src) return SAT_OK;
    unsigned char* d = reinterpret_cast<unsigned char*>(dst);
    const unsigned char* s = reinterpret_cast<const unsigned char*>(src);
    if (((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(s)) & 15) == 0) {
        const size_t body = nbytes / 16, tail = nbytes - body * 16;
        if (body) hipLaunchKernelGGL(dev_copy16_kernel, dim3(cdiv((long)body, 256)), dim3(256), 0, st, reinterpret_cast<uint4*>(d), reinterpret_cast<const uint4*>(s), (long)body);
        if (tail) hipLaunchKernelGGL(dev_copy1_kernel, dim3(1), dim3(64), 0, st, d + body * 16, s + body * 16, (long)tail);
    } else if (((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(s) | nbytes) & 3) == 0) {
        // 4-byte aligned (every buffer of the library is): one word per lane
        hipLaunchKernelGGL(dev_copy4_kernel, dim3(cdiv((long)(nbytes / 4), 256)), dim3(256), 0, st, reinterpret_cast<unsigned*>(d), reinterpret_cast<const unsigned*>(s), (long)(nbytes / 4));
    } else {
        hipLaunchKernelGGL(dev_copy1_kernel, dim3(cdiv((long)nbytes, 256)), dim3(256), 0, st, d, s, (long)nbytes);
    }
    return launch_ok("dev_copy");
}

}  // namespace sat
