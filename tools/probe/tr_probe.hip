// Probe 2: lane ti of each 16-lane group points at flat element 16*ti; LDS value = flat index (< 256, exact in bf16).
// Received value v  ->  source lane = v / 16, element within its 8 bytes = v % 16.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
    __shared__ __attribute__((aligned(16))) __bf16 t[256];
    for (int i = threadIdx.x; i < 256; i += 64) t[i] = (__bf16)(float)i;
    __syncthreads();
    int lane = threadIdx.x, ti = lane & 15;
    const __bf16* ptr = t + 16 * ti;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)ptr);
    unsigned long long raw = __builtin_bit_cast(unsigned long long, v);
    for (int e = 0; e < 4; ++e) { unsigned short bits = (unsigned short)(raw >> (16 * e)); out[lane * 4 + e] = __uint_as_float(((unsigned)bits) << 16); }
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 4); float h[256];
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l = 0; l < 16; ++l) { printf("lane %2d:", l); for (int e = 0; e < 4; ++e) printf("  src lane %2d elem %d", (int)h[l*4+e] / 16, (int)h[l*4+e] % 16); printf("\n"); }
    return 0;
}
