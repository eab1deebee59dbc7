// Dev probe: what rate can buffer_load ... lds (LDS-DMA) sustain per CU?  Each workgroup of NW waves keeps `stages` tiles of TILE bytes
// in flight into LDS and does nothing else; the source is a large buffer (HBM) or a small one (L2 resident).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/dma_probe tools/probe/dma_probe.hip ; run: tools/probe/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ void raw_buffer_load_lds(i32x4 rsrc, lptr_t lds, int size, int voffset, int soffset, int offset, int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");

template <int NW, int STAGES, int PIECES /* 1 KB DMA instructions per wave and tile */>
__global__ __launch_bounds__(64 * NW) void probe(const char* src, long src_bytes, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int TILE = NW * PIECES * 1024;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long b = (unsigned long long)src;
    i32x4 r; r[0] = (int)(unsigned)b; r[1] = (int)(unsigned)((b >> 32) & 0xFFFF); r[2] = (int)0x80000000; r[3] = 0x00020000;
    long tile0 = (long)blockIdx.x * iters;              // this workgroup streams its own contiguous range of tiles
    const long ntiles = src_bytes / TILE;
    auto issue = [&](int buf, long t) {
        const long base = (t % ntiles) * TILE;
#pragma unroll
        for (int p = 0; p < PIECES; ++p)
            raw_buffer_load_lds(r, (lptr_t)(smem + buf * TILE + (p * NW + wave) * 1024), 16, (int)(base + (p * NW + wave) * 1024 + lane * 16), 0, 0, 0);
    };
    for (int s = 0; s < STAGES - 1; ++s) issue(s, tile0 + s);
    int nxt = STAGES - 1;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (STAGES >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(nxt, tile0 + it + STAGES - 1);
        nxt = (nxt + 1 == STAGES) ? 0 : nxt + 1;
        acc += reinterpret_cast<float*>(smem)[(threadIdx.x * 4 + it) & 255];      // touch LDS a little
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 12345.678f) sink[0] = acc;
}

template <int NW, int STAGES, int PIECES>
void run(const char* name, const char* src, long bytes, int wgs_per_cu, float* sink) {
    constexpr int TILE = NW * PIECES * 1024;
    const int lds = STAGES * TILE;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<NW, STAGES, PIECES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    const int grid = 256 * wgs_per_cu, iters = 400;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<NW, STAGES, PIECES>), dim3(grid), dim3(64 * NW), lds, 0, src, bytes, iters, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double tb = (double)grid * iters * TILE / (best * 1e-3) / 1e12;
    printf("%-8s waves %d stages %d tile %3d KB  wg/cu %d (lds %3d KB)  in flight/CU %4d KB : %6.2f TB/s  (%5.1f B/clk/CU at 2.4 GHz)  %s\n", name, NW, STAGES, TILE / 1024,
           wgs_per_cu, lds / 1024, wgs_per_cu * (STAGES - 1) * TILE / 1024, tb, tb * 1e12 / 256 / 2.4e9, hipGetErrorString(hipGetLastError()));
}

int main() {
    const long big = 1L << 30, small = 16L << 20;       // 1 GiB: HBM;  16 MiB: L2 (4 MiB per XCD x 8) / MALL
    char* buf; hipMalloc(&buf, big); hipMemset(buf, 1, big);
    float* sink; hipMalloc(&sink, 64);
    for (int pass = 0; pass < 2; ++pass) {
        const char* nm = pass ? "L2/MALL" : "HBM"; const long bytes = pass ? small : big;
        run<4, 2, 8>(nm, buf, bytes, 1, sink); run<4, 2, 8>(nm, buf, bytes, 2, sink); run<4, 2, 8>(nm, buf, bytes, 3, sink);
        run<4, 3, 8>(nm, buf, bytes, 1, sink); run<4, 4, 8>(nm, buf, bytes, 1, sink);
        run<8, 2, 4>(nm, buf, bytes, 1, sink); run<8, 2, 8>(nm, buf, bytes, 1, sink); run<8, 2, 8>(nm, buf, bytes, 2, sink);
        run<4, 2, 4>(nm, buf, bytes, 4, sink); run<4, 2, 4>(nm, buf, bytes, 8, sink); run<4, 2, 2>(nm, buf, bytes, 8, sink);
        run<4, 4, 4>(nm, buf, bytes, 2, sink); run<8, 3, 4>(nm, buf, bytes, 1, sink); run<8, 4, 4>(nm, buf, bytes, 1, sink);
    }
    return 0;
}
