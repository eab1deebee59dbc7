// Probe: can an epilogue that writes bf16 results straight from the MFMA C^T register layout (8-byte pieces, a lane pair = 16 contiguous bytes of
// one row, 32 rows per store instruction, a lane's row segment completed over 16 consecutive stores) reach streaming bandwidth?
//   pattern 0: 16 bytes per lane, fully coalesced (reference: what the LDS-staged epilogue does)
//   pattern 1: the register-layout pattern: out[m][n] = a[m][n] + b[m][n] with m = lane % 32 + 32 * block, n = 8 * g + 4 * (lane / 32) + {0..3}
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/store_pattern tools/probe/store_pattern.hip ; run: /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef unsigned short bf16_t;

__global__ __launch_bounds__(256) void k_coalesced(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ c, long n16) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; e < n16; e += stride) {
        uint4 x = a[e], y = b[e];
        x.x ^= y.x; x.y ^= y.y; x.z ^= y.z; x.w ^= y.w;
        c[e] = x;
    }
}

// M rows, N columns (bf16); a wave owns 32 rows x NW columns; lane: row = lane % 32, column pieces 8 * g + 4 * (lane / 32), g = 0 .. NW / 8 - 1
template <int NW, int JOIN>
__global__ __launch_bounds__(256) void k_reglayout(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ c, long M, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ncg = N / NW;                                  // column groups
    const long nblk = (M / 32) * ncg;
    for (long blk = (long)blockIdx.x * 4 + wave; blk < nblk; blk += (long)gridDim.x * 4) {
        const long mb = blk / ncg; const int cg = (int)(blk - mb * ncg);
        const long row = mb * 32 + (lane & 31);
        const int col0 = cg * NW + 4 * (lane >> 5);
        uint2 va[NW / 8], vb[NW / 8];
#pragma unroll
        for (int g = 0; g < NW / 8; ++g) {
            va[g] = *reinterpret_cast<const uint2*>(a + row * N + col0 + 8 * g);
            if (JOIN) vb[g] = *reinterpret_cast<const uint2*>(b + row * N + col0 + 8 * g);
        }
#pragma unroll
        for (int g = 0; g < NW / 8; ++g) {
            uint2 o = va[g];
            if (JOIN) { o.x ^= vb[g].x; o.y ^= vb[g].y; }
            *reinterpret_cast<uint2*>(c + row * N + col0 + 8 * g) = o;
        }
    }
}

// pattern 2: after one cross-half exchange every lane owns 16 contiguous bytes: row = lane % 32, columns 16 * g + 8 * (lane / 32) + {0..7}
template <int NW, int JOIN>
__global__ __launch_bounds__(256) void k_reglayout16(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ c, long M, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ncg = N / NW;
    const long nblk = (M / 32) * ncg;
    for (long blk = (long)blockIdx.x * 4 + wave; blk < nblk; blk += (long)gridDim.x * 4) {
        const long mb = blk / ncg; const int cg = (int)(blk - mb * ncg);
        const long row = mb * 32 + (lane & 31);
        const int col0 = cg * NW + 8 * (lane >> 5);
        uint4 va[NW / 16], vb[NW / 16];
#pragma unroll
        for (int g = 0; g < NW / 16; ++g) {
            va[g] = *reinterpret_cast<const uint4*>(a + row * N + col0 + 16 * g);
            if (JOIN) vb[g] = *reinterpret_cast<const uint4*>(b + row * N + col0 + 16 * g);
        }
#pragma unroll
        for (int g = 0; g < NW / 16; ++g) {
            uint4 o = va[g];
            if (JOIN) { o.x ^= vb[g].x; o.y ^= vb[g].y; o.z ^= vb[g].z; o.w ^= vb[g].w; }
            *reinterpret_cast<uint4*>(c + row * N + col0 + 16 * g) = o;
        }
    }
}

int main() {
    const long M = 524288; const int N = 256;
    const long bytes = M * N * 2;
    bf16_t *a, *b, *c;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch, double moved) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-52s %8.1f us  %7.0f GB/s\n", name, ms * 100.0, moved / (ms * 1e-4) / 1e9);
    };
    const long n16 = bytes / 16;
    run("coalesced 16 B / lane: c = a ^ b (3 streams)", [&] { hipLaunchKernelGGL(k_coalesced, dim3(256 * 8), dim3(256), 0, 0, (const uint4*)a, (const uint4*)b, (uint4*)c, n16); }, 3.0 * bytes);
    run("register layout, NW = 128: c = a (2 streams)", [&] { hipLaunchKernelGGL((k_reglayout<128, 0>), dim3(256 * 8), dim3(256), 0, 0, a, b, c, M, N); }, 2.0 * bytes);
    run("register layout, NW = 128: c = a ^ b (3 streams)", [&] { hipLaunchKernelGGL((k_reglayout<128, 1>), dim3(256 * 8), dim3(256), 0, 0, a, b, c, M, N); }, 3.0 * bytes);
    run("register layout, NW = 256: c = a ^ b (3 streams)", [&] { hipLaunchKernelGGL((k_reglayout<256, 1>), dim3(256 * 8), dim3(256), 0, 0, a, b, c, M, N); }, 3.0 * bytes);
    run("register layout, NW = 64: c = a ^ b (3 streams)", [&] { hipLaunchKernelGGL((k_reglayout<64, 1>), dim3(256 * 8), dim3(256), 0, 0, a, b, c, M, N); }, 3.0 * bytes);
    run("register layout + exchange (16 B / lane), NW = 128, 3 streams", [&] { hipLaunchKernelGGL((k_reglayout16<128, 1>), dim3(256 * 8), dim3(256), 0, 0, a, b, c, M, N); }, 3.0 * bytes);
    run("register layout + exchange (16 B / lane), NW = 256, 3 streams", [&] { hipLaunchKernelGGL((k_reglayout16<256, 1>), dim3(256 * 8), dim3(256), 0, 0, a, b, c, M, N); }, 3.0 * bytes);
    run("register layout + exchange, NW = 128, 3 streams, 16 blocks / CU", [&] { hipLaunchKernelGGL((k_reglayout16<128, 1>), dim3(256 * 16), dim3(256), 0, 0, a, b, c, M, N); }, 3.0 * bytes);
    return 0;
}
