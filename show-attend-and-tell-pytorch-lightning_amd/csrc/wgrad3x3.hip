// Weight gradient of the 3x3 / stride 1 / pad 1 convolutions (torchvision ResNet blocks, model.py:19-29), all nine filter taps per
// workgroup.
//
// The implicit-GEMM form (gemm_glds.hip, B_CONV_WGRAD) treats dW[K][9*C] as nine independent column panels: every panel gathers its own
// shifted copy of the input pixels, so the activations cross the L2 -> LDS path once per tap and per 128-filter row tile (18 .. 36 passes
// over x and dy for the ResNet-50 shapes; measured 2 - 4 TB/s of fill for 36 - 170 MB of algorithmic bytes: 86 - 197 us per launch).
// Here a workgroup of NINE waves owns a 64-filter x 64-channel block of dW for ALL taps: a k-tile is 64 consecutive output pixels
// (64 / W output rows of one image) and brings in
//     dy   [64 pixels][64 filters]                     8 KB, k-major, as in gemm_glds.hip
//     halo [(64/W + 2) rows x (W + 2) pixels][64 ch]   13 - 26 KB: every input pixel the nine taps of those outputs touch, ONCE,
//                                                      zero padding = out-of-range lanes of the buffer descriptor
// and wave t multiplies dy^T with the halo window shifted by tap t = (r, s): its k-major B fragments are ds_read_b64_tr_b16 reads
// whose pixel rows are (row + r, col + s) of the halo - eight per-lane offsets computed once.  Per k-tile 144 MFMAs (32x32x16) for
// ~33 KB of LDS fill, 2.2x less fill per product than the 128x128 form and no activation byte fetched twice by the same workgroup.
// Split over pixel ranges (grid z) with fp32 partials in the caller's slab, combined by the fixed-order reduce kernel the other forms use.
#include <stdlib.h>

#include "gemm_bf16_common.h"
#include "profile.h"

namespace sat {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr3_t;
__device__ void raw_buffer_load_lds_w(i32x4 rsrc, lptr3_t lds, int size, int voffset, int soffset, int offset, int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");

namespace {
constexpr int W3_OOB = (int)0x80000000;
constexpr int W3_WAVES = 9, W3_THREADS = 64 * W3_WAVES;
constexpr int W3_MAXP = 4;          // LDS-DMA instructions per wave and k-tile, at most (33 pieces / 9 waves)
constexpr int W3_STAGES = 3;        // LDS ring: two k-tiles in flight under the one being multiplied (128 VGPRs: one 9-wave workgroup per CU)

__device__ __forceinline__ i32x4 w3_rsrc(const void* base) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    i32x4 r;
    r[0] = (int)(unsigned)b; r[1] = (int)(unsigned)((b >> 32) & 0xFFFF);
    r[2] = W3_OOB; r[3] = 0x00020000;
    return r;
}

struct W3Args {
    const __bf16* dy; const __bf16* x; float* out;      // out: dW (Z == 1) or the slab [Z][K][9*C]
    int Nimg, H, W, C, K;
    int nr;                  // output rows per k-tile = 64 / W
    int units;               // k-tiles in the whole problem = Nimg * H / nr
    int per;                 // k-tiles per workgroup (grid.z = ceil(units / per))
    int HP;                  // halo pixels (nr + 2) * (W + 2), rounded up to a multiple of 8
    int Z;
};

__global__ __launch_bounds__(W3_THREADS) void wgrad3x3_kernel(W3Args a) {
    extern __shared__ __attribute__((aligned(1024))) __bf16 smem[];      // W3_STAGES x (dy tile 64 x 64 | halo HP x 64)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64, bz = blockIdx.z;
    const int W = a.W, H = a.H, C = a.C, K = a.K, nr = a.nr, HW2 = W + 2;
    const int STAGE = 64 * 64 + a.HP * 64;
    const int u0 = bz * a.per, u1 = min(a.units, u0 + a.per);
    const int upi = H / nr;                                               // k-tiles per image
    const i32x4 rA = w3_rsrc(a.dy), rX = w3_rsrc(a.x);

    // ---- LDS-DMA pieces of this wave: piece i of a k-tile goes to wave i % 9; pieces 0..7 = dy rows, 8.. = halo pixels (8 per piece)
    const int npieces = 8 + a.HP / 8;
    int p_off[W3_MAXP]; int p_hr[W3_MAXP]; bool p_dy[W3_MAXP], p_on[W3_MAXP], p_colok[W3_MAXP]; int p_lds[W3_MAXP];
#pragma unroll
    for (int j = 0; j < W3_MAXP; ++j) {
        const int piece = j * W3_WAVES + wave;
        p_on[j] = piece < npieces; p_dy[j] = piece < 8; p_hr[j] = 0; p_colok[j] = false; p_off[j] = 0;
        const int sl = lane & 7;
        if (p_dy[j]) {                                 // dy: k-row = pixel of the tile, 64 filters = 8 slots of 16 bytes
            const int krow = piece * 8 + (lane >> 3);
            const int lq = (sl >> 2) ^ ((krow >> 1) & 1);
            p_off[j] = (krow * K + co0 + (lq * 4 + (sl & 3)) * 8) * 2;
            p_lds[j] = piece * 512;
        } else {                                       // halo pixel hp: input pixel (p0 - 1 + hr, hc) of the image
            const int hp = (piece - 8) * 8 + (lane >> 3);
            const int hr = hp / HW2, hc = hp - hr * HW2 - 1;
            const int lq = (sl >> 2) ^ ((hp >> 1) & 1);
            p_hr[j] = hr;
            p_colok[j] = p_on[j] && hr < nr + 2 && hc >= 0 && hc < W;
            p_off[j] = ((hr * W + hc) * C + ci0 + (lq * 4 + (sl & 3)) * 8) * 2;
            p_lds[j] = 64 * 64 + (piece - 8) * 512;
        }
    }
    auto issue = [&](int buf, int u) {                // every piece of k-tile u (wave-uniform: image n, first output row p0)
        const int n = u / upi, p0 = (u - n * upi) * nr;
        const int baseA = ((n * H + p0) * W) * K * 2;                      // first output pixel of the tile
        const int baseX = ((n * H + p0 - 1) * W) * C * 2;                  // input row p0 - 1 (row validity per lane below)
        __bf16* st = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < W3_MAXP; ++j) {
            if (!p_on[j]) continue;
            if (p_dy[j]) raw_buffer_load_lds_w(rA, (lptr3_t)(st + p_lds[j]), 16, p_off[j] + baseA, 0, 0, 0);
            else {
                const int ih = p0 - 1 + p_hr[j];
                const bool ok = p_colok[j] && ih >= 0 && ih < H;
                raw_buffer_load_lds_w(rX, (lptr3_t)(st + p_lds[j]), 16, ok ? p_off[j] + baseX : W3_OOB, 0, 0, 0);
            }
        }
    };

    // ---- fragment read offsets (elements).  A = dy^T: k-major [pixel][filter]; B = halo window of this wave's tap
    const int tg = lane >> 4, ti = lane & 15;
    const int t_h = tg >> 1, t_mh = tg & 1, t_q = ti >> 2, t_p = ti & 3;
    const int tap_r = wave / 3, tap_s = wave - tap_r * 3;
    int a_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int col = i * 32 + 16 * t_mh + 4 * t_p, kr = 8 * t_h + t_q;
        a_off[i] = kr * 64 + (((col >> 5) ^ ((kr >> 1) & 1)) << 5) + (col & 31);
    }
    int b_off[4][2][2];                                // [k-step][rows k, k + 4][channel block]
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int k = ks * 16 + 8 * t_h + t_q + 4 * hf;
            const int pr = k / W, q = k - pr * W;
            const int hp = (pr + tap_r) * HW2 + q + tap_s;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = j * 32 + 16 * t_mh + 4 * t_p;
                b_off[ks][hf][j] = 64 * 64 + hp * 64 + (((col >> 5) ^ ((hp >> 1) & 1)) << 5) + (col & 31);
            }
        }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int mine = 0;                                      // LDS-DMA instructions this wave issues per k-tile (3 or 4: wave-uniform)
#pragma unroll
    for (int j = 0; j < W3_MAXP; ++j) mine += p_on[j] ? 1 : 0;
    mine = __builtin_amdgcn_readfirstlane(mine);
    if (u0 < u1) {
        issue(0, u0);
        if (u0 + 1 < u1) issue(1, u0 + 1);
        int cur = 0, nxt = 2;
        for (int u = u0; u < u1; ++u) {
            // tile u has landed when at most the pieces of tile u + 1 are still outstanding (counted wait: the newer tile stays in flight)
            if (u + 1 < u1) {
                if (mine == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (mine == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else if (mine == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (u + 2 < u1) issue(nxt, u + 2);        // the stage of tile u - 1: every wave has finished reading it (it passed the barrier)
            const __bf16* st = smem + cur * STAGE;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const __bf16* p = st + a_off[i] + ks * 16 * 64;
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p + 4 * 64));
                    af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(st + b_off[ks][0][j]));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(st + b_off[ks][1][j]));
                    bf[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
            cur = (cur + 1 == W3_STAGES) ? 0 : cur + 1;
            nxt = (nxt + 1 == W3_STAGES) ? 0 : nxt + 1;
        }
    }
    // ---- write-out: dW[filter][tap][channel] (KRSC) or this split's slab; C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int li = lane & 31, lh = lane >> 5;
    const long N9 = 9L * C;
    float* out = a.out + (a.Z > 1 ? (long)bz * K * N9 : 0L);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = co0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int col = ci0 + j * 32 + li;
                out[(long)row * N9 + (long)wave * C + col] = acc[i][j][r];
            }
}
}  // namespace

// 0 when the geometry is not this kernel's (the caller keeps the implicit-GEMM form)
int wgrad3x3_eligible(const ConvGeom& g) {
    if (!dev_switch(SW_WGRAD3X3)) return 0;
    if (g.R != 3 || g.S != 3 || g.stride != 1 || g.pad != 1 || (g.sw && g.sw != 1)) return 0;
    if (g.C % 64 || g.K % 64) return 0;
    if (g.H != g.P || g.W != g.Q) return 0;
    if (!(g.W == 8 || g.W == 16 || g.W == 32 || g.W == 64)) return 0;
    if (g.H % (64 / g.W)) return 0;
    if ((long)g.N * g.H * g.W * (g.C > g.K ? g.C : g.K) >= (1L << 30)) return 0;      // 32-bit byte offsets
    return 1;
}

int launch_wgrad3x3(const void* dy, const void* x, float* dw, const ConvGeom& g, float* slab, long slab_elems, hipStream_t st) {
    W3Args a;
    a.dy = reinterpret_cast<const __bf16*>(dy); a.x = reinterpret_cast<const __bf16*>(x);
    a.Nimg = g.N; a.H = g.H; a.W = g.W; a.C = g.C; a.K = g.K;
    a.nr = 64 / g.W; a.units = g.N * (g.H / a.nr);
    a.HP = ((a.nr + 2) * (g.W + 2) + 7) / 8 * 8;
    SAT_REQUIRE(8 + a.HP / 8 <= W3_WAVES * W3_MAXP, "wgrad3x3: %d LDS-DMA pieces per k-tile", 8 + a.HP / 8);
    const long tiles = (long)(g.C / 64) * (g.K / 64), out_elems = (long)g.K * 9 * g.C;
    // pixel splits: about two workgroups per CU, at least 8 k-tiles each, partial slabs within the caller's scratch
    static const int target = getenv("SAT_W3_TARGET") ? atoi(getenv("SAT_W3_TARGET")) : 256;      // one 9-wave workgroup per CU
    long z = target / tiles; if (z < 1) z = 1;
    if (z > a.units / 8) z = a.units / 8 > 0 ? a.units / 8 : 1;
    while (z > 1 && (!slab || z * out_elems > slab_elems)) --z;
    a.per = (int)cdiv(a.units, z); a.Z = (int)cdiv(a.units, a.per);
    a.out = a.Z > 1 ? slab : dw;
    const size_t lds = W3_STAGES * (size_t)(64 * 64 + a.HP * 64) * sizeof(__bf16);
    static bool attr_set = false;
    if (!attr_set) {
        SAT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_set = true;
    }
    {
        char pname[128];
        if (profile_enabled()) {
            static const bool shapes = getenv("SAT_PROFILE_SHAPES") != nullptr;
            if (shapes) snprintf(pname, sizeof pname, "gemm_wgrad3x3 M%d N%d K%d z%d", g.K, 9 * g.C, g.N * g.H * g.W, a.Z);
            else snprintf(pname, sizeof pname, "gemm_wgrad3x3");
        }
        ProfScope prof(pname, 2.0 * g.K * 9.0 * g.C * g.N * g.H * g.W, 2.0 * g.N * g.H * g.W * ((double)g.C + g.K) + 4.0 * out_elems, st);
        hipLaunchKernelGGL(wgrad3x3_kernel, dim3(g.C / 64, g.K / 64, a.Z), dim3(W3_THREADS), lds, st, a);
        SAT_TRY(launch_ok("wgrad3x3_kernel"));
        if (a.Z > 1) {
            BArgs k{};
            k.M = g.K; k.N = 9 * g.C; k.C = dw; k.ldc = 9 * g.C; k.slab = slab; k.nsplit = a.Z; k.wide_store = 1; k.wide_slab = 1; k.epi = EPI_NONE;
            SAT_TRY(launch_splitk_reduce<float>(k, st));
        }
    }
    return SAT_OK;
}

}  // namespace sat
