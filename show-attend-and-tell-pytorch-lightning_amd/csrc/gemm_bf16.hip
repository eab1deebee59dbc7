// bf16 MFMA GEMM / implicit GEMM for gfx950 (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
// Same operand modes as gemm.hip, 16x the matrix-core rate.  Operands may live in HBM as bf16
// (encoder activations / filter copies) or fp32 (decoder state and weights: rounded to bf16 on
// the way into LDS); results are written as bf16 or fp32.
//
// LDS images (bf16):
//   k-contiguous operand  -> [row][32 + 8]   one ds_read_b128 per 32-row x 16-k fragment,
//                                            80-byte rows => conflict free
//   k-major operand       -> [k][rows + 32]  kept as loaded (16-byte writes); fragments come from
//                                            two ds_read_b64_tr_b16 (hardware transpose), row stride
//                                            = 64 B mod 256 B => the 4 rows of a block hit disjoint banks
// Block = 256 threads (2 x 2 waves), tile BM x BN x KB (KB = 64 for k-contiguous operand forms, 32 for the
// transposed-read weight-gradient forms), register-staged double buffer.
#include <stdlib.h>

#include "gemm_bf16_common.h"
#include "profile.h"

namespace sat {

// ---- 16-byte gathers (raw bits; zero when out of range) ---------------------------------------
struct RowCtx { long base; int n, y0, x0; bool ok; };

template <int AM>
__device__ __forceinline__ void row_setup(const BArgs& a, int m, RowCtx& c) {
    c.ok = m < a.M; c.base = 0; c.n = c.y0 = c.x0 = 0;
    if (!c.ok) return;
    if (AM == A_ROW) {
        long r = m;
        if (a.a_rows) { int g = a.a_rows[m]; if (g < 0) { c.ok = false; return; } r = g; }
        c.base = r * a.lda;
    } else if (AM == A_CONV_FWD) {
        const ConvGeom& g = a.g;
        int pq = g.P * g.Q; c.n = m / pq; int r = m - c.n * pq; int p = r / g.Q, q = r - p * g.Q;
        c.y0 = p * g.stride - g.pad; c.x0 = q * (g.sw ? g.sw : g.stride) - g.pad;
    } else {
        const ConvGeom& g = a.g;
        int h, w;
        if (g.cls) { int hw = g.Hc * g.Wc; c.n = m / hw; int r = m - c.n * hw; int hc = r / g.Wc; h = 2 * hc + g.ph; w = 2 * (r - hc * g.Wc) + g.pw; }
        else { int hw = g.H * g.W; c.n = m / hw; int r = m - c.n * hw; h = r / g.W; w = r - h * g.W; }
        c.y0 = h + g.pad; c.x0 = w + g.pad;
    }
}

// Decoded k index of one k-tile row, shared through LDS so that the integer divisions are done by 32 lanes per
// tile instead of by every gathering lane: conv fwd / dgrad: (r, s, channel); wgrad: (image, y0, x0) of the pixel.
struct KEnt { int e0, e1, e2; };      // e2 < 0 : beyond the end of the reduction

template <int AM, int BMo>
__device__ __forceinline__ KEnt k_decode(const BArgs& a, int k, int kend) {
    KEnt t; t.e0 = t.e1 = 0; t.e2 = -1;
    if (k >= kend) return t;
    const ConvGeom& g = a.g;
    if (AM == A_CONV_FWD) { int rs = k / g.C; t.e2 = k - rs * g.C; t.e0 = rs / g.S; t.e1 = rs - t.e0 * g.S; }
    else if (AM == A_CONV_DGRAD) {
        int rs = k / g.K; t.e2 = k - rs * g.K;
        if (g.cls) { int ri = rs / g.sc; t.e0 = g.r0 + 2 * ri; t.e1 = g.s0 + 2 * (rs - ri * g.sc); }     // taps of this parity class only
        else { t.e0 = rs / g.S; t.e1 = rs - t.e0 * g.S; }
    }
    else if (BMo == B_CONV_WGRAD) {
        int pq = g.P * g.Q; int img = k / pq; int rem = k - img * pq; int p = rem / g.Q, q = rem - p * g.Q;
        t.e2 = img; t.e0 = p * g.stride - g.pad; t.e1 = q * (g.sw ? g.sw : g.stride) - g.pad;
    }
    return t;
}

template <int AM, typename T>
__device__ __forceinline__ uint4 row_fetch(const BArgs& a, const RowCtx& c, int k, int kend, const KEnt& t) {
    const uint4 z = make_uint4(0, 0, 0, 0);
    if (!c.ok || k >= kend) return z;
    const T* A = reinterpret_cast<const T*>(a.A);
    if (AM == A_ROW) return *reinterpret_cast<const uint4*>(A + c.base + k);
    const ConvGeom& g = a.g;
    if (AM == A_CONV_FWD) {
        int y = c.y0 + t.e0, x = c.x0 + t.e1;
        if ((unsigned)y >= (unsigned)g.H || (unsigned)x >= (unsigned)g.W) return z;
        return *reinterpret_cast<const uint4*>(A + (((long)c.n * g.H + y) * g.W + x) * g.C + t.e2);
    }
    int ty = c.y0 - t.e0, tx = c.x0 - t.e1;
    if (ty < 0 || tx < 0) return z;
    int p, q;
    if (g.stride == 1) { p = ty; q = tx; }
    else if (g.stride == 2) { if ((ty | tx) & 1) return z; p = ty >> 1; q = tx >> 1; }
    else { p = ty / g.stride; q = tx / g.stride; if (p * g.stride != ty || q * g.stride != tx) return z; }
    if (p >= g.P || q >= g.Q) return z;
    return *reinterpret_cast<const uint4*>(A + (((long)c.n * g.P + p) * g.Q + q) * g.K + t.e2);
}

template <typename T>
__device__ __forceinline__ uint4 brow_fetch(const BArgs& a, int n, int k, int kend) {
    if (n >= a.N || k >= kend) return make_uint4(0, 0, 0, 0);
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(a.B) + (long)n * a.ldb + k);
}

template <typename T>
__device__ __forceinline__ uint4 kmajor_fetch(const void* base, long ld, int k, int kend, int j, int J) {
    if (k >= kend || j >= J) return make_uint4(0, 0, 0, 0);
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(base) + (long)k * ld + j);
}

struct ColCtx { int r, s, ch; bool ok; };

template <int BMo, typename T>
__device__ __forceinline__ uint4 bk_fetch(const BArgs& a, const ColCtx& c, int n, int k, int kend, const KEnt& t) {
    if (BMo == B_KMAJOR) return kmajor_fetch<T>(a.B, a.ldb, k, kend, n, a.N);
    const uint4 z = make_uint4(0, 0, 0, 0);
    if (!c.ok || k >= kend) return z;
    const ConvGeom& g = a.g;
    const T* B = reinterpret_cast<const T*>(a.B);
    if (BMo == B_CONV_WGRAD) {          // t = (y0, x0, image) of output pixel k
        int y = t.e0 + c.r, x = t.e1 + c.s;
        if ((unsigned)y >= (unsigned)g.H || (unsigned)x >= (unsigned)g.W) return z;
        return *reinterpret_cast<const uint4*>(B + (((long)t.e2 * g.H + y) * g.W + x) * g.C + c.ch);
    }
    // B_CONV_DGRAD_W: t = (r, s, ko) of reduction index k
    return *reinterpret_cast<const uint4*>(B + (((long)t.e2 * g.R + t.e0) * g.S + t.e1) * g.C + n);
}

// raw 16 bytes -> bf16 in LDS.  bf16 source: 8 elements (ds_write_b128); fp32 source: 4 elements (ds_write_b64)
template <typename T>
__device__ __forceinline__ void lds_put(__bf16* dst, const uint4& raw) {
    if (sizeof(T) == 2) { *reinterpret_cast<uint4*>(dst) = raw; }
    else {
        bf16x4 o;
        o[0] = (__bf16)__uint_as_float(raw.x); o[1] = (__bf16)__uint_as_float(raw.y);
        o[2] = (__bf16)__uint_as_float(raw.z); o[3] = (__bf16)__uint_as_float(raw.w);
        *reinterpret_cast<bf16x4*>(dst) = o;
    }
}

template <int BM, int BN, int KB, int AM, int BMo, typename TA, typename TB, typename TC>
__global__ __launch_bounds__(NT) void gemm_bf16_kernel(BArgs a) {
    constexpr bool AK = (AM == A_KMAJOR);
    constexpr bool BKM = (BMo != B_ROW);
    constexpr int LDR = KB + 8;                  // k-contiguous rows
    constexpr int LDA_K = BM + 32, LDB_K = BN + 32;   // k-major rows
    constexpr int A_EL = AK ? KB * LDA_K : BM * LDR;
    constexpr int B_EL = BKM ? KB * LDB_K : BN * LDR;
    constexpr int STAGE = A_EL + B_EL;
    constexpr int VA = VecN<TA>::n, VB = VecN<TB>::n;
    constexpr int NVA = BM * KB / VA / NT, NVB = BN * KB / VB / NT;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr bool CONV = (AM == A_CONV_FWD || AM == A_CONV_DGRAD || BMo == B_CONV_WGRAD);
    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * STAGE];
    __shared__ KEnt ktab[2][KB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, by, bz;
    xcd_tile(bx, by, bz);
    const int bm = by * BM, bn = bx * BN;
    const int kbeg = bz * a.kchunk;
    const int kend = min(a.K, kbeg + a.kchunk);
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
    const int li = lane & 31, lh = lane >> 5;
    // transposed-read lane roles (ds_read_b64_tr_b16): 16-lane group g -> (k half, row half)
    const int tg = lane >> 4, ti = lane & 15;
    const int t_h = tg >> 1, t_mh = tg & 1, t_q = ti >> 2, t_p = ti & 3;

    RowCtx actx[NVA]; int a0[NVA], a1[NVA];
    ColCtx bctx[NVB]; int b0[NVB], b1[NVB];
#pragma unroll
    for (int j = 0; j < NVA; ++j) {
        int v = tid + j * NT;
        if (!AK) { a0[j] = v / (KB / VA); a1[j] = (v % (KB / VA)) * VA; row_setup<AM>(a, bm + a0[j], actx[j]); }
        else { a0[j] = v / (BM / VA); a1[j] = (v % (BM / VA)) * VA; }
    }
#pragma unroll
    for (int j = 0; j < NVB; ++j) {
        int v = tid + j * NT;
        if (!BKM) { b0[j] = v / (KB / VB); b1[j] = (v % (KB / VB)) * VB; }
        else {
            b0[j] = v / (BN / VB); b1[j] = (v % (BN / VB)) * VB;
            int n = bn + b1[j];
            bctx[j].ok = n < a.N; bctx[j].r = bctx[j].s = bctx[j].ch = 0;
            if (BMo == B_CONV_WGRAD && bctx[j].ok) { int rs = n / a.g.C; bctx[j].ch = n - rs * a.g.C; bctx[j].r = rs / a.g.S; bctx[j].s = rs - bctx[j].r * a.g.S; }
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint4 ra[NVA], rb[NVB];
    const KEnt none = {0, 0, -1};
    auto fill_tab = [&](int buf, int k0) { if (CONV && tid < KB) ktab[buf][tid] = k_decode<AM, BMo>(a, k0 + tid, kend); };
    auto fetch = [&](int k0, int tb) {
#pragma unroll
        for (int j = 0; j < NVA; ++j)
            ra[j] = AK ? kmajor_fetch<TA>(a.A, a.lda, k0 + a0[j], kend, bm + a1[j], a.M)
                       : row_fetch<AM, TA>(a, actx[j], k0 + a1[j], kend, (AM == A_ROW) ? none : ktab[tb][a1[j]]);
#pragma unroll
        for (int j = 0; j < NVB; ++j)
            rb[j] = BKM ? bk_fetch<BMo, TB>(a, bctx[j], bn + b1[j], k0 + b0[j], kend, (BMo == B_KMAJOR) ? none : ktab[tb][b0[j]])
                        : brow_fetch<TB>(a, bn + b0[j], k0 + b1[j], kend);
    };
    auto stash = [&](int buf) {
        __bf16* as = smem + buf * STAGE; __bf16* bs = as + A_EL;
#pragma unroll
        for (int j = 0; j < NVA; ++j) lds_put<TA>(as + (AK ? a0[j] * LDA_K + a1[j] : a0[j] * LDR + a1[j]), ra[j]);
#pragma unroll
        for (int j = 0; j < NVB; ++j) lds_put<TB>(bs + (BKM ? b0[j] * LDB_K + b1[j] : b0[j] * LDR + b1[j]), rb[j]);
    };

    int cur = 0;
    if (kbeg < kend) {
        if (CONV) { fill_tab(0, kbeg); fill_tab(1, kbeg + KB); __syncthreads(); }
        fetch(kbeg, 0);
        stash(0);
        __syncthreads();
        for (int k0 = kbeg, it = 0; k0 < kend; k0 += KB, ++it) {
            const bool more = (k0 + KB) < kend;
            if (more) fetch(k0 + KB, (it + 1) & 1);
            const __bf16* as = smem + cur * STAGE;
            const __bf16* bs = as + A_EL;
#pragma unroll
            for (int kk = 0; kk < KB; kk += 16) {
                bf16x8 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if (!AK) af[i] = *reinterpret_cast<const bf16x8*>(as + (wm + i * 32 + li) * LDR + kk + 8 * lh);
                    else {
                        const __bf16* p = as + (kk + 8 * t_h + t_q) * LDA_K + wm + i * 32 + 16 * t_mh + 4 * t_p;
                        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
                        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p + 4 * LDA_K));
                        af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (!BKM) bf[j] = *reinterpret_cast<const bf16x8*>(bs + (wn + j * 32 + li) * LDR + kk + 8 * lh);
                    else {
                        const __bf16* p = bs + (kk + 8 * t_h + t_q) * LDB_K + wn + j * 32 + 16 * t_mh + 4 * t_p;
                        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
                        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p + 4 * LDB_K));
                        bf[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
            if (more) stash(cur ^ 1);
            fill_tab(it & 1, k0 + 2 * KB);          // tile it+2; tile it's entries were consumed a full iteration ago
            __syncthreads();
            cur ^= 1;
        }
    }

    BnAcc<BM, BN, NT> bnacc(a, bn, tid);
    store_tile<BM, BN, TC, 2 * STAGE>(a, acc, smem, bm, bn, bz, wm, wn, tid, lane, bnacc, false);
}

static const char* mname(int am, int bm) {
    if (am == A_CONV_FWD) return "conv_fwd";
    if (am == A_CONV_DGRAD) return "conv_dgrad";
    if (bm == B_CONV_WGRAD) return "conv_wgrad";
    if (am == A_ROW && bm == B_ROW) return "nt";
    if (am == A_ROW && bm == B_KMAJOR) return "nn";
    return "tn";
}

template <int AM> struct KTile { static constexpr int v = 64; };   // measured on the C2 step: 64 beats 32 for every operand form

template <int BM, int BN, int AM, int BMo, typename TA, typename TB, typename TC>
static int runb(const BArgs& k, hipStream_t st) {
    constexpr int KB = KTile<AM>::v;
    dim3 grid(cdiv(k.N, BN), cdiv(k.M, BM), k.nsplit);
    char pname[128];
    if (profile_enabled()) {
        static const bool shapes = getenv("SAT_PROFILE_SHAPES") != nullptr;      // dev: one profile line per problem shape
        if (shapes) snprintf(pname, sizeof pname, "gemm_bf16_%s_%dx%d_%s M%d N%d K%d z%d", mname(AM, BMo), BM, BN, sizeof(TA) == 2 ? "b" : "f", k.M, k.N, k.K, k.nsplit);
        else snprintf(pname, sizeof pname, "gemm_bf16_%s_%dx%d_%s", mname(AM, BMo), BM, BN, sizeof(TA) == 2 ? "b" : "f");
    }
    ProfScope prof(pname, 2.0 * k.M * k.N * k.K, (double)sizeof(TA) * k.M * k.K + (double)sizeof(TB) * k.N * k.K + (double)sizeof(TC) * k.M * k.N, st);
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, KB, AM, BMo, TA, TB, TC>), grid, dim3(NT), 0, st, k);
    SAT_TRY(launch_ok("gemm_bf16_kernel"));
    if (k.nsplit > 1) SAT_TRY(launch_splitk_reduce<TC>(k, st));
    return SAT_OK;
}

template <int AM, int BMo, typename TA, typename TB, typename TC>
static int runb_tiles(const BArgs& k, int BMt, hipStream_t st) {
    if (BMt == 128) return runb<128, 128, AM, BMo, TA, TB, TC>(k, st);
    return runb<64, 64, AM, BMo, TA, TB, TC>(k, st);
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// 1 when the bf16-MFMA kernel can take this problem (16-byte gathers need aligned, divisible extents)
int gemm_bf16_eligible(const GemmArgs& g) {
    const int va = g.a_bf16 ? 8 : 4, vb = g.b_bf16 ? 8 : 4;
    if (!al16(g.A) || !al16(g.B)) return 0;
    switch (g.amode) {
        case A_ROW: if (g.K % va || g.lda % va) return 0; break;
        case A_KMAJOR: if (g.M % va || g.lda % va) return 0; break;
        case A_CONV_FWD: if (g.g.C % va) return 0; break;
        case A_CONV_DGRAD: if (g.g.K % va) return 0; break;
        default: return 0;
    }
    switch (g.bmode) {
        case B_ROW: if (g.K % vb || g.ldb % vb) return 0; break;
        case B_KMAJOR: if (g.N % vb || g.ldb % vb) return 0; break;
        case B_CONV_WGRAD: if (g.g.C % vb) return 0; break;
        case B_CONV_DGRAD_W: if (g.g.C % vb) return 0; break;
        default: return 0;
    }
    return 1;
}

int& gemm_tile_override() { static int v = -1; return v; }      // dev: sat_debug_option("tile_override", BMt): 64, 128, 256 (256x256), 257 (256x128)

static int launch_gemm_bf16_tile(const GemmArgs& g, int tile_mode, hipStream_t st);

// 8-wave tile forms of gemm_glds.hip (256x256 / 256x128, one workgroup per CU): chosen where they cut the L2 -> LDS operand traffic
// of the 128x128 form - every operand tile is re-read once per tile of the OTHER dimension - and still fill the chip.
static int pick_wide_tile(const GemmArgs& g) {
    // Off by default: the forms win -4 .. -22 % launch by launch on cold caches (tools/sweep_tiles.py) but the whole C2 step does not
    // get faster with them (25.1 vs 25.0 ms, bench.py A/B on one box): inside the step the operands of these launches are L2 / MALL
    // resident from their producers and the 128x128 form's two workgroups per CU overlap better.  SAT_WIDE_TILES=1 turns them on.
    const int wide = dev_switch(SW_WIDE_TILES);
    if (!wide || !g.a_bf16 || !g.b_bf16 || g.a_rows || g.c_rows || g.K % 64) return 0;
    // measured on the C2 step's convolutions (tools/sweep_tiles.py, resnet50, 128 images):
    //   weight gradients with >= 256 filters and >= 256 filter columns: 256x256 (-10 .. -22 %; the operand panels are re-read half as often)
    //   forward 3x3 / 1x1 forms with >= 256 filters over >= 8192 pixels:    256x128 (-4 .. -9 %)
    //   data gradients: no form wins
    if (g.amode == A_KMAJOR && !g.c_bf16) return (g.M >= 256 && g.N >= 256 && g.K >= 4096) ? 256 : 0;
    if ((g.amode == A_CONV_FWD || (g.amode == A_ROW && g.bmode == B_ROW)) && g.c_bf16 && !g.accumulate)
        return (g.N >= 256 && g.M >= 8192 && (long)cdiv(g.M, 256) * cdiv(g.N, 128) >= 128) ? 257 : 0;
    return 0;
}

int launch_gemm_bf16(const GemmArgs& g, hipStream_t st) {
    if (g.M == 0 || g.N == 0) return SAT_OK;
    int mode = gemm_tile_override() > 0 ? gemm_tile_override() : pick_wide_tile(g);
    if (mode >= 256) {
        const int r = launch_gemm_bf16_tile(g, mode, st);
        if (r != -1) return r;
    }
    return launch_gemm_bf16_tile(g, (mode == 64 || mode == 128) ? mode : 0, st);
}

// tile_mode: 0 = automatic 64 / 128; 64, 128 forced; 256 = 256x256, 257 = 256x128 (direct-to-LDS forms only: -1 when the problem does not fit them)
static int launch_gemm_bf16_tile(const GemmArgs& g, int tile_mode, hipStream_t st) {
    SAT_REQUIRE(g.A && g.B && g.C, "gemm_bf16: null operand");
    SAT_REQUIRE(gemm_bf16_eligible(g), "gemm_bf16: operands not 16-byte gatherable (mode %d,%d M=%d N=%d K=%d)", g.amode, g.bmode, g.M, g.N, g.K);
    BArgs k;
    k.A = g.A; k.lda = g.lda; k.a_rows = g.a_rows; k.B = g.B; k.ldb = g.ldb; k.C = g.C; k.ldc = g.ldc; k.c_rows = g.c_rows;
    k.M = g.M; k.N = g.N; k.K = g.K; k.accumulate = g.accumulate; k.epi = g.epi; k.bias = g.bias; k.e0 = g.e0; k.lde0 = g.lde0;
    k.c0 = g.c0; k.c1 = g.c1; k.g = g.g; k.slab = g.slab;
    const int KB = 64;
    int BMt = 64;
    static const long tiles128_min = getenv("SAT_TILES128_MIN") ? atol(getenv("SAT_TILES128_MIN")) : 192;
    const long tiles128 = (long)cdiv(g.M, 128) * cdiv(g.N, 128);
    if (tiles128 >= tiles128_min && g.M >= 128 && g.N >= 128) BMt = 128;
    // short reductions over a grid of less than two 128-tile rounds of the chip (the per-time-step vocabulary projection of decoder_tf=None:
    // 640 x 6400, K = 256 -> 250 tiles): one workgroup per CU walks load - 4 k-tiles - 64 KB store alone; 64-wide tiles put four on a CU
    // (26.8 -> 15.1 us)
    if (BMt == 128 && tiles128 < 512 && g.K <= 256 && !g.slab) BMt = 64;
    // long reductions (weight gradients): split-K supplies the parallelism, so keep the 64x64-per-wave tile
    if (g.slab && g.M >= 128 && g.N >= 128 && g.K >= 64 * KB) BMt = 128;
    if (tile_mode) BMt = tile_mode;
    const int tbm = BMt >= 256 ? 256 : BMt, tbn = BMt == 256 ? 256 : (BMt == 257 ? 128 : BMt);
    long blocks = (long)cdiv(g.M, tbm) * cdiv(g.N, tbn);
    int ns = 1;
    if (g.slab && blocks < 256 && g.K >= 16 * KB) {
        // split-K: enough blocks to keep every CU streaming (~TARGET blocks), but the fp32 partial slabs (written and
        // re-read: 8 bytes per output element per split) must stay a fraction of the operand bytes
        static const int target = getenv("SAT_SPLIT_TARGET") ? atoi(getenv("SAT_SPLIT_TARGET")) : 768;
        static const int frac = getenv("SAT_SPLIT_FRAC") ? atoi(getenv("SAT_SPLIT_FRAC")) : 8;
        static const int target128 = getenv("SAT_SPLIT_TARGET128") ? atoi(getenv("SAT_SPLIT_TARGET128")) : 512;
        static const int target256 = getenv("SAT_SPLIT_TARGET256") ? atoi(getenv("SAT_SPLIT_TARGET256")) : 256;
        const int tgt = (BMt >= 256) ? target256 : (BMt == 128) ? target128 : target;      // resident workgroups: 2 per CU with 128-wide tiles, ~3 with 64-wide; rounding DOWN keeps the grid inside one residency round
        static const int floor_mode = getenv("SAT_SPLIT_FLOOR") ? atoi(getenv("SAT_SPLIT_FLOOR")) : 1;
        int want = floor_mode ? (int)(tgt / blocks) : (int)((tgt + blocks - 1) / blocks), maxs = g.K / (8 * KB);
        if (want < 1) want = 1;
        double in_bytes = ((double)g.M + g.N) * g.K * 2.0;
        int lim = (int)(in_bytes / ((double)frac * g.M * g.N)); if (lim < 4) lim = 4;
        ns = want < maxs ? want : maxs; if (ns > lim) ns = lim; if (ns > 256) ns = 256; if (ns < 1) ns = 1;
        while (ns > 1 && (long)ns * g.M * g.N > g.slab_elems) --ns;
    }
    int ktiles = cdiv(g.K, KB); if (ktiles < 1) ktiles = 1;
    int per = cdiv(ktiles, ns);
    k.kchunk = per * KB; k.nsplit = cdiv(ktiles, per);
    k.wide_store = (g.c_bf16 && !g.c_rows && k.nsplit == 1 && g.N % 8 == 0 && g.ldc % 8 == 0 && al16(g.C) &&
                    (g.epi == EPI_NONE || g.epi == EPI_BIAS || g.epi == EPI_BIAS_RELU)) ? 1 : 0;
    if (!g.c_bf16 && !g.c_rows && g.N % 4 == 0 && g.ldc % 4 == 0 && al16(g.C)) k.wide_store = 1;      // fp32 result (with split-K: of the reduce kernel)
    k.wide_slab = (k.nsplit > 1 && g.N % 4 == 0 && al16(g.slab)) ? 1 : 0;
    k.tile_stats = nullptr;
    if (g.tile_rows) *g.tile_rows = 0;
    if (g.tile_stats && g.tile_rows && g.c_bf16 && k.wide_store && !g.accumulate && !g.g.cls && !g.bn_x) { k.tile_stats = g.tile_stats; *g.tile_rows = tbm; }
    // BatchNorm-backward statistics of a data-gradient launch (accumulating launches included: the statistics see the final values)
    if (g.tile_stats && g.tile_rows && g.bn_x && g.bn_mean && g.bn_invstd && g.c_bf16 && k.wide_store && k.nsplit == 1 && !g.g.cls && g.ldc == g.N &&
        g.epi == EPI_NONE && al16(g.bn_x) && BMt != 256) {
        k.tile_stats = g.tile_stats; *g.tile_rows = tbm;
        k.bn_x = reinterpret_cast<const __bf16*>(g.bn_x); k.bn_mask = g.bn_mask; k.bn_mean = g.bn_mean; k.bn_invstd = g.bn_invstd;
    }

    k.acc_prefetch = dev_switch(SW_ACC_PREFETCH);
    if (g.add_src) {
        SAT_REQUIRE(g.c_bf16 && k.wide_store && k.nsplit == 1 && g.accumulate && al16(g.add_src) && (!g.add_mask || g.ldc % 8 == 0),
                    "gemm_bf16: add_src needs an accumulating bf16 launch on the 16-byte store path (M=%d N=%d ldc=%ld)", g.M, g.N, g.ldc);
        k.add_src = reinterpret_cast<const __bf16*>(g.add_src); k.add_mask = g.add_mask;
    }
    static const bool log_shapes = getenv("SAT_LOG_GEMM") != nullptr;     // dev: one stderr line per launch, in launch order
    if (log_shapes) fprintf(stderr, "GEMMLOG am=%d bm=%d M=%d N=%d K=%d ns=%d acc=%d types=%d%d%d epi=%d\n", g.amode, g.bmode, g.M, g.N, g.K, k.nsplit,
                            g.accumulate, g.a_bf16, g.b_bf16, g.c_bf16, g.epi);
#define SAT_BCASE(AMV, BMV, TA, TB, TC) return runb_tiles<AMV, BMV, TA, TB, TC>(k, BMt, st);
    const bool ab = g.a_bf16, bb = g.b_bf16, cb = g.c_bf16;
    if (ab && bb) {         // bf16 operands in HBM: direct-to-LDS staging where the form allows it
        int bm_used = BMt;
        const int r = launch_gemm_glds(k, g.amode, g.bmode, cb, BMt, st, &bm_used);
        if (r != -1) { if (k.tile_stats && g.tile_rows) *g.tile_rows = bm_used; return r; }
    }
    if (BMt >= 256) { if (g.tile_rows) *g.tile_rows = 0; return -1; }
    if (ab && bb) {         // encoder: bf16 activations / filters
        if (g.amode == A_CONV_FWD && g.bmode == B_ROW && cb) SAT_BCASE(A_CONV_FWD, B_ROW, __bf16, __bf16, __bf16)
        if (g.amode == A_ROW && g.bmode == B_ROW && cb) SAT_BCASE(A_ROW, B_ROW, __bf16, __bf16, __bf16)
        if (g.amode == A_ROW && g.bmode == B_ROW && !cb) SAT_BCASE(A_ROW, B_ROW, __bf16, __bf16, float)
        if (g.amode == A_CONV_DGRAD && g.bmode == B_CONV_DGRAD_W && cb) SAT_BCASE(A_CONV_DGRAD, B_CONV_DGRAD_W, __bf16, __bf16, __bf16)
        if (g.amode == A_ROW && g.bmode == B_KMAJOR && cb) SAT_BCASE(A_ROW, B_KMAJOR, __bf16, __bf16, __bf16)
        if (g.amode == A_ROW && g.bmode == B_KMAJOR && !cb) SAT_BCASE(A_ROW, B_KMAJOR, __bf16, __bf16, float)
        if (g.amode == A_KMAJOR && g.bmode == B_CONV_WGRAD && !cb) SAT_BCASE(A_KMAJOR, B_CONV_WGRAD, __bf16, __bf16, float)
        if (g.amode == A_KMAJOR && g.bmode == B_KMAJOR && !cb) SAT_BCASE(A_KMAJOR, B_KMAJOR, __bf16, __bf16, float)
    } else if (!ab && !bb && !cb) {   // decoder: fp32 in HBM, rounded to bf16 on the way into LDS
        if (g.amode == A_ROW && g.bmode == B_ROW) SAT_BCASE(A_ROW, B_ROW, float, float, float)
        if (g.amode == A_ROW && g.bmode == B_KMAJOR) SAT_BCASE(A_ROW, B_KMAJOR, float, float, float)
        if (g.amode == A_KMAJOR && g.bmode == B_KMAJOR) SAT_BCASE(A_KMAJOR, B_KMAJOR, float, float, float)
    }
#undef SAT_BCASE
    return fail(SAT_EUNSUPPORTED, "gemm_bf16: combination amode=%d bmode=%d types(%d,%d,%d) not built", g.amode, g.bmode, ab, bb, cb);
}

}  // namespace sat
