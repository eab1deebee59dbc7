// fp32 MFMA GEMM / implicit GEMM for gfx950.
//
// C[M x N] = A[M x K] * B[K x N], fp32 in / fp32 accumulate on
// v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fmaf chain, 64 FLOP/clk/SIMD).
// One kernel template covers the dense Linear forms (NT / NN / TN) and the three
// convolution forms (forward, data gradient, weight gradient) by switching the
// global->LDS gather of each operand; the LDS tile keeps the operand the way it
// was read (k-contiguous rows get an odd leading dimension, k-major tiles are read
// lane-linear) so every fragment read is a conflict-free ds_read_b32.
//
// Block = 256 threads = 4 waves (2 x 2); tile BM x BN x 16, register-staged double
// buffer: the next tile's global loads are issued before the MFMAs of the current
// one and written to the other LDS buffer afterwards (one barrier per k-tile).
#include "gemm.h"
#include "profile.h"

namespace sat {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;
constexpr int NTHREADS = 256;

struct KArgs {
    const float* A; long lda; const int* a_rows;
    const float* B; long ldb;
    float* C; long ldc; const int* c_rows;
    int M, N, K;
    int accumulate, epi;
    const float* bias; const float* e0; long lde0; int c0, c1;
    ConvGeom g;
    int avec, bvec;       // 16-byte loads allowed
    int kchunk, nsplit;   // split-K
    float* slab;
};

__device__ __forceinline__ float epilogue_value(const KArgs& a, int row, int col, float v) {
    switch (a.epi) {
        case EPI_BIAS: v += a.bias[col]; break;
        case EPI_BIAS_SIGMOID_RANGE:
            if (a.bias) v += a.bias[col];
            if (col >= a.c0 && col < a.c1) v = fast_sigmoid(v);
            break;
        case EPI_ADD_TANH: {
            long er = a.a_rows ? (long)a.a_rows[row] : (long)row;
            v = fast_tanh(v + a.e0[er * a.lde0 + col]);
        } break;
        case EPI_MUL_DTANH: {
            float u = a.e0[(long)row * a.lde0 + col];
            v *= (1.0f - u * u);
        } break;
        case EPI_BIAS_RELU: v += a.bias[col]; v = v < 0.f ? 0.f : v; break;        // NaN stays NaN
        default: break;
    }
    return v;
}

__device__ __forceinline__ void store_out(const KArgs& a, int row, int col, float v) {
    long orow = row;
    if (a.c_rows) { int r = a.c_rows[row]; if (r < 0) return; orow = r; }
    float* p = a.C + orow * a.ldc + col;
    if (a.accumulate) v += *p;
    *p = epilogue_value(a, row, col, v);
}

// ---- operand fetch: 4 consecutive elements starting at (idx0, idx1) ------------
// Row-like A modes: vector runs along k.  `row` is the GEMM row, k the GEMM k.
template <int AM>
struct ARowCtx { long base; int n, y0, x0; bool ok; };

template <int AM>
__device__ __forceinline__ void a_row_setup(const KArgs& a, int m, ARowCtx<AM>& c) {
    c.ok = m < a.M; c.base = 0; c.n = c.y0 = c.x0 = 0;
    if (!c.ok) return;
    if (AM == A_ROW) {
        long r = m;
        if (a.a_rows) { int g = a.a_rows[m]; if (g < 0) { c.ok = false; return; } r = g; }
        c.base = r * a.lda;
    } else if (AM == A_CONV_FWD) {
        const ConvGeom& g = a.g;
        int pq = g.P * g.Q; c.n = m / pq; int r = m - c.n * pq; int p = r / g.Q, q = r - p * g.Q;
        c.y0 = p * g.stride - g.pad; c.x0 = q * g.stride - g.pad;
    } else {  // A_CONV_DGRAD: m = (n,h,w)
        const ConvGeom& g = a.g;
        int hw = g.H * g.W; c.n = m / hw; int r = m - c.n * hw; int h = r / g.W, w = r - h * g.W;
        c.y0 = h + g.pad; c.x0 = w + g.pad;
    }
}

template <int AM>
__device__ __forceinline__ float4 a_row_fetch(const KArgs& a, const ARowCtx<AM>& c, int k, int kend) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!c.ok || k >= kend) return v;
    if (AM == A_ROW) {
        const float* p = a.A + c.base + k;
        if (a.avec) return *reinterpret_cast<const float4*>(p);
        v.x = p[0];
        if (k + 1 < kend) v.y = p[1];
        if (k + 2 < kend) v.z = p[2];
        if (k + 3 < kend) v.w = p[3];
        return v;
    } else if (AM == A_CONV_FWD) {
        const ConvGeom& g = a.g;
        int rs = k / g.C, ch = k - rs * g.C; int r = rs / g.S, s = rs - r * g.S;
        int y = c.y0 + r, x = c.x0 + s;
        if ((unsigned)y >= (unsigned)g.H || (unsigned)x >= (unsigned)g.W) return v;
        return *reinterpret_cast<const float4*>(a.A + (((long)c.n * g.H + y) * g.W + x) * g.C + ch);
    } else {
        const ConvGeom& g = a.g;
        int rs = k / g.K, ko = k - rs * g.K; int r = rs / g.S, s = rs - r * g.S;
        int ty = c.y0 - r, tx = c.x0 - s;
        if (ty < 0 || tx < 0) return v;
        int p = ty / g.stride, q = tx / g.stride;
        if (p * g.stride != ty || q * g.stride != tx || p >= g.P || q >= g.Q) return v;
        return *reinterpret_cast<const float4*>(a.A + (((long)c.n * g.P + p) * g.Q + q) * g.K + ko);
    }
}

// k-major dense operand (used for A_KMAJOR and B_KMAJOR): element (k, j..j+3), ld = leading dim
__device__ __forceinline__ float4 kmajor_fetch(const float* base, long ld, int k, int kend, int j, int J, int vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k >= kend || j >= J) return v;
    const float* p = base + (long)k * ld + j;
    if (vec) return *reinterpret_cast<const float4*>(p);
    v.x = p[0];
    if (j + 1 < J) v.y = p[1];
    if (j + 2 < J) v.z = p[2];
    if (j + 3 < J) v.w = p[3];
    return v;
}

// B row mode: B[n][k]
__device__ __forceinline__ float4 b_row_fetch(const KArgs& a, int n, int k, int kend) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n >= a.N || k >= kend) return v;
    const float* p = a.B + (long)n * a.ldb + k;
    if (a.bvec) return *reinterpret_cast<const float4*>(p);
    v.x = p[0];
    if (k + 1 < kend) v.y = p[1];
    if (k + 2 < kend) v.z = p[2];
    if (k + 3 < kend) v.w = p[3];
    return v;
}

struct BColCtx { int r, s, ch; bool ok; };

template <int BMo>
__device__ __forceinline__ float4 b_k_fetch(const KArgs& a, const BColCtx& c, int n, int k, int kend) {
    if (BMo == B_KMAJOR) return kmajor_fetch(a.B, a.ldb, k, kend, n, a.N, a.bvec);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!c.ok || k >= kend) return v;
    const ConvGeom& g = a.g;
    if (BMo == B_CONV_WGRAD) {       // k = output pixel (img,p,q); n = (r,s,ch)
        int pq = g.P * g.Q; int img = k / pq; int rem = k - img * pq; int p = rem / g.Q, q = rem - p * g.Q;
        int y = p * g.stride - g.pad + c.r, x = q * g.stride - g.pad + c.s;
        if ((unsigned)y >= (unsigned)g.H || (unsigned)x >= (unsigned)g.W) return v;
        return *reinterpret_cast<const float4*>(a.B + (((long)img * g.H + y) * g.W + x) * g.C + c.ch);
    } else {                          // B_CONV_DGRAD_W: k = (r,s,ko); n = ch
        int rs = k / g.K, ko = k - rs * g.K; int r = rs / g.S, s = rs - r * g.S;
        return *reinterpret_cast<const float4*>(a.B + (((long)ko * g.R + r) * g.S + s) * g.C + n);
    }
}

template <int BM, int BN, int AM, int BMo>
__global__ __launch_bounds__(NTHREADS) void gemm_f32_kernel(KArgs a) {
    constexpr bool AK = (AM == A_KMAJOR);
    constexpr bool BKM = (BMo != B_ROW);
    constexpr int LDA_R = BK + 1, LDA_K = BM + 4;
    constexpr int LDB_R = BK + 1, LDB_K = BN + 4;
    constexpr int A_ELEMS = AK ? BK * LDA_K : BM * LDA_R;
    constexpr int B_ELEMS = BKM ? BK * LDB_K : BN * LDB_R;
    constexpr int NVA = BM * BK / 4 / NTHREADS;
    constexpr int NVB = BN * BK / 4 / NTHREADS;
    constexpr int TM = BM / 64, TN = BN / 64;
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_ELEMS + B_ELEMS)];
    constexpr int STAGE = A_ELEMS + B_ELEMS;     // one double-buffer stage: [A tile | B tile]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bm = blockIdx.y * BM, bn = blockIdx.x * BN;
    const int kbeg = blockIdx.z * a.kchunk;
    const int kend = min(a.K, kbeg + a.kchunk);
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
    const int li = lane & 31, lh = lane >> 5;

    // per-thread fixed part of the gathers
    ARowCtx<AM> actx[NVA];
    int a_i0[NVA], a_i1[NVA];      // row-like: (row, kq*4) ; k-major: (kk, m)
    BColCtx bctx[NVB];
    int b_i0[NVB], b_i1[NVB];      // row: (n_local, kq*4) ; k-major: (kk, n)
#pragma unroll
    for (int j = 0; j < NVA; ++j) {
        int v = tid + j * NTHREADS;
        if (!AK) { a_i0[j] = v >> 2; a_i1[j] = (v & 3) * 4; a_row_setup<AM>(a, bm + a_i0[j], actx[j]); }
        else { a_i0[j] = v / (BM / 4); a_i1[j] = (v % (BM / 4)) * 4; }
    }
#pragma unroll
    for (int j = 0; j < NVB; ++j) {
        int v = tid + j * NTHREADS;
        if (!BKM) { b_i0[j] = v >> 2; b_i1[j] = (v & 3) * 4; }
        else {
            b_i0[j] = v / (BN / 4); b_i1[j] = (v % (BN / 4)) * 4;
            int n = bn + b_i1[j];
            bctx[j].ok = n < a.N; bctx[j].r = bctx[j].s = bctx[j].ch = 0;
            if (BMo == B_CONV_WGRAD && bctx[j].ok) {
                int rs = n / a.g.C; bctx[j].ch = n - rs * a.g.C; bctx[j].r = rs / a.g.S; bctx[j].s = rs - bctx[j].r * a.g.S;
            }
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[NVA], rb[NVB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < NVA; ++j) {
            if (!AK) ra[j] = a_row_fetch<AM>(a, actx[j], k0 + a_i1[j], kend);
            else ra[j] = kmajor_fetch(a.A, a.lda, k0 + a_i0[j], kend, bm + a_i1[j], a.M, a.avec);
        }
#pragma unroll
        for (int j = 0; j < NVB; ++j) {
            if (!BKM) rb[j] = b_row_fetch(a, bn + b_i0[j], k0 + b_i1[j], kend);
            else rb[j] = b_k_fetch<BMo>(a, bctx[j], bn + b_i1[j], k0 + b_i0[j], kend);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NVA; ++j) {
            if (!AK) { float* p = smem + buf * STAGE + a_i0[j] * LDA_R + a_i1[j]; p[0] = ra[j].x; p[1] = ra[j].y; p[2] = ra[j].z; p[3] = ra[j].w; }
            else *reinterpret_cast<float4*>(smem + buf * STAGE + a_i0[j] * LDA_K + a_i1[j]) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < NVB; ++j) {
            if (!BKM) { float* p = smem + buf * STAGE + A_ELEMS + b_i0[j] * LDB_R + b_i1[j]; p[0] = rb[j].x; p[1] = rb[j].y; p[2] = rb[j].z; p[3] = rb[j].w; }
            else *reinterpret_cast<float4*>(smem + buf * STAGE + A_ELEMS + b_i0[j] * LDB_K + b_i1[j]) = rb[j];
        }
    };

    int cur = 0;
    if (kbeg < kend) {
        fetch(kbeg);
        stash(0);
        __syncthreads();
        for (int k0 = kbeg; k0 < kend; k0 += BK) {
            const bool more = (k0 + BK) < kend;
            if (more) fetch(k0 + BK);
            const float* as = smem + cur * STAGE;
            const float* bs = as + A_ELEMS;
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                float av[TM], bv[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    av[i] = AK ? as[(kk + lh) * LDA_K + wm + i * 32 + li] : as[(wm + i * 32 + li) * LDA_R + kk + lh];
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bv[j] = BKM ? bs[(kk + lh) * LDB_K + wn + j * 32 + li] : bs[(wn + j * 32 + li) * LDB_R + kk + lh];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
            if (more) stash(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }

    // epilogue: C/D map of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = bm + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                int col = bn + wn + j * 32 + li;
                if (row < a.M && col < a.N) {
                    if (a.nsplit > 1) a.slab[((long)blockIdx.z * a.M + row) * a.N + col] = acc[i][j][r];
                    else store_out(a, row, col, acc[i][j][r]);
                }
            }
}

__global__ void splitk_reduce_kernel(KArgs a) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)a.M * a.N;
    if (idx >= total) return;
    float v = 0.f;
    for (int z = 0; z < a.nsplit; ++z) v += a.slab[(long)z * total + idx];   // fixed order: deterministic
    store_out(a, (int)(idx / a.N), (int)(idx % a.N), v);
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int pick_split(int M, int N, int K, int BMt, int BNt) {
    long blocks = (long)cdiv(M, BMt) * cdiv(N, BNt);
    if (blocks >= 256 || K < 8 * BK * 2) return 1;
    int want = (int)((512 + blocks - 1) / blocks);
    int maxs = K / (8 * BK);
    int s = want < maxs ? want : maxs;
    return s < 1 ? 1 : (s > 64 ? 64 : s);
}

static void pick_tile(int M, int N, int& BMt, int& BNt) {
    long big = (long)cdiv(M, 128) * cdiv(N, 128);
    if (big >= 384 && M >= 128 && N >= 128) { BMt = BNt = 128; } else { BMt = BNt = 64; }
}

size_t gemm_slab_bytes(int M, int N, int K) {
    int BMt, BNt; pick_tile(M, N, BMt, BNt);
    int s = pick_split(M, N, K, BMt, BNt);
    return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}

static const char* mode_name(int am, int bm) {
    if (am == A_CONV_FWD) return "conv_fwd";
    if (am == A_CONV_DGRAD) return "conv_dgrad";
    if (bm == B_CONV_WGRAD) return "conv_wgrad";
    if (am == A_ROW && bm == B_ROW) return "nt";
    if (am == A_ROW && bm == B_KMAJOR) return "nn";
    return "tn";
}

template <int BM, int BN, int AM, int BMo>
static int run(const KArgs& k, hipStream_t st) {
    dim3 grid(cdiv(k.N, BN), cdiv(k.M, BM), k.nsplit);
    char pname[96];
    if (profile_enabled()) snprintf(pname, sizeof pname, "gemm_f32_%s_%dx%d", mode_name(AM, BMo), BM, BN);
    ProfScope prof(pname, 2.0 * k.M * k.N * k.K, 4.0 * ((double)k.M * k.K + (double)k.N * k.K + (double)k.M * k.N), st);
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, AM, BMo>), grid, dim3(NTHREADS), 0, st, k);
    SAT_TRY(launch_ok("gemm_f32_kernel"));
    if (k.nsplit > 1) {
        long total = (long)k.M * k.N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, k);
        SAT_TRY(launch_ok("splitk_reduce_kernel"));
    }
    return SAT_OK;
}

template <int AM, int BMo>
static int run_tiles(const KArgs& k, int BMt, hipStream_t st) {
    if (BMt == 128) return run<128, 128, AM, BMo>(k, st);
    return run<64, 64, AM, BMo>(k, st);
}

int launch_gemm(const GemmArgs& g, hipStream_t st) {
    SAT_REQUIRE(g.M >= 0 && g.N >= 0 && g.K >= 0, "gemm: negative dims");
    if (g.M == 0 || g.N == 0) return SAT_OK;
    SAT_REQUIRE(g.A && g.B && g.C, "gemm: null operand");
    if (g.bf16_mfma && gemm_bf16_eligible(g)) return launch_gemm_bf16(g, st);
    SAT_REQUIRE(!g.a_bf16 && !g.b_bf16 && !g.c_bf16, "gemm: bf16 operands need the bf16 MFMA kernel (16-byte gatherable shapes)");
    SAT_REQUIRE(!g.add_src, "gemm: add_src is a bf16-storage feature");
    KArgs k;
    k.A = (const float*)g.A; k.lda = g.lda; k.a_rows = g.a_rows; k.B = (const float*)g.B; k.ldb = g.ldb;
    k.C = (float*)g.C; k.ldc = g.ldc; k.c_rows = g.c_rows; k.M = g.M; k.N = g.N; k.K = g.K;
    k.accumulate = g.accumulate; k.epi = g.epi; k.bias = g.bias; k.e0 = g.e0; k.lde0 = g.lde0;
    k.c0 = g.c0; k.c1 = g.c1; k.g = g.g; k.slab = g.slab;
    if ((g.epi == EPI_BIAS || g.epi == EPI_BIAS_RELU) && !g.bias) return fail(SAT_EINVAL, "gemm: bias epilogue without bias");
    if ((g.epi == EPI_ADD_TANH || g.epi == EPI_MUL_DTANH) && !g.e0) return fail(SAT_EINVAL, "gemm: epilogue operand missing");
    const bool conv_a = (g.amode == A_CONV_FWD || g.amode == A_CONV_DGRAD);
    const bool conv_b = (g.bmode == B_CONV_WGRAD || g.bmode == B_CONV_DGRAD_W);
    if (conv_a || conv_b) {
        SAT_REQUIRE(g.g.C % 4 == 0 && g.g.K % 4 == 0, "conv gemm: C (%d) and K (%d) must be multiples of 4", g.g.C, g.g.K);
        SAT_REQUIRE(aligned16(g.A) && aligned16(g.B), "conv gemm: operands must be 16-byte aligned");
        SAT_REQUIRE(g.g.stride >= 1 && g.g.P > 0 && g.g.Q > 0, "conv gemm: bad geometry");
    }
    // 16-byte vector eligibility
    if (g.amode == A_ROW) k.avec = (g.K % 4 == 0) && (g.lda % 4 == 0) && aligned16(g.A);
    else if (g.amode == A_KMAJOR) k.avec = (g.M % 4 == 0) && (g.lda % 4 == 0) && aligned16(g.A);
    else k.avec = 1;
    if (g.bmode == B_ROW) k.bvec = (g.K % 4 == 0) && (g.ldb % 4 == 0) && aligned16(g.B);
    else if (g.bmode == B_KMAJOR) k.bvec = (g.N % 4 == 0) && (g.ldb % 4 == 0) && aligned16(g.B);
    else k.bvec = 1;

    int BMt, BNt; pick_tile(g.M, g.N, BMt, BNt);
    int ns = 1;
    if (g.slab && g.K > 0) {
        ns = pick_split(g.M, g.N, g.K, BMt, BNt);
        while (ns > 1 && (long)ns * g.M * g.N > g.slab_elems) --ns;
    }
    int ktiles = cdiv(g.K, BK);
    int per = cdiv(ktiles, ns);
    k.kchunk = per * BK;
    k.nsplit = (g.K == 0) ? 1 : cdiv(ktiles, per);
    if (k.kchunk == 0) k.kchunk = BK;

#define SAT_GEMM_CASE(AMV, BMV) \
    if (g.amode == AMV && g.bmode == BMV) return run_tiles<AMV, BMV>(k, BMt, st);
    SAT_GEMM_CASE(A_ROW, B_ROW)
    SAT_GEMM_CASE(A_ROW, B_KMAJOR)
    SAT_GEMM_CASE(A_KMAJOR, B_KMAJOR)
    SAT_GEMM_CASE(A_CONV_FWD, B_ROW)
    SAT_GEMM_CASE(A_CONV_DGRAD, B_CONV_DGRAD_W)
    SAT_GEMM_CASE(A_KMAJOR, B_CONV_WGRAD)
#undef SAT_GEMM_CASE
    return fail(SAT_EUNSUPPORTED, "gemm: operand mode pair (%d,%d) not built", g.amode, g.bmode);
}

}  // namespace sat
