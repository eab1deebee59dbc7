// bf16 MFMA GEMM / implicit GEMM whose operand tiles go HBM -> LDS through the load unit
// (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass), for bf16 operands in HBM.
//
// The register-staged kernel (gemm_bf16.hip) pays ~13 cycles of LDS-store data path per ds_write_b128 wave
// instruction: 32 KB of operands per 128x128x64 step cost about as long as the 16 MFMAs that consume them.
// Here a wave instruction deposits 64 x 16 B contiguously at a wave-uniform LDS address while every lane
// supplies its own SOURCE address, so the conflict-free LDS images are obtained by permuting the sources:
//
//   k-contiguous operand (rows x 64 k, 128-byte rows): 16-byte slot s of row r is stored at slot s ^ ((r >> 1) & 7);
//       a ds_read_b128 fragment read (32 rows x one slot per 16-lane group) then covers all 64 banks once.
//   k-major operand (64 k x W cols, W = 64 | 128):     the 64-byte quarter q of k-row k is stored at quarter
//       q ^ (W == 128 ? k & 3 : (k >> 1) & 1); the four k-rows of a ds_read_b64_tr_b16 half-wave hit disjoint quarters.
//
// The copies are buffer loads (buffer_load_dwordx4 ... offen lds): a 32-bit byte offset per lane plus a wave-uniform
// SGPR offset, with the hardware range check standing in for the zero padding of the convolutions (an offset of
// 0x80000000 is out of range of the 2 GiB descriptor: the lane deposits zeros).  Address generation is the cost that
// matters at one or two waves per SIMD (the 64-bit gather arithmetic of a conv tile used to take longer than its 16
// MFMAs), so everything that can be is hoisted out of the k loop:
//   dense operands      lane offset fixed, the k-tile advances the SGPR offset only          (0 VALU per piece)
//   conv activations    lane offset of the pixel at tap 0 + a uniform tap offset; validity of the 9 taps is a
//                       bit mask per lane computed once per workgroup                        (~4 VALU per piece)
//   wgrad activations   pixel offset and tap mask come from a 64-entry LDS table per k-tile  (~5 VALU per piece)
// Rows past M / N read clamped (in-range) addresses: their products are never stored.  The forms need the
// reduction to be a multiple of 64 and, for convolutions, channels % 64 == 0 (a k-tile then has a single filter
// tap, tracked incrementally); everything else stays on gemm_bf16.hip.
#include <stdlib.h>

#include <initializer_list>

#pragma once
#include "gemm_bf16_common.h"
#include "profile.h"

#ifndef SAT_GLDS_ABLATE
#define SAT_GLDS_ABLATE 0        // timing ablations, dev builds only (results are garbage): see tools/ablate_gemm.py
#endif

namespace sat {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

// llvm.amdgcn.raw.buffer.load.lds: buffer_load_dwordx4 voffset, rsrc, soffset offen lds (M0 = wave-uniform LDS address)
__device__ void raw_buffer_load_lds(i32x4 rsrc, lptr_t lds, int size, int voffset, int soffset, int offset, int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");

constexpr int OOB = (int)0x80000000;      // beyond num_records of the descriptors below: the lane loads zeros

__device__ __forceinline__ i32x4 make_rsrc(const void* base) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    i32x4 r;
    r[0] = (int)(unsigned)b; r[1] = (int)(unsigned)((b >> 32) & 0xFFFF);      // stride 0: raw buffer
    r[2] = OOB;                                                                // num_records = 2 GiB
    r[3] = 0x00020000;                                                         // gfx9 raw-buffer word 3 (DATA_FORMAT = 32 bit)
    return r;
}
__device__ __forceinline__ void dma16(const i32x4& rsrc, int voff, int soff, __bf16* lds_wave_base) {
    raw_buffer_load_lds(rsrc, (lptr_t)lds_wave_base, 16, voff, soff, 0, 0);
}

// bits [lo, hi) set
__device__ __forceinline__ unsigned bit_range(int lo, int hi) {
    if (lo < 0) lo = 0;
    if (hi > 32) hi = 32;
    if (hi <= lo) return 0u;
    const unsigned upper = (hi >= 32) ? 0xFFFFFFFFu : ((1u << hi) - 1u);
    return upper & ~((1u << lo) - 1u);
}
// tap mask of a pixel: bit (i * ncol + j) set when row tap i lies in [rlo, rhi) and column tap j in [clo, chi)
__device__ __forceinline__ unsigned tap_mask(int rlo, int rhi, int clo, int chi, int nrow, int ncol) {
    const unsigned cm = bit_range(clo, chi < ncol ? chi : ncol);
    unsigned m = 0;
    if (rlo < 0) rlo = 0;
    if (rhi > nrow) rhi = nrow;
    for (int i = rlo; i < rhi; ++i) m |= cm << (i * ncol);
    return m;
}

// timing ablations: the value stays live although nothing consumes it (a "v" operand does not parse in the host pass, where the
// kernel body would then be dropped without a diagnostic)
template <typename T> __device__ __forceinline__ void keep_alive(const T& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"v"(v));
#endif
}
template <typename T, typename U> __device__ __forceinline__ void keep_alive(const T& v, const U& w) { keep_alive(v); keep_alive(w); }

struct PixEnt { int off; unsigned mask; };   // wgrad: byte offset of the input pixel under tap (0,0) of an output pixel, valid taps

// WR x WC waves per workgroup; wave (wr, wc) owns the (BM / WR) x (BN / WC) block of the tile.  2 x 2 waves on 128x128 (and 128x64,
// 64x64) is the original form; 4 x 2 waves on 256x128 and 256x256 tiles (one workgroup per CU) halve the L2 -> LDS traffic per
// product and, on 256x256, read 6 LDS fragments per 8 MFMAs instead of 4 per 4.
#ifndef SAT_GLDS_BN_PREFETCH
#define SAT_GLDS_BN_PREFETCH 1
#endif
template <int BM, int BN, int WR, int WC, int AM, int BMo, typename TC, bool APF = false>
__global__ __launch_bounds__(64 * WR * WC, (BM == 128 && BN == 64 && sizeof(TC) == 2) ? ((AM == A_ROW && BMo == B_ROW) ? 6 : 4) : ((BM == 128 && BN == 128 && sizeof(TC) == 4) ? 3 : 1)) void gemm_glds_kernel(BArgs a) {          // 128 x 64, bf16 result: four resident workgroups per CU (the 3x3 data-gradient form sat at 97 VGPRs, one over)
    constexpr int KB = 64;
    constexpr int NW = WR * WC;
    constexpr bool AK = (AM == A_KMAJOR);
    constexpr bool BKM = (BMo != B_ROW);
    constexpr bool ACONV = (AM == A_CONV_FWD || AM == A_CONV_DGRAD);
    constexpr int A_EL = BM * KB, B_EL = BN * KB;
    constexpr int STAGE = A_EL + B_EL;
    constexpr int NA = BM / (8 * NW), NB = BN / (8 * NW);          // LDS-DMA wave-instructions per wave and k-tile
    constexpr int TM = BM / WR / 32, TN = BN / WC / 32;
    static_assert(NA >= 1 && NB >= 1 && TM >= 1 && TN >= 1, "tile too small for this wave grid");
    constexpr bool WG = (BMo == B_CONV_WGRAD);
    constexpr int G = NA + NB;                          // (vmcnt bookkeeping)
    constexpr int SMAX = 4;                             // deepest ring
    extern __shared__ __attribute__((aligned(1024))) __bf16 smem[];      // nstage x (A tile | B tile)
    __shared__ PixEnt ptab[SMAX][WG ? KB : 1];
    const int S = a.nstage;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, by, bz;
    xcd_tile(bx, by, bz, a.y_fastest);
    const int bm = by * BM, bn = bx * BN;
    const int kbeg = bz * a.kchunk;
    const int kend = min(a.K, kbeg + a.kchunk);
    const int wrow = wave / WC, wm = wrow * (BM / WR), wn = (wave % WC) * (BN / WC);
    const int li = lane & 31, lh = lane >> 5;
    const int tg = lane >> 4, ti = lane & 15;
    const int t_h = tg >> 1, t_mh = tg & 1, t_q = ti >> 2, t_p = ti & 3;
    const ConvGeom& g = a.g;
    const i32x4 rA = make_rsrc(a.A), rB = make_rsrc(a.B);

    // taps of the reduction in k order (conv forms): nrow x ncol of them; dgrad parity classes walk their own taps
    const int nrow = (AM == A_CONV_DGRAD && g.cls) ? g.rc : g.R;
    const int ncol = (AM == A_CONV_DGRAD && g.cls) ? g.sc : g.S;

    // ---- per-lane byte offsets (fixed over the k loop) and tap masks
    int a_off0[NA]; unsigned a_mask[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int chunk = j * NW + wave;
        a_mask[j] = 0;
        if (!AK) {
            const int row = chunk * 8 + (lane >> 3), ks = (lane & 7) ^ ((row >> 1) & 7);
            int m = bm + row;
            const bool ok = m < a.M;
            if (!ok) m = a.M - 1;
            if (AM == A_ROW) a_off0[j] = (int)(((long)m * a.lda + ks * 8) * 2);
            else if (AM == A_CONV_FWD) {
                const int pq = g.P * g.Q; const int n = m / pq; const int r = m - n * pq; const int p = r / g.Q, q = r - p * g.Q;
                const int y0 = p * g.stride - g.pad, x0 = q * g.stride - g.pad;
                a_off0[j] = (((n * g.H + y0) * g.W + x0) * g.C + ks * 8) * 2;
                if (ok) a_mask[j] = tap_mask(-y0, g.H - y0, -x0, g.W - x0, nrow, ncol);     // y0 + r in [0, H), x0 + s in [0, W)
            } else {
                // input pixel (n, h, w); tap i reaches output row pb - i (stride 1: pb = h + pad; parity class: see gemm.h)
                int n, pb, qb;
                if (g.cls) {
                    const int hw = g.Hc * g.Wc; n = m / hw; const int r = m - n * hw; const int hc = r / g.Wc, wc = r - hc * g.Wc;
                    pb = hc + ((g.ph + g.pad - g.r0) >> 1); qb = wc + ((g.pw + g.pad - g.s0) >> 1);
                } else {
                    const int hw = g.H * g.W; n = m / hw; const int r = m - n * hw; const int h = r / g.W, w = r - h * g.W;
                    pb = h + g.pad; qb = w + g.pad;
                }
                a_off0[j] = (((n * g.P + pb) * g.Q + qb) * g.K + ks * 8) * 2;
                if (ok) a_mask[j] = tap_mask(pb - g.P + 1, pb + 1, qb - g.Q + 1, qb + 1, nrow, ncol);    // 0 <= pb - i < P
            }
        } else {
            constexpr int SP = BM / 8;
            const int krow = chunk * (64 / SP) + lane / SP, sl = lane % SP;
            const int lq = (sl >> 2) ^ (BM >= 128 ? (krow & 3) : ((krow >> 1) & 1));
            int col = bm + (lq * 4 + (sl & 3)) * 8;
            if (col >= a.M) col = a.M - 8;
            a_off0[j] = (int)(((long)krow * a.lda + col) * 2);
        }
    }
    int b_off0[NB]; unsigned b_bit[NB]; int b_krow[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int chunk = j * NW + wave;
        b_bit[j] = 0; b_krow[j] = 0;
        if (!BKM) {
            const int row = chunk * 8 + (lane >> 3), ks = (lane & 7) ^ ((row >> 1) & 7);
            int n = bn + row;
            if (n >= a.N) n = a.N - 1;
            b_off0[j] = (int)(((long)n * a.ldb + ks * 8) * 2);
        } else {
            constexpr int SP = BN / 8;
            const int krow = chunk * (64 / SP) + lane / SP, sl = lane % SP;
            const int lq = (sl >> 2) ^ (BN >= 128 ? (krow & 3) : ((krow >> 1) & 1));
            int n = bn + (lq * 4 + (sl & 3)) * 8;
            const bool ok = n < a.N;
            if (!ok) n = a.N - 8;
            b_krow[j] = krow;
            if (BMo == B_KMAJOR) b_off0[j] = (int)(((long)krow * a.ldb + n) * 2);
            else if (BMo == B_CONV_DGRAD_W) b_off0[j] = (krow * g.R * g.S * g.C + n) * 2;       // filter ko = c0 + krow of the tile's tap
            else {   // wgrad: column n = (r, s, channel) of the filter gradient; the pixel comes from the table
                const int rs = n / g.C, ch = n - rs * g.C, r = rs / g.S, sx = rs - r * g.S;
                b_off0[j] = ((r * g.W + sx) * g.C + ch) * 2;
                b_bit[j] = ok ? (1u << rs) : 0u;
            }
        }
    }

    // ---- wave-uniform state of the NEXT tile to load
    int sA = 0, sB = 0;           // SGPR byte offsets
    int tapA = 0;                 // conv activations: byte offset of the tile's tap + channel block (added per lane)
    unsigned tap_bit = 1;         // conv activations: bit of the tile's tap in the lane masks
    int c0 = 0, ti_r = 0, ti_c = 0;
    const int chans = (AM == A_CONV_FWD) ? g.C : g.K;
    auto set_tap = [&]() {
        tap_bit = 1u << (ti_r * ncol + ti_c);
        if (AM == A_CONV_FWD) tapA = ((ti_r * g.W + ti_c) * g.C + c0) * 2;
        else if (AM == A_CONV_DGRAD) {
            tapA = (c0 - (ti_r * g.Q + ti_c) * g.K) * 2;
            const int r = g.cls ? g.r0 + 2 * ti_r : ti_r, sx = g.cls ? g.s0 + 2 * ti_c : ti_c;
            sB = ((c0 * g.R + r) * g.S + sx) * g.C * 2;
        }
    };
    // Workgroups that share operand tiles (same row / column / k-split) would walk the same few KB of the reduction in
    // lockstep and queue on a handful of L2 channels (measured on the weight gradients: 3 TB/s of L2 -> LDS traffic at
    // an 87 % hit rate).  Each workgroup therefore starts its k loop at its own rotation of the tile sequence and wraps
    // around: same tiles, same result up to the (fixed) summation order, requests spread over the whole k window.
    const int T = (kend - kbeg + KB - 1) / KB;
    int tcur = (T > 0 && a.rotate) ? (int)((unsigned)(bx * 5 + by * 3 + bz) % (unsigned)(T < a.rotate ? T : a.rotate)) : 0;
    const int rot = tcur;
    auto seek = [&](int tile) {          // wave-uniform state for k-tile `tile` of this workgroup's chunk
        const int k0 = kbeg + tile * KB;
        if (ACONV) { const int rs = k0 / chans; c0 = k0 - rs * chans; ti_r = rs / ncol; ti_c = rs - ti_r * ncol; set_tap(); }
        if (AM == A_ROW) sA = k0 * 2;
        if (AK) sA = (int)((long)k0 * a.lda * 2);
        if (BMo == B_ROW) sB = k0 * 2;
        if (BMo == B_KMAJOR) sB = (int)((long)k0 * a.ldb * 2);
    };
    seek(tcur);
    auto advance_tile = [&]() {
        if (++tcur == T) { tcur = 0; seek(0); return; }
        if (AM == A_ROW) sA += KB * 2;
        if (AK) sA += (int)(KB * a.lda * 2);
        if (BMo == B_ROW) sB += KB * 2;
        if (BMo == B_KMAJOR) sB += (int)(KB * a.ldb * 2);
        if (ACONV) {
            c0 += KB;
            if (c0 >= chans) { c0 = 0; if (++ti_c == ncol) { ti_c = 0; ++ti_r; } }
            set_tap();
        }
    };

    // wgrad pixel table of one k-tile: the waves take turns (tile t is decoded by wave t mod NW)
    auto fill_tab = [&](int slot, int t) {
        if (WG && wave == (t & (NW - 1))) {
            PixEnt e; e.off = 0; e.mask = 0;
            int tt = t + rot; if (tt >= T) tt -= T;          // sequence position -> k-tile (positions past the end are never loaded)
            const int k = kbeg + tt * KB + lane;
            if (tt < T && k < kend) {
                const int pq = g.P * g.Q; const int img = k / pq; const int rem = k - img * pq; const int p = rem / g.Q, q = rem - p * g.Q;
                const int y0 = p * g.stride - g.pad, x0 = q * g.stride - g.pad;
                e.off = ((img * g.H + y0) * g.W + x0) * g.C * 2;
                e.mask = tap_mask(-y0, g.H - y0, -x0, g.W - x0, g.R, g.S);
            }
            ptab[slot][lane] = e;
        }
    };

    // one LDS-DMA wave-instruction of the next tile: A pieces jj < NA, then B pieces
    constexpr int abl = SAT_GLDS_ABLATE;          // dev builds (make ABLATE=bits): 1 = no operand loads, 2 = no MFMAs, 4 = no fragment reads, 8 = no result stores
    auto issue_piece = [&](int buf, int tb, int jj) {
        if constexpr (abl & 1) return;
        __bf16* as = smem + buf * STAGE; __bf16* bs = as + A_EL;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            if (j != jj) continue;
            if (ACONV) dma16(rA, (a_mask[j] & tap_bit) ? a_off0[j] + tapA : OOB, 0, as + (j * NW + wave) * 512);
            else dma16(rA, a_off0[j], sA, as + (j * NW + wave) * 512);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (j + NA != jj) continue;
            if (WG) {
                const PixEnt e = ptab[tb][b_krow[j]];
                dma16(rB, (e.mask & b_bit[j]) ? e.off + b_off0[j] : OOB, 0, bs + (j * NW + wave) * 512);
            } else dma16(rB, b_off0[j], sB, bs + (j * NW + wave) * 512);
        }
    };
    auto issue = [&](int buf, int tb) {
#pragma unroll
        for (int jj = 0; jj < G; ++jj) issue_piece(buf, tb, jj);
        advance_tile();
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // BatchNorm-backward statistics (data-gradient launches, bf16 result): the tile's slice of the BatchNorm input and its sign bits are
    // fetched NOW, so that the loads fly under the whole k loop instead of in front of the write phase (they were 3 us of exposed
    // latency per tile there).  In-order completion keeps the counted waits below valid: these loads are older than every LDS-DMA.
    constexpr bool BNS = (AM == A_CONV_DGRAD) || (AM == A_ROW && BMo == B_KMAJOR);          // the data-gradient forms: the only ones that carry bn_x
    constexpr bool BPF = SAT_GLDS_BN_PREFETCH;
    BnAcc<BM, BN, 64 * NW, BNS, BPF> bnacc(a, bn, tid);
    if constexpr (sizeof(TC) == 2) { bnacc.prefetch(a, bm, bn, tid); __builtin_amdgcn_sched_barrier(0); }      // issued here, not sunk to their use

    // ---- per-lane fragment read offsets (elements)
    int a_off[TM], a_sw[TM], b_off[TN], b_sw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if (!AK) { const int row = wm + i * 32 + li; a_off[i] = row * KB; a_sw[i] = (row >> 1) & 7; }
        else {
            const int col = wm + i * 32 + 16 * t_mh + 4 * t_p, kr = 8 * t_h + t_q;
            const int sw = (BM >= 128) ? (kr & 3) : ((kr >> 1) & 1);
            a_off[i] = kr * BM + (((col >> 5) ^ sw) << 5) + (col & 31); a_sw[i] = 0;
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        if (!BKM) { const int row = wn + j * 32 + li; b_off[j] = row * KB; b_sw[j] = (row >> 1) & 7; }
        else {
            const int col = wn + j * 32 + 16 * t_mh + 4 * t_p, kr = 8 * t_h + t_q;
            const int sw = (BN >= 128) ? (kr & 3) : ((kr >> 1) & 1);
            b_off[j] = kr * BN + (((col >> 5) ^ sw) << 5) + (col & 31); b_sw[j] = 0;
        }
    }

    // ---- k loop: ring of S stages, tiles it+1 .. it+S-1 in flight while tile `it` is multiplied.
    // Per iteration: wait until this wave's pieces of tile `it` have landed (counted vmcnt: the newer tiles stay in
    // flight), barrier (every wave's pieces landed; every wave is done reading the stage about to be refilled),
    // then issue tile it+S-1 piecewise between the MFMA groups of tile `it`.
    if (kbeg < kend) {
        if (WG) {
            for (int t = 0; t < S; ++t) fill_tab(t, t);
            __syncthreads();
        }
        const int pre = (S == 1) ? 1 : S - 1;          // a one-stage launch has a single k-tile: load it, multiply it
        for (int t = 0; t < pre && t < T; ++t) issue(t, t);
        int cur = 0, nxt = S - 1;                      // stage of tile it / of tile it + S - 1
        for (int it = 0; it < T; ++it) {
            const int ahead = (S == 1) ? 0 : min(T - 1 - it, S - 2);  // tiles newer than `it` already issued
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const bool more = S > 1 && it + S - 1 < T;
            const __bf16* as = smem + cur * STAGE;
            const __bf16* bs = as + A_EL;
            // fragments are double buffered in registers: the reads of k-step s+1 are in flight under the MFMAs of step s
            bf16x8 af[2][TM], bf[2][TN];
            auto load_frags = [&](int kk, bf16x8 (&fa)[TM], bf16x8 (&fb)[TN]) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if (!AK) fa[i] = *reinterpret_cast<const bf16x8*>(as + a_off[i] + ((((kk >> 3) + lh) ^ a_sw[i]) << 3));
                    else {
                        const __bf16* p = as + a_off[i] + kk * BM;
                        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
                        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p + 4 * BM));
                        fa[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (!BKM) fb[j] = *reinterpret_cast<const bf16x8*>(bs + b_off[j] + ((((kk >> 3) + lh) ^ b_sw[j]) << 3));
                    else {
                        const __bf16* p = bs + b_off[j] + kk * BN;
                        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
                        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p + 4 * BN));
                        fb[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
            };
            if (!(abl & 4) || it == 0) load_frags(0, af[0], bf[0]);
#pragma unroll
            for (int ks = 0; ks < KB / 16; ++ks) {
                if (ks + 1 < KB / 16 && (!(abl & 4) || it == 0)) load_frags((ks + 1) * 16, af[(ks + 1) & 1], bf[(ks + 1) & 1]);
                // this k-step's share of the next tile's LDS-DMA pieces, issued under the MFMAs
                if (more) {
#pragma unroll
                    for (int jj = (ks * G) / 4; jj < ((ks + 1) * G) / 4; ++jj) issue_piece(nxt, nxt, jj);          // G pieces spread over the four k-steps
                }
                if constexpr (!(abl & 2)) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) keep_alive(af[ks & 1][i], bf[ks & 1][j]);      // keep the fragment reads alive
                }
            }
            if (more) advance_tile();
            if (S == 1 && it + 1 < T) {            // one-stage ring with several k-tiles: refill the only stage once every wave has read it
                __syncthreads();
                if (WG) { fill_tab(0, it + 1); __syncthreads(); }
                issue(0, 0);
            }
            // pixel table of tile it + S goes where tile it's was (consumed S - 1 iterations ago)
            if (WG && S > 1) fill_tab(cur, it + S);
            cur = (cur + 1 == S) ? 0 : cur + 1;
            nxt = (nxt + 1 == S) ? 0 : nxt + 1;
        }
    }
    if constexpr (abl & 8) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) keep_alive(acc[i][j]);
        return;
    }
    if constexpr (WR == 2 && WC == 2) store_tile<BM, BN, TC, 2 * STAGE, APF, BNS, BPF>(a, acc, smem, bm, bn, bz, wm, wn, tid, lane, bnacc, sizeof(TC) == 2);
    else store_tile_w<BM, BN, WR, WC, TC, BNS, BPF>(a, acc, smem, bm, bn, bz, wm, wn, wrow, tid, lane, bnacc, sizeof(TC) == 2);      // the launch allocates at least the epilogue's staging size
}

static const char* gname(int am, int bm) {
    if (am == A_CONV_FWD) return "conv_fwd";
    if (am == A_CONV_DGRAD) return "conv_dgrad";
    if (bm == B_CONV_WGRAD) return "conv_wgrad";
    if (am == A_ROW && bm == B_ROW) return "nt";
    if (am == A_ROW && bm == B_KMAJOR) return "nn";
    return "tn";
}

template <int BM, int BN, int WR, int WC, int AM, int BMo, typename TC>
static int rung(const BArgs& k, hipStream_t st) {
    dim3 grid(cdiv(k.N, BN), cdiv(k.M, BM), k.nsplit);
    char pname[128];
    if (profile_enabled()) {
        static const bool shapes = getenv("SAT_PROFILE_SHAPES") != nullptr;      // dev: one profile line per problem shape
        if (shapes) snprintf(pname, sizeof pname, "gemm_glds_%s_%dx%d M%d N%d K%d z%d", gname(AM, BMo), BM, BN, k.M, k.N, k.K, k.nsplit);
        else snprintf(pname, sizeof pname, "gemm_glds_%s_%dx%d", gname(AM, BMo), BM, BN);
    }
    ProfScope prof(pname, 2.0 * k.M * k.N * k.K, 2.0 * k.M * k.K + 2.0 * k.N * k.K + (double)sizeof(TC) * k.M * k.N, st);
    size_t lds = (size_t)k.nstage * (BM + BN) * 64 * sizeof(__bf16);
    // the epilogue stages the result tile in the same memory: bf16 tile (+ statistics) or a wave row of the fp32 tile
    const size_t epi_lds = (WR == 2 && WC == 2) ? ((sizeof(TC) == 2) ? (size_t)BM * (BN + 8) * 2 + 4 * BN * 2 * 4 : (size_t)(BM / 2) * (BN + 4) * 4)
                                                : store_lds_bytes<BM, BN, WR, WC, TC>();
    if (lds < epi_lds) lds = epi_lds;
    constexpr size_t LDS_DYN_MAX = 160 * 1024 - 2560;       // the whole LDS less the kernel's static part (the weight-gradient pixel tables: 2 KiB)
    SAT_REQUIRE(lds <= LDS_DYN_MAX, "gemm_glds: %zu bytes of LDS for a %dx%d tile with %d stages", lds, BM, BN, k.nstage);
    static bool attr_set = false;        // per instantiation: allow the whole LDS (160 KiB of dynamic LDS)
    if (!attr_set) {
        SAT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<BM, BN, WR, WC, AM, BMo, TC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)LDS_DYN_MAX));
        attr_set = true;
    }
    // accumulating bf16 launches of the data-gradient forms (identity joins, projection sums): the instantiation whose epilogue has the old values in flight
    // MEASURED (tools/ab_step.py, C2 step): 22.52 ms with it, 22.20 ms without - the registers of the old values (214 VGPRs) take the third workgroup per CU
    // from these launches and that costs more than the serialised loads; with the prefetch inside the ONE instantiation every launch lost the third
    // workgroup (25.3 ms).  Compiled out unless SAT_GLDS_APF is defined.
#ifndef SAT_GLDS_APF
#define SAT_GLDS_APF 0
#endif
    constexpr bool HAS_APF = SAT_GLDS_APF && sizeof(TC) == 2 && WR == 2 && WC == 2 && BM == 128 && BM * (BN / 8) / 256 <= 8 &&
                             ((AM == A_ROW && BMo == B_KMAJOR) || AM == A_CONV_DGRAD);
    if constexpr (HAS_APF) {
        if (k.accumulate && k.acc_prefetch && k.wide_store && k.nsplit == 1) {
            static bool attr_apf = false;
            if (!attr_apf) {
                SAT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<BM, BN, WR, WC, AM, BMo, TC, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_DYN_MAX));
                attr_apf = true;
            }
            hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, WR, WC, AM, BMo, TC, true>), grid, dim3(64 * WR * WC), lds, st, k);
            return launch_ok("gemm_glds_kernel (accumulate)");
        }
    }
    hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, WR, WC, AM, BMo, TC>), grid, dim3(64 * WR * WC), lds, st, k);
    SAT_TRY(launch_ok("gemm_glds_kernel"));
    if (k.nsplit > 1) SAT_TRY(launch_splitk_reduce<TC>(k, st));
    return SAT_OK;
}


// one translation unit per tile form (gemm_glds_t*.hip) so that the forms compile in parallel
#define SAT_GLDS_INSTANTIATE(FN)                                                                  \
    template int FN<A_CONV_FWD, B_ROW, __bf16>(const BArgs&, hipStream_t);                        \
    template int FN<A_ROW, B_ROW, __bf16>(const BArgs&, hipStream_t);                             \
    template int FN<A_ROW, B_ROW, float>(const BArgs&, hipStream_t);                              \
    template int FN<A_CONV_DGRAD, B_CONV_DGRAD_W, __bf16>(const BArgs&, hipStream_t);             \
    template int FN<A_ROW, B_KMAJOR, __bf16>(const BArgs&, hipStream_t);                          \
    template int FN<A_ROW, B_KMAJOR, float>(const BArgs&, hipStream_t);                           \
    template int FN<A_KMAJOR, B_CONV_WGRAD, float>(const BArgs&, hipStream_t);                    \
    template int FN<A_KMAJOR, B_KMAJOR, float>(const BArgs&, hipStream_t);

}  // namespace sat
