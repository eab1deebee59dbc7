// Dispatcher of the direct-to-LDS GEMM / implicit-GEMM kernels (kernel template: gemm_glds_kernel.h; one translation unit per tile
// form: gemm_glds_t*.hip).
#include <stdlib.h>

#include "gemm_bf16_common.h"

namespace sat {

template <int AM, int BMo, typename TC> int glds_run_64(const BArgs& k, hipStream_t st);
template <int AM, int BMo, typename TC> int glds_run_128(const BArgs& k, hipStream_t st);
template <int AM, int BMo, typename TC> int glds_run_128x64(const BArgs& k, hipStream_t st);
#ifdef SAT_DEV_TILES      // `make DEV=1`: the 8-wave 256-row forms (measured slower inside the step) are kept out of the shipped library
template <int AM, int BMo, typename TC> int glds_run_256x128(const BArgs& k, hipStream_t st);
template <int AM, int BMo, typename TC> int glds_run_256(const BArgs& k, hipStream_t st);
#endif

// tile forms: 0 = 64x64, 1 = 128x128, 2 = 128x64 (4 waves); 3 = 256x128, 4 = 256x256 (8 waves, one workgroup per CU)
enum { TILE_64 = 0, TILE_128 = 1, TILE_128x64 = 2, TILE_256x128 = 3, TILE_256 = 4 };
template <int AM, int BMo, typename TC>
static int rung_tiles(const BArgs& k, int tile, hipStream_t st, int* bm_used) {
    switch (tile) {
#ifdef SAT_DEV_TILES
        case TILE_256: if (bm_used) *bm_used = 256; return glds_run_256<AM, BMo, TC>(k, st);
        case TILE_256x128: if (bm_used) *bm_used = 256; return glds_run_256x128<AM, BMo, TC>(k, st);
#else
        case TILE_256: case TILE_256x128: return -1;          // not in this build: the caller falls back to the 128 / 64 forms
#endif
        case TILE_128: if (bm_used) *bm_used = 128; return glds_run_128<AM, BMo, TC>(k, st);
        case TILE_128x64: if (bm_used) *bm_used = 128; return glds_run_128x64<AM, BMo, TC>(k, st);
        default: if (bm_used) *bm_used = 64; return glds_run_64<AM, BMo, TC>(k, st);
    }
}

int& glds_force_tile() { static int v = getenv("SAT_GLDS_TILE") ? atoi(getenv("SAT_GLDS_TILE")) : -1; return v; }
int& glds_ablate() { static int v = 0; return v; }
int& glds_tall_k() { static int v = getenv("SAT_GLDS_TALL_K") ? atoi(getenv("SAT_GLDS_TALL_K")) : 1 << 30; return v; }
int& glds_tall_conv() { static int v = getenv("SAT_GLDS_TALL_CONV") ? atoi(getenv("SAT_GLDS_TALL_CONV")) : 1152; return v; }
int& glds_stages8() { static int v = getenv("SAT_GLDS_STAGES8") ? atoi(getenv("SAT_GLDS_STAGES8")) : 0; return v; }

// -1: this problem does not fit the direct-to-LDS forms (caller keeps the register-staged kernel)
int launch_gemm_glds(const BArgs& k0, int amode, int bmode, int c_bf16, int BMt, hipStream_t st, int* bm_used) {
    const int force_tile = glds_force_tile();      // dev: force a tile form (enum above); SAT_GLDS_TILE or sat_debug_option("glds_tile", n)
    static const int off = getenv("SAT_NO_GLDS") ? atoi(getenv("SAT_NO_GLDS")) : 0;
    static const int force_stages = getenv("SAT_GLDS_STAGES") ? atoi(getenv("SAT_GLDS_STAGES")) : 0;
    static const int deep_from = getenv("SAT_GLDS_DEEP_FROM") ? atoi(getenv("SAT_GLDS_DEEP_FROM")) : 1 << 30;
    static const int rotate = getenv("SAT_GLDS_ROT") ? atoi(getenv("SAT_GLDS_ROT")) : 1 << 20;     // 0: off; n: rotation window in k-tiles
    if (off) return -1;
    BArgs k = k0;
    // ring depth: short reductions keep two stages (two workgroups per CU overlap each other); from `deep_from` k-tiles
    // on, one workgroup per CU with three tiles in flight hides the HBM / L2 latency that a cold operand costs
    const int ktiles = cdiv(k.kchunk < k.K ? k.kchunk : k.K, 64);
    // a single k-tile needs one stage: less LDS, a third workgroup per CU for the streaming 1x1 layers with 64 input channels
    // short reductions into a bf16 result (<= 168 VGPRs): one stage leaves room for a third workgroup per CU, which hides more of
    // the launch / load / store phases of these latency-bound tiles than the second stage does
    // (round 3, with the 128 x 64 tiles: EVERY bf16-result launch keeps one stage - the resident workgroups hide each other's latencies better than
    // a second stage of the same workgroup does: C2 21.26 -> 21.00 ms, C3 shard 14.19 -> 13.87, C4 shard 33.97 -> 33.79; fp32 results keep two
    // stages: one stage everywhere is 0.2 ms slower at C2)
    static const int s1_upto = getenv("SAT_GLDS_S1_UPTO") ? atoi(getenv("SAT_GLDS_S1_UPTO")) : 1 << 30;
    k.nstage = force_stages ? force_stages : (ktiles >= deep_from ? 4 : ((ktiles == 1 || (c_bf16 && ktiles <= s1_upto)) ? 1 : 2));
    k.rotate = rotate;
    if (k.nstage < 1) k.nstage = 1;
    if (k.nstage > 4) k.nstage = 4;
    // tile form (the caller's BMt says 64- or 128-wide): 128x64 for the 64-column outputs over many rows (halves the workgroup count
    // and the filter-tile re-reads of the 64x64 form), 8-wave forms on request
    static const int tall = getenv("SAT_GLDS_TALL") ? atoi(getenv("SAT_GLDS_TALL")) : 2;      // 0 off, 1 row-major A only, 2 also k-major A (1x1 weight gradients)
    int tile = (BMt >= 256) ? (BMt == 256 ? TILE_256 : TILE_256x128) : (BMt == 128 ? TILE_128 : TILE_64);      // BMt: 64, 128, 256 (= 256x256), 257 (= 256x128)
    if (tile == TILE_64 && tall && bmode != B_CONV_WGRAD && k.N <= 64 && (amode == A_KMAJOR ? (tall > 1 && k.M >= 128) : k.M >= 8192)) tile = TILE_128x64;
    // 128 x 64 instead of 128 x 128 for the dense row-major products (1x1 convolutions forward / data gradient, the decoder's wide products) and for
    // the 3x3 forward / data-gradient forms up to K = 1152 (64 and 128 input channels): twice the workgroups at half the accumulators - four to
    // five resident per CU instead of three, which hides more of each workgroup's load / store phases than the second filter-tile read costs.
    // Whole-step A/B (tools/ab_step.py, round 3): C2 21.93 -> 21.40 ms with the dense rule, 21.16 with both; C3 shard 14.54 -> 14.08; C4 shard
    // 34.00 -> 33.79 (its 3x3 forms with K >= 2304 keep 128 x 128: with them on 128 x 64 the step is 34.9); C1 unchanged.  Split-K launches,
    // k-major operands (weight gradients) and the filter-gradient forms keep 128 x 128.  (sat_debug_option "glds_tall_k" / "glds_tall_conv": the
    // largest reduction length that takes the narrow tile, 0 = off.)
    // ... and longer 3x3 reductions when 128 x 128 tiles would not give every CU two workgroups (C2's last stage: 8192 rows x 512 filters = 256
    // tiles; C4's 16384 x 512 = 512 tiles stays on 128 x 128: narrow tiles there cost 1 ms of its 33.7)
    static const int tall_conv_tiles = getenv("SAT_GLDS_TALL_CONV_TILES") ? atoi(getenv("SAT_GLDS_TALL_CONV_TILES")) : 512;
    if (tile == TILE_128 && bmode != B_CONV_WGRAD && k.nsplit <= 1 &&
        ((amode == A_ROW && k.K <= glds_tall_k()) || ((amode == A_CONV_FWD || amode == A_CONV_DGRAD) && (k.K <= glds_tall_conv() || (long)cdiv(k.M, 128) * cdiv(k.N, 128) < tall_conv_tiles)))) tile = TILE_128x64;
    // (k-major A - the 1x1 weight gradients - on 128 x 64 / 64 x 64 tiles: C2 20.65 -> 20.83 / 21.07 ms; they keep 128 x 128)
    if (force_tile >= 0) tile = force_tile;
    if (tile == TILE_256 || tile == TILE_256x128) {          // one workgroup per CU: the ring may be three deep on 256x128 (144 KiB), two on 256x256
        const int maxs = (tile == TILE_256) ? 2 : 3;
        const int deep8 = glds_stages8();
        if (ktiles > 1) k.nstage = deep8 ? (deep8 < maxs ? deep8 : maxs) : (ktiles >= 3 ? maxs : 2);
    }
    {
        static const int order = getenv("SAT_TILE_ORDER") ? atoi(getenv("SAT_TILE_ORDER")) : -1;      // dev: -1 automatic, 0 / 1 forced
        const int tbm = (tile == TILE_64) ? 64 : 128, tbn = (tile == TILE_128) ? 128 : 64;
        const int mt = cdiv(k.M, tbm), nt = cdiv(k.N, tbn);
        k.y_fastest = order >= 0 ? order : (amode == A_ROW && bmode != B_CONV_WGRAD && mt <= 16 && nt >= 2 * mt);
    }
    const ConvGeom& g = k.g;
    if (g.sw && g.sw != g.stride) return -1;          // anisotropic stride: register-staged kernel only
    if (k.K % 64 || k.kchunk % 64 || k.a_rows) return -1;
    if (amode == A_CONV_FWD && (g.C % 64 || g.R * g.S > 32)) return -1;
    if (amode == A_CONV_DGRAD && (g.K % 64 || g.R * g.S > 32 || (g.stride != 1 && !g.cls))) return -1;
    if (bmode == B_CONV_WGRAD && g.R * g.S > 32) return -1;
    // 32-bit byte offsets under 2 GiB descriptors
    const long lim = 1L << 30;      // elements
    long a_el = 0, b_el = 0;
    switch (amode) {
        case A_ROW: a_el = (long)k.M * k.lda; break;
        case A_KMAJOR: a_el = (long)k.K * k.lda; break;
        case A_CONV_FWD: a_el = (long)g.N * g.H * g.W * g.C; break;
        default: a_el = (long)g.N * g.P * g.Q * g.K; break;
    }
    switch (bmode) {
        case B_ROW: b_el = (long)k.N * k.ldb; break;
        case B_KMAJOR: b_el = (long)k.K * k.ldb; break;
        case B_CONV_WGRAD: b_el = (long)g.N * g.H * g.W * g.C; break;
        default: b_el = (long)g.K * g.R * g.S * g.C; break;
    }
    if (a_el >= lim || b_el >= lim) return -1;
    if ((amode == A_KMAJOR && k.M < 8) || (bmode != B_ROW && k.N < 8)) return -1;
#define SAT_GCASE(AMV, BMV, TC) return rung_tiles<AMV, BMV, TC>(k, tile, st, bm_used);
    if (amode == A_CONV_FWD && bmode == B_ROW && c_bf16) SAT_GCASE(A_CONV_FWD, B_ROW, __bf16)
    if (amode == A_ROW && bmode == B_ROW && c_bf16) SAT_GCASE(A_ROW, B_ROW, __bf16)
    if (amode == A_ROW && bmode == B_ROW && !c_bf16) SAT_GCASE(A_ROW, B_ROW, float)
    if (amode == A_CONV_DGRAD && bmode == B_CONV_DGRAD_W && c_bf16) SAT_GCASE(A_CONV_DGRAD, B_CONV_DGRAD_W, __bf16)
    if (amode == A_ROW && bmode == B_KMAJOR && c_bf16) SAT_GCASE(A_ROW, B_KMAJOR, __bf16)
    if (amode == A_ROW && bmode == B_KMAJOR && !c_bf16) SAT_GCASE(A_ROW, B_KMAJOR, float)
    if (amode == A_KMAJOR && bmode == B_CONV_WGRAD && !c_bf16) SAT_GCASE(A_KMAJOR, B_CONV_WGRAD, float)
    if (amode == A_KMAJOR && bmode == B_KMAJOR && !c_bf16) SAT_GCASE(A_KMAJOR, B_KMAJOR, float)
#undef SAT_GCASE
    return -1;
}

}  // namespace sat
