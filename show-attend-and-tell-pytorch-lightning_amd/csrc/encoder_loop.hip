// The residual-block loop of the ResNet trunk inside the library (training, bf16 storage): one call runs the forward of a run of
// blocks, one call their backward, over ONE activation arena - instead of ~10 (forward) / ~14 (backward) Python -> ctypes calls and as many
// tensor allocations per block (VERDICT r2 "what's missing" 1).  Reference side: the blocks are torchvision's BasicBlock /
// Bottleneck inside `self.encoder(img)` (model.py:19-29, 483) and their autograd.
//
// The loop calls the same per-layer entry points the Python driver calls (sat_conv2d_*_bf16*, sat_bn_train_*), in the same order with the
// same arguments: results are bit-identical to the per-layer path (tests/test_gpu_encoder.py).  The arena layout is a pure function of the
// block descriptors (sat_encoder_blocks_arena_bytes); nothing is allocated here.
#include "../../include/sat_hip.h"
#include "common.h"

namespace sat {
namespace {

typedef unsigned short bf16_t;      // storage only

struct Stats { float* mean; float* invstd; uint8_t* mask; };

struct BlockBufs {
    // forward (kept for the backward pass)
    void *c1, *a1, *c2, *a2, *c3, *cd, *out;
    Stats s1, s2, s3, sd;                       // sd.mask unused
    float *tl_a, *tl_b;                         // forward tile statistics (main path / shortcut)
    // backward
    void *g, *dx3, *da2, *dx2, *da1, *dx1, *dxd, *dx;
    float *t1, *t2, *tprev;                     // backward tile statistics: for bn1, bn2, and for the previous block's last BatchNorm
};

struct Dims { long rows_in, rows_mid, rows_out; int P, Q; };

inline Dims dims_of(const sat_block_desc& b) {
    Dims d;
    d.P = (b.H + 2 - 3) / b.stride + 1; d.Q = (b.W + 2 - 3) / b.stride + 1;
    d.rows_in = (long)b.N * b.H * b.W;
    d.rows_out = (long)b.N * d.P * d.Q;
    // bottleneck: conv1 (1x1, stride 1) keeps the input map, the 3x3 strides; basic: conv1 (3x3) strides
    d.rows_mid = b.kind == 1 ? d.rows_in : d.rows_out;
    return d;
}
inline sat_conv_geom geom(int N, int H, int W, int C, int K, int R, int stride, int pad) {
    sat_conv_geom g; g.N = N; g.H = H; g.W = W; g.C = C; g.K = K; g.R = R; g.S = R; g.stride = stride; g.pad = pad; g.stride_w = 0; return g;
}

struct Arena {
    char* base; size_t off;
    void* take(size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return base ? base + o : nullptr; }
};

// every buffer of every block, in a fixed order (the same walk sizes the arena and hands out the pointers)
void layout(const sat_block_desc* blocks, int n, char* base, BlockBufs* out, size_t* total) {
    Arena a{base, 0};
    for (int i = 0; i < n; ++i) {
        const sat_block_desc& b = blocks[i];
        const Dims d = dims_of(b);
        BlockBufs w = {};
        auto stats = [&](int C, long rows) { Stats s; s.mean = (float*)a.take((size_t)C * 4); s.invstd = (float*)a.take((size_t)C * 4); s.mask = (uint8_t*)a.take((size_t)rows * C / 8); return s; };
        const int c1w = b.kind == 1 ? b.mid : b.cout;          // width of conv1's output
        w.c1 = a.take((size_t)d.rows_mid * c1w * 2); w.a1 = a.take((size_t)d.rows_mid * c1w * 2); w.s1 = stats(c1w, d.rows_mid);
        sat_conv_geom g1 = b.kind == 1 ? geom(b.N, b.H, b.W, b.cin, b.mid, 1, 1, 0) : geom(b.N, b.H, b.W, b.cin, b.cout, 3, b.stride, 1);
        size_t tl = sat_conv2d_fwd_stats_bytes(&g1);
        if (b.kind == 1) {
            w.c2 = a.take((size_t)d.rows_out * b.mid * 2); w.a2 = a.take((size_t)d.rows_out * b.mid * 2); w.s2 = stats(b.mid, d.rows_out);
            w.c3 = a.take((size_t)d.rows_out * b.cout * 2); w.s3 = stats(b.cout, d.rows_out);
            sat_conv_geom g2 = geom(b.N, b.H, b.W, b.mid, b.mid, 3, b.stride, 1), g3 = geom(b.N, d.P, d.Q, b.mid, b.cout, 1, 1, 0);
            size_t t2 = sat_conv2d_fwd_stats_bytes(&g2), t3 = sat_conv2d_fwd_stats_bytes(&g3);
            if (t2 > tl) tl = t2;
            if (t3 > tl) tl = t3;
        } else {
            w.c2 = a.take((size_t)d.rows_out * b.cout * 2); w.s2 = stats(b.cout, d.rows_out);
            sat_conv_geom g2 = geom(b.N, d.P, d.Q, b.cout, b.cout, 3, 1, 1);
            size_t t2 = sat_conv2d_fwd_stats_bytes(&g2);
            if (t2 > tl) tl = t2;
        }
        w.tl_a = (float*)a.take(tl);
        if (b.has_ds) {
            w.cd = a.take((size_t)d.rows_out * b.cout * 2);
            w.sd.mean = (float*)a.take((size_t)b.cout * 4); w.sd.invstd = (float*)a.take((size_t)b.cout * 4); w.sd.mask = nullptr;
            sat_conv_geom gd = geom(b.N, b.H, b.W, b.cin, b.cout, 1, b.stride, 0);
            w.tl_b = (float*)a.take(sat_conv2d_fwd_stats_bytes(&gd));
        }
        w.out = a.take((size_t)d.rows_out * b.cout * 2);
        // backward
        w.g = a.take((size_t)d.rows_out * b.cout * 2);
        if (b.kind == 1) {
            w.dx3 = a.take((size_t)d.rows_out * b.cout * 2); w.da2 = a.take((size_t)d.rows_out * b.mid * 2); w.dx2 = a.take((size_t)d.rows_out * b.mid * 2);
            w.da1 = a.take((size_t)d.rows_mid * b.mid * 2); w.dx1 = a.take((size_t)d.rows_mid * b.mid * 2);
            sat_conv_geom g3 = geom(b.N, d.P, d.Q, b.mid, b.cout, 1, 1, 0), g2 = geom(b.N, b.H, b.W, b.mid, b.mid, 3, b.stride, 1);
            w.t2 = (float*)a.take(sat_conv2d_dgrad_stats_bytes(&g3)); w.t1 = (float*)a.take(sat_conv2d_dgrad_stats_bytes(&g2));
        } else {
            w.dx2 = a.take((size_t)d.rows_out * b.cout * 2); w.da1 = a.take((size_t)d.rows_out * b.cout * 2); w.dx1 = a.take((size_t)d.rows_out * b.cout * 2);
            sat_conv_geom g2 = geom(b.N, d.P, d.Q, b.cout, b.cout, 3, 1, 1);
            w.t1 = (float*)a.take(sat_conv2d_dgrad_stats_bytes(&g2));
        }
        if (b.has_ds) w.dxd = a.take((size_t)d.rows_out * b.cout * 2);
        w.dx = a.take((size_t)d.rows_in * b.cin * 2);
        sat_conv_geom gf = b.kind == 1 ? geom(b.N, b.H, b.W, b.cin, b.mid, 1, 1, 0) : geom(b.N, b.H, b.W, b.cin, b.cout, 3, b.stride, 1);
        w.tprev = (float*)a.take(sat_conv2d_dgrad_stats_bytes(&gf));
        if (out) out[i] = w;
    }
    *total = a.off;
}

constexpr int MAX_BLOCKS = 64;

// BatchNorm forward as encoder.py:bn_fwd dispatches it (training, bf16 storage)
int bn_forward(const sat_block_desc& b, int which, const void* x, long rows, int C, const float* tiles, int tile_rows, const void* residual,
               const Stats* res_stats, int res_which, int relu, void* y, Stats& s, bool want_mask, float* scratch, void* st) {
    uint8_t* mask = (want_mask && relu && C % 8 == 0) ? s.mask : nullptr;
    if (res_stats)
        return sat_bn_train_fwd_tiles_bf16_resbn(x, rows, C, tiles, tile_rows, b.gamma[which], b.beta[which], b.eps[which], b.momentum[which], b.running_mean[which],
                                                 b.running_var[which], s.mean, s.invstd, residual, res_stats->mean, res_stats->invstd, b.gamma[res_which],
                                                 b.beta[res_which], relu, y, mask, scratch, st);
    if (tiles && tile_rows > 0)
        return sat_bn_train_fwd_tiles_bf16(x, rows, C, tiles, tile_rows, b.gamma[which], b.beta[which], b.eps[which], b.momentum[which], b.running_mean[which],
                                           b.running_var[which], s.mean, s.invstd, residual, relu, y, mask, scratch, st);
    return sat_bn_train_fwd_t(1, x, rows, C, b.gamma[which], b.beta[which], b.eps[which], b.momentum[which], b.running_mean[which], b.running_var[which], s.mean,
                              s.invstd, residual, relu, y, mask, scratch, st);
}

// BatchNorm backward as encoder.py:bn_bwd dispatches it
int bn_backward(const sat_block_desc& b, int which, const void* dy, const void* x, const void* y, long rows, int C, const Stats& s, const uint8_t* mask, int relu,
                void* dx, void* dres, const float* tiles, int tile_rows, float* scratch, void* st) {
    if (tiles && tile_rows > 0 && (!relu || mask))
        return sat_bn_train_bwd_tiles_bf16(dy, x, rows, C, tiles, tile_rows, s.mean, s.invstd, b.gamma[which], relu, dx, b.dgamma[which], b.dbeta[which], dres, 0, mask,
                                           scratch, st);
    return sat_bn_train_bwd_t(1, dy, x, y, rows, C, s.mean, s.invstd, b.gamma[which], relu, dx, b.dgamma[which], b.dbeta[which], dres, 0, mask, scratch, st);
}

}  // namespace
}  // namespace sat

using namespace sat;

extern "C" {

size_t sat_encoder_blocks_arena_bytes(const sat_block_desc* blocks, int32_t nblocks) {
    if (!blocks || nblocks <= 0 || nblocks > MAX_BLOCKS) return 0;
    size_t total = 0;
    layout(blocks, nblocks, nullptr, nullptr, &total);
    return total;
}

int sat_encoder_blocks_fwd(const sat_block_desc* blocks, int32_t nblocks, const void* x, void* arena, size_t arena_bytes, float* bn_scratch, void** out_ptr,
                           void* stream) {
    if (!blocks || !x || !arena || !bn_scratch || nblocks <= 0 || nblocks > MAX_BLOCKS) return fail(SAT_EINVAL, "encoder_blocks_fwd: bad argument");
    BlockBufs bufs[MAX_BLOCKS];
    size_t total = 0;
    layout(blocks, nblocks, (char*)arena, bufs, &total);
    SAT_REQUIRE(arena_bytes >= total, "encoder_blocks_fwd: arena %zu < %zu bytes", arena_bytes, total);
    const void* xin = x;
    for (int i = 0; i < nblocks; ++i) {
        const sat_block_desc& b = blocks[i];
        const BlockBufs& w = bufs[i];
        const Dims d = dims_of(b);
        int32_t tr = 0, trd = 0;
        const void* last; int last_bn, last_C = b.cout;
        if (b.kind == 0) {
            sat_conv_geom g1 = geom(b.N, b.H, b.W, b.cin, b.cout, 3, b.stride, 1), g2 = geom(b.N, d.P, d.Q, b.cout, b.cout, 3, 1, 1);
            SAT_TRY(sat_conv2d_fwd_bf16_stats(xin, b.w1, w.c1, &g1, w.tl_a, &tr, stream));
            Stats s1 = w.s1;
            SAT_TRY(bn_forward(b, 0, w.c1, d.rows_out, b.cout, w.tl_a, tr, nullptr, nullptr, 0, 1, w.a1, s1, true, bn_scratch, stream));
            SAT_TRY(sat_conv2d_fwd_bf16_stats(w.a1, b.w2, w.c2, &g2, w.tl_a, &tr, stream));
            last = w.c2; last_bn = 1;
        } else {
            sat_conv_geom g1 = geom(b.N, b.H, b.W, b.cin, b.mid, 1, 1, 0), g2 = geom(b.N, b.H, b.W, b.mid, b.mid, 3, b.stride, 1),
                          g3 = geom(b.N, d.P, d.Q, b.mid, b.cout, 1, 1, 0);
            SAT_TRY(sat_conv2d_fwd_bf16_stats(xin, b.w1, w.c1, &g1, w.tl_a, &tr, stream));
            Stats s1 = w.s1;
            SAT_TRY(bn_forward(b, 0, w.c1, d.rows_in, b.mid, w.tl_a, tr, nullptr, nullptr, 0, 1, w.a1, s1, true, bn_scratch, stream));
            SAT_TRY(sat_conv2d_fwd_bf16_stats(w.a1, b.w2, w.c2, &g2, w.tl_a, &tr, stream));
            Stats s2 = w.s2;
            SAT_TRY(bn_forward(b, 1, w.c2, d.rows_out, b.mid, w.tl_a, tr, nullptr, nullptr, 0, 1, w.a2, s2, true, bn_scratch, stream));
            SAT_TRY(sat_conv2d_fwd_bf16_stats(w.a2, b.w3, w.c3, &g3, w.tl_a, &tr, stream));
            last = w.c3; last_bn = 2;
        }
        const void* idn = xin;
        const Stats* res = nullptr;
        Stats sd = w.sd;
        if (b.has_ds) {
            sat_conv_geom gd = geom(b.N, b.H, b.W, b.cin, b.cout, 1, b.stride, 0);
            SAT_TRY(sat_conv2d_fwd_bf16_stats(xin, b.wd, w.cd, &gd, w.tl_b, &trd, stream));
            if (b.fwd_res_bn && tr > 0 && trd > 0) {
                // projection shortcut: only its statistics are taken here; the last BatchNorm's kernel normalises cd on the fly
                SAT_TRY(sat_bn_train_fwd_tiles_bf16(w.cd, d.rows_out, b.cout, w.tl_b, trd, b.gamma[3], b.beta[3], b.eps[3], b.momentum[3], b.running_mean[3],
                                                    b.running_var[3], sd.mean, sd.invstd, nullptr, 0, nullptr, nullptr, bn_scratch, stream));
                idn = w.cd; res = &sd;
            } else {
                // materialised shortcut: its BatchNorm output overwrites nothing the backward needs (cd stays), so it goes to the g buffer
                SAT_TRY(bn_forward(b, 3, w.cd, d.rows_out, b.cout, w.tl_b, trd, nullptr, nullptr, 0, 0, w.g, sd, false, bn_scratch, stream));
                idn = w.g;
            }
        }
        Stats sl = (b.kind == 0) ? w.s2 : w.s3;
        SAT_TRY(bn_forward(b, last_bn, last, d.rows_out, last_C, w.tl_a, tr, idn, res, 3, 1, w.out, sl, true, bn_scratch, stream));
        xin = w.out;
    }
    if (out_ptr) *out_ptr = const_cast<void*>(xin);
    return SAT_OK;
}

// Backward of blocks [last .. first] (descending; the whole table describes the trunk, a call may cover one ResNet stage of it so that the caller
// can start a gradient exchange per stage).  dout: gradient of block `last`'s output (bf16, its shape) with the backward statistics of that block's
// last BatchNorm per row tile if the launch that wrote dout produced them (dout_tiles / dout_tile_rows: what the previous call returned, else NULL / 0).
// *dx_ptr: where the gradient of block `first`'s input lies (inside the arena), *dx_tiles / *dx_tile_rows: the statistics that came with it.
// Weight gradients go to `side_stream` when given (fork / join through `event`: the caller's stream waits for the side stream before this call
// returns), each stream with its own split-K scratch.
int sat_encoder_blocks_bwd(const sat_block_desc* blocks, int32_t nblocks, int32_t first, int32_t last, const void* x, void* arena, size_t arena_bytes, const void* dout,
                           const float* dout_tiles, int32_t dout_tile_rows, float* bn_scratch, float* slab_main, int64_t slab_main_elems, float* slab_side,
                           int64_t slab_side_elems, void* side_stream, void* event, void** dx_ptr, float** dx_tiles, int32_t* dx_tile_rows, void* stream) {
    if (!blocks || !x || !arena || !dout || !bn_scratch || !slab_main || nblocks <= 0 || nblocks > MAX_BLOCKS || first < 0 || last >= nblocks || first > last)
        return fail(SAT_EINVAL, "encoder_blocks_bwd: bad argument");
    if (side_stream && (!event || !slab_side)) return fail(SAT_EINVAL, "encoder_blocks_bwd: side stream without event / scratch");
    BlockBufs bufs[MAX_BLOCKS];
    size_t total = 0;
    layout(blocks, nblocks, (char*)arena, bufs, &total);
    SAT_REQUIRE(arena_bytes >= total, "encoder_blocks_bwd: arena %zu < %zu bytes", arena_bytes, total);
    hipStream_t st = (hipStream_t)stream, sd = (hipStream_t)side_stream;
    bool side_dirty = false;
    auto wgrad = [&](const void* dy, const void* xx, float* dw, const sat_conv_geom& g) -> int {
        if (!sd) return sat_conv2d_wgrad_bf16(dy, xx, dw, &g, slab_main, slab_main_elems, stream);
        SAT_CHECK_HIP(hipEventRecord((hipEvent_t)event, st));          // everything enqueued so far (dy is complete) ...
        SAT_CHECK_HIP(hipStreamWaitEvent(sd, (hipEvent_t)event, 0));   // ... precedes the launch on the side stream
        side_dirty = true;
        return sat_conv2d_wgrad_bf16(dy, xx, dw, &g, slab_side, slab_side_elems, side_stream);
    };
    const void* d_cur = dout;
    const float* d_tiles = (dout_tiles && dout_tile_rows > 0) ? dout_tiles : nullptr; int d_tile_rows = d_tiles ? dout_tile_rows : 0;
    for (int i = last; i >= first; --i) {
        const sat_block_desc& b = blocks[i];
        const BlockBufs& w = bufs[i];
        const Dims d = dims_of(b);
        const void* xin = i > 0 ? bufs[i - 1].out : x;
        const Stats& sl = b.kind == 0 ? w.s2 : w.s3;
        const bool have_mask = b.dgrad_join && (b.cout % 8 == 0);
        const bool join = have_mask && !b.has_ds;
        const bool mask_ds = have_mask && b.has_ds;
        void* g = (join || mask_ds) ? nullptr : w.g;
        const void* da1; int32_t t1r = 0, t2r = 0;
        sat_conv_geom gfirst; const void* wfirst; float* dwfirst;
        if (b.kind == 0) {
            sat_conv_geom g2 = geom(b.N, d.P, d.Q, b.cout, b.cout, 3, 1, 1);
            SAT_TRY(bn_backward(b, 1, d_cur, w.c2, w.out, d.rows_out, b.cout, w.s2, w.s2.mask, 1, w.dx2, g, d_tiles, d_tile_rows, bn_scratch, stream));
            SAT_TRY(wgrad(w.dx2, w.a1, b.dw2, g2));
            if (b.bn_bwd_epilogue) SAT_TRY(sat_conv2d_dgrad_bf16_bnstats(w.dx2, b.w2, w.da1, &g2, 0, w.c1, w.s1.mask, w.s1.mean, w.s1.invstd, w.t1, &t1r, stream));
            else SAT_TRY(sat_conv2d_dgrad_bf16(w.dx2, b.w2, w.da1, &g2, 0, stream));
            da1 = w.da1;
            gfirst = geom(b.N, b.H, b.W, b.cin, b.cout, 3, b.stride, 1); wfirst = b.w1; dwfirst = b.dw1;
            SAT_TRY(bn_backward(b, 0, da1, w.c1, w.a1, d.rows_out, b.cout, w.s1, w.s1.mask, 1, w.dx1, nullptr, w.t1, t1r, bn_scratch, stream));
        } else {
            sat_conv_geom g3 = geom(b.N, d.P, d.Q, b.mid, b.cout, 1, 1, 0), g2 = geom(b.N, b.H, b.W, b.mid, b.mid, 3, b.stride, 1);
            SAT_TRY(bn_backward(b, 2, d_cur, w.c3, w.out, d.rows_out, b.cout, w.s3, w.s3.mask, 1, w.dx3, g, d_tiles, d_tile_rows, bn_scratch, stream));
            SAT_TRY(wgrad(w.dx3, w.a2, b.dw3, g3));
            if (b.bn_bwd_epilogue) SAT_TRY(sat_conv2d_dgrad_bf16_bnstats(w.dx3, b.w3, w.da2, &g3, 0, w.c2, w.s2.mask, w.s2.mean, w.s2.invstd, w.t2, &t2r, stream));
            else SAT_TRY(sat_conv2d_dgrad_bf16(w.dx3, b.w3, w.da2, &g3, 0, stream));
            SAT_TRY(bn_backward(b, 1, w.da2, w.c2, w.a2, d.rows_out, b.mid, w.s2, w.s2.mask, 1, w.dx2, nullptr, w.t2, t2r, bn_scratch, stream));
            SAT_TRY(wgrad(w.dx2, w.a1, b.dw2, g2));
            if (b.bn_bwd_epilogue && b.stride == 1) SAT_TRY(sat_conv2d_dgrad_bf16_bnstats(w.dx2, b.w2, w.da1, &g2, 0, w.c1, w.s1.mask, w.s1.mean, w.s1.invstd, w.t1, &t1r, stream));
            else SAT_TRY(sat_conv2d_dgrad_bf16(w.dx2, b.w2, w.da1, &g2, 0, stream));
            da1 = w.da1;
            gfirst = geom(b.N, b.H, b.W, b.cin, b.mid, 1, 1, 0); wfirst = b.w1; dwfirst = b.dw1;
            SAT_TRY(bn_backward(b, 0, da1, w.c1, w.a1, d.rows_in, b.mid, w.s1, w.s1.mask, 1, w.dx1, nullptr, w.t1, t1r, bn_scratch, stream));
        }
        SAT_TRY(wgrad(w.dx1, xin, dwfirst, gfirst));
        if (b.has_ds) {
            sat_conv_geom gd = geom(b.N, b.H, b.W, b.cin, b.cout, 1, b.stride, 0);
            if (mask_ds) SAT_TRY(bn_backward(b, 3, d_cur, w.cd, nullptr, d.rows_out, b.cout, w.sd, sl.mask, 1, w.dxd, nullptr, nullptr, 0, bn_scratch, stream));
            else SAT_TRY(bn_backward(b, 3, g, w.cd, nullptr, d.rows_out, b.cout, w.sd, nullptr, 0, w.dxd, nullptr, nullptr, 0, bn_scratch, stream));
            SAT_TRY(wgrad(w.dxd, xin, b.dwd, gd));
            SAT_TRY(sat_conv2d_dgrad_bf16(w.dx1, wfirst, w.dx, &gfirst, 0, stream));
            // the strided 1x1 shortcut only reaches the even pixels: accumulate it on top (the other parity classes are skipped)
            SAT_TRY(sat_conv2d_dgrad_bf16(w.dxd, b.wd, w.dx, &gd, 1, stream));
            d_cur = w.dx; d_tiles = nullptr; d_tile_rows = 0;
            continue;
        }
        // identity path + conv path; the launch that writes the sum also leaves the statistics of the previous block's last BatchNorm
        const bool have_prev = i > 0 && b.bn_bwd_epilogue;
        const void* pbx = nullptr; Stats ps = {};
        if (have_prev) { const BlockBufs& pw = bufs[i - 1]; pbx = blocks[i - 1].kind == 0 ? pw.c2 : pw.c3; ps = blocks[i - 1].kind == 0 ? pw.s2 : pw.s3; }
        int32_t tpr = 0;
        if (join) {
            SAT_TRY(sat_conv2d_dgrad_bf16_fused(w.dx1, wfirst, w.dx, &gfirst, d_cur, sl.mask, pbx, pbx ? ps.mask : nullptr, pbx ? ps.mean : nullptr,
                                                pbx ? ps.invstd : nullptr, pbx ? w.tprev : nullptr, &tpr, stream));
            d_cur = w.dx;
        } else {
            if (pbx) SAT_TRY(sat_conv2d_dgrad_bf16_bnstats(w.dx1, wfirst, g, &gfirst, 1, pbx, ps.mask, ps.mean, ps.invstd, w.tprev, &tpr, stream));
            else SAT_TRY(sat_conv2d_dgrad_bf16(w.dx1, wfirst, g, &gfirst, 1, stream));
            d_cur = g;
        }
        d_tiles = (pbx && tpr > 0) ? w.tprev : nullptr; d_tile_rows = tpr;
    }
    if (sd && side_dirty) {
        SAT_CHECK_HIP(hipEventRecord((hipEvent_t)event, sd));
        SAT_CHECK_HIP(hipStreamWaitEvent(st, (hipEvent_t)event, 0));
    }
    if (dx_ptr) *dx_ptr = const_cast<void*>(d_cur);
    if (dx_tiles) *dx_tiles = const_cast<float*>(d_tiles);
    if (dx_tile_rows) *dx_tile_rows = d_tile_rows;
    return SAT_OK;
}

}
