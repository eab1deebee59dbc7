"""Encoder host side: the reference's ``get_encoder`` (model.py:16-63) for the ResNet family with the
layers executed by ``libsat_hip.so`` (NHWC activations, KRSC filters, implicit-GEMM convolutions on MFMA).

The returned module is an ``nn.Sequential`` whose children sit at the same indices as the reference's
(0 Normalize, 1 conv1, 2 bn1, 3 relu, 4 maxpool, 5-8 layer1-4, 9 the 1x1 projection), so state-dict keys
match.  The children only hold parameters; ``forward`` runs the HIP pipeline and returns the
annotations as a (B, D, h, w) tensor in channels-last memory, i.e. (B, h*w, D) row-major underneath.
"""
import ctypes as C
import glob
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib as L

#: arch -> (block kind, blocks per stage, width per group[, groups of the 3x3 convolution]); torchvision's table.  The resnext archs belong to the
#: same branch of the reference's get_encoder (model.py:28: "resnet" in arch or "resnext" in arch)
RESNETS = {
    "resnet18": ("basic", (2, 2, 2, 2), 64), "resnet34": ("basic", (3, 4, 6, 3), 64),
    "resnet50": ("bottleneck", (3, 4, 6, 3), 64), "resnet101": ("bottleneck", (3, 4, 23, 3), 64),
    "resnet152": ("bottleneck", (3, 8, 36, 3), 64),
    "wide_resnet50_2": ("bottleneck", (3, 4, 6, 3), 128), "wide_resnet101_2": ("bottleneck", (3, 4, 23, 3), 128),
    "resnext50_32x4d": ("bottleneck", (3, 4, 6, 3), 4, 32), "resnext101_32x8d": ("bottleneck", (3, 4, 23, 3), 8, 32),
}


class Normalize(nn.Module):
    """Holder of the Normalize(mean, std) constants (model.py:59); the arithmetic is fused into the NHWC
    conversion kernel.  Unlike the reference's inplace=True it does not mutate the caller's batch."""

    def __init__(self, mean, std, inplace=True):
        super().__init__()
        self.mean, self.std, self.inplace = [float(v) for v in mean], [float(v) for v in std], inplace


class Block(nn.Module):
    """Parameter holder of one torchvision BasicBlock / Bottleneck (stride on the 3x3)."""

    def __init__(self, kind, cin, planes, stride, wpg, groups=1):
        super().__init__()
        self.kind, self.stride, self.groups = kind, stride, groups
        if kind == "basic":
            self.cout = planes
            self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False); self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False); self.bn2 = nn.BatchNorm2d(planes)
        else:
            mid = int(planes * (wpg / 64.0)) * groups; self.cout = planes * 4          # torchvision Bottleneck: width = int(planes * base_width / 64) * groups
            self.conv1 = nn.Conv2d(cin, mid, 1, 1, 0, bias=False); self.bn1 = nn.BatchNorm2d(mid)
            self.conv2 = nn.Conv2d(mid, mid, 3, stride, 1, groups=groups, bias=False); self.bn2 = nn.BatchNorm2d(mid)
            self.conv3 = nn.Conv2d(mid, self.cout, 1, 1, 0, bias=False); self.bn3 = nn.BatchNorm2d(self.cout)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != self.cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, self.cout, 1, stride, 0, bias=False), nn.BatchNorm2d(self.cout))


def _shadow_(module):
    """Host-time detail: ``nn.Module`` keeps sub-modules and parameters in dicts behind a Python-level ``__getattr__`` (~1 us a lookup, ~1250
    lookups per resnet50 step).  Putting the same objects into the instance ``__dict__`` makes ``blk.conv1.weight`` a plain attribute read;
    ``Module.__setattr__`` drops such an entry when the attribute is reassigned, and ``.cuda()`` / ``.to()`` keep Parameter objects (they
    replace BUFFER tensors, which is why buffers are not shadowed)."""
    for mod in module.modules():
        for name, sub in mod._modules.items():
            if sub is not None:
                mod.__dict__[name] = sub
        for name, prm in mod._parameters.items():
            if prm is not None:
                mod.__dict__[name] = prm


def _channels_last_(module):
    for mod in module.modules():
        if isinstance(mod, nn.Conv2d):
            mod.weight.data = mod.weight.data.contiguous(memory_format=torch.channels_last)


# ----------------------------------------------------------------------------- raw layer calls
_geoms = {}
_stats_n = {}


def _stats_elems(g, fwd):
    """floats of the tile-statistics array of a convolution launch (forward / data gradient), asked of the library once per geometry"""
    key = (id(g), fwd)
    n = _stats_n.get(key)
    if n is None:
        lib = L.lib()
        n = _stats_n[key] = (lib.sat_conv2d_fwd_stats_bytes(C.byref(g)) if fwd else lib.sat_conv2d_dgrad_stats_bytes(C.byref(g))) // 4
    return n


def _geom(N, H, W, Cc, K, R, S, stride, pad, stride_w=0):
    key = (N, H, W, Cc, K, R, S, stride, pad, stride_w)
    g = _geoms.get(key)
    if g is None:          # the library only reads it: one struct per geometry, not one per call
        g = _geoms[key] = L.ConvGeom(N=N, H=H, W=W, C=Cc, K=K, R=R, S=S, stride=stride, pad=pad, stride_w=int(stride_w or 0))
    return g


def _krsc(w):
    """(K,C,R,S) parameter -> tensor whose memory is K,R,S,C."""
    if w.dim() == 4 and not w.is_contiguous(memory_format=torch.channels_last):
        w = w.contiguous(memory_format=torch.channels_last)
    return w


def _out_hw(H, W, R, S, stride, pad, stride_w=0):
    return (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // (stride_w or stride) + 1


BF16 = torch.bfloat16


def _is_bf(t):
    return t.dtype == BF16


def cast_bf16(t):
    """fp32 -> bf16 copy through the library (filters, small gradient maps)."""
    t = t.contiguous() if t.dim() != 4 else _krsc(t)
    out = torch.empty(t.shape, dtype=BF16, device=t.device)
    if t.dim() == 4:
        out = out.contiguous(memory_format=torch.channels_last)
    L.check(L.lib().sat_cast_f32_to_bf16(L.ptr(t), L.ptr(out), t.numel(), L.stream_ptr()), "sat_cast_f32_to_bf16")
    return out


def grouped_dense(w, groups):
    """Grouped filter (K, C/groups, R, S) of a resnext 3x3 convolution -> the dense block-diagonal filter (K, C, R, S) (KRSC memory, zeros off the
    diagonal blocks) that the implicit-GEMM kernels read: the grouped convolution runs as an ordinary one.  At 4 - 8 channels per group the
    matrix cores would idle 8 - 10x on the real products anyway and the layer is bound by its activation traffic; the dense form costs what
    a wide_resnet 3x3 of the same width costs.  (Filter reshaping only: ``sat_grouped_filter_expand``.)"""
    K, cg, R, S = w.shape
    w = _krsc(w)
    out = torch.empty(K, cg * groups, R, S, dtype=w.dtype, device=w.device).contiguous(memory_format=torch.channels_last)
    L.check(L.lib().sat_grouped_filter_expand(L.ptr(w), L.ptr(out), K, cg * groups, R * S, groups, int(_is_bf(w)), L.stream_ptr()), "sat_grouped_filter_expand")
    return out


def grouped_grad(dw_dense, param, groups):
    """gradient of the grouped filter = the diagonal blocks of the dense filter's gradient (fp32, KRSC memory) -> ``param``'s gradient buffer"""
    K, C, R, S = dw_dense.shape
    out = L.grad_buffer(param)
    direct = tuple(out.shape) == tuple(param.shape) and out.permute(0, 2, 3, 1).is_contiguous()
    dst = out if direct else torch.empty(K, R, S, C // groups, dtype=torch.float32, device=dw_dense.device)
    src = dw_dense.permute(0, 2, 3, 1)
    assert src.is_contiguous() and dw_dense.dtype == torch.float32
    L.check(L.lib().sat_grouped_filter_grad_extract(L.ptr(src), L.ptr(dst), K, C, R * S, groups, L.stream_ptr()), "sat_grouped_filter_grad_extract")
    return out if direct else dst.permute(0, 3, 1, 2)


def conv_fwd(x, w, stride, pad, bias=None, stride_w=0):
    """x NHWC (fp32 or bf16); w (K,C,R,S) in KRSC memory with x's dtype.  ``stride_w``: horizontal stride when it differs (bf16 stem)."""
    N, H, W, Cc = x.shape
    K, _, R, S = w.shape
    P, Q = _out_hw(H, W, R, S, stride, pad, stride_w)
    y = torch.empty(N, P, Q, K, dtype=x.dtype, device=x.device)
    g = _geom(N, H, W, Cc, K, R, S, stride, pad, stride_w)
    fn = L.lib().sat_conv2d_fwd_bf16 if _is_bf(x) else L.lib().sat_conv2d_fwd
    L.check(fn(L.ptr(x), L.ptr(_krsc(w)), L.ptr(bias), L.ptr(y), C.byref(g), L.stream_ptr()), "sat_conv2d_fwd")
    return y


def conv_fwd_stats(x, w, stride, pad, stride_w=0):
    """bf16 convolution whose epilogue also leaves the BatchNorm statistics of its output per row tile.  Returns
    (y, tiles) where tiles = (tile_stats, tile_rows) for ``bn_fwd(..., tiles=tiles)``, or None when the launch took a kernel
    without that epilogue (fp32 storage, odd shapes)."""
    if not _is_bf(x):
        return conv_fwd(x, w, stride, pad), None
    lib = L.lib()
    N, H, W, Cc = x.shape
    K, _, R, S = w.shape
    P, Q = _out_hw(H, W, R, S, stride, pad, stride_w)
    y = torch.empty(N, P, Q, K, dtype=x.dtype, device=x.device)
    g = _geom(N, H, W, Cc, K, R, S, stride, pad, stride_w)
    stats = torch.empty(_stats_elems(g, True), dtype=torch.float32, device=x.device)
    rows = C.c_int32(0)
    L.check(lib.sat_conv2d_fwd_bf16_stats(L.ptr(x), L.ptr(_krsc(w)), L.ptr(y), C.byref(g), L.ptr(stats), C.byref(rows), L.stream_ptr()),
            "sat_conv2d_fwd_bf16_stats")
    return y, ((stats, rows.value) if rows.value > 0 else None)


#: backward BatchNorm statistics from the epilogue of the data-gradient launch that produces the BatchNorm's output gradient
#: (tests switch it off to compare with the separate statistics pass)
_BN_BWD_EPILOGUE = os.environ.get("SAT_BN_BWD_EPILOGUE", "1") != "0"


def conv_dgrad(dy, w, x_shape, stride, pad, out=None, accumulate=False, bn=None, add=None):
    """``bn`` = (bn_input, (mean, invstd[, relu_mask])) of the BatchNorm(+ReLU) whose OUTPUT is this convolution's input: the launch then
    also leaves that BatchNorm's backward statistics per row tile (bf16 storage, stride 1) and the call returns (dx, tiles) with
    tiles = (tile_stats, tile_rows) for ``bn_bwd(..., tiles=tiles)``, or (dx, None) when the launch could not produce them.
    ``add`` = (tensor, relu_mask or None): dx = data gradient + tensor gated by the mask bits (bf16 storage, stride 1): the identity path of a
    residual block (``sat_conv2d_dgrad_bf16_fused``)."""
    N, H, W, Cc = x_shape
    K, _, R, S = w.shape
    dx = out if out is not None else torch.empty(N, H, W, Cc, dtype=dy.dtype, device=dy.device)
    g = _geom(N, H, W, Cc, K, R, S, stride, pad)
    lib = L.lib()
    if add is not None:
        src, mask = add
        assert _is_bf(dy) and stride == 1 and tuple(src.shape) == tuple(dx.shape) and src.is_contiguous() and src.dtype == BF16
        stats, rows = None, C.c_int32(0)
        bx = st = None
        if bn is not None and _BN_BWD_EPILOGUE and bn[0] is not None and bn[0].dtype == BF16 and tuple(bn[0].shape) == tuple(dx.shape) and bn[0].is_contiguous():
            bx, st = bn
            stats = torch.empty(_stats_elems(g, False), dtype=torch.float32, device=dy.device)
        L.check(lib.sat_conv2d_dgrad_bf16_fused(L.ptr(dy), L.ptr(_krsc(w)), L.ptr(dx), C.byref(g), L.ptr(src), L.ptr(mask), L.ptr(bx),
                                                L.ptr(st[2] if st is not None and len(st) > 2 else None), L.ptr(st[0] if st is not None else None),
                                                L.ptr(st[1] if st is not None else None), L.ptr(stats), C.byref(rows), L.stream_ptr()), "sat_conv2d_dgrad_bf16_fused")
        tiles = (stats, rows.value) if (stats is not None and rows.value > 0) else None
        return (dx, tiles) if bn is not None else dx
    if bn is not None:
        bx, st = bn
        if _BN_BWD_EPILOGUE and _is_bf(dy) and stride == 1 and bx is not None and bx.dtype == BF16 and tuple(bx.shape) == tuple(dx.shape) and bx.is_contiguous():
            stats = torch.empty(_stats_elems(g, False), dtype=torch.float32, device=dy.device)
            rows = C.c_int32(0)
            L.check(lib.sat_conv2d_dgrad_bf16_bnstats(L.ptr(dy), L.ptr(_krsc(w)), L.ptr(dx), C.byref(g), int(accumulate), L.ptr(bx), L.ptr(st[2] if len(st) > 2 else None),
                                                      L.ptr(st[0]), L.ptr(st[1]), L.ptr(stats), C.byref(rows), L.stream_ptr()), "sat_conv2d_dgrad_bf16_bnstats")
            return dx, ((stats, rows.value) if rows.value > 0 else None)
    fn = lib.sat_conv2d_dgrad_bf16 if _is_bf(dy) else lib.sat_conv2d_dgrad
    L.check(fn(L.ptr(dy), L.ptr(_krsc(w)), L.ptr(dx), C.byref(g), int(accumulate), L.stream_ptr()), "sat_conv2d_dgrad")
    return (dx, None) if bn is not None else dx


_slabs = {}


def _slab(device, nbytes, tag="main"):
    key = (device.index if device.index is not None else torch.cuda.current_device(), tag)
    cur = _slabs.get(key)
    if cur is None or cur.numel() * 4 < nbytes:
        cur = torch.empty(max(nbytes // 4, 4 << 20), dtype=torch.float32, device=device)
        _slabs[key] = cur
    return cur


#: weight gradients of the residual blocks on a second HIP stream: they feed nothing on the backward chain (only the optimizer), and
#: most of the chain's kernels leave matrix cores / HBM half idle, so the two streams fill each other's gaps
_WGRAD_STREAM = os.environ.get("SAT_WGRAD_STREAM", "1") != "0"
#: identity blocks: dx = dgrad + relu_mask(dout) inside the data-gradient launch (dev switch, SAT_DGRAD_JOIN=0 writes the masked gradient out)
_DGRAD_JOIN = os.environ.get("SAT_DGRAD_JOIN", "1") != "0"
#: projection blocks, forward: the shortcut's BatchNorm is applied inside the last BatchNorm's kernel (dev switch, SAT_FWD_RES_BN=0 writes it out)
_FWD_RES_BN = os.environ.get("SAT_FWD_RES_BN", "1") != "0"
_WGRAD_STREAMS = int(os.environ.get("SAT_WGRAD_STREAMS", "1"))          # side streams the launches are dealt to in turn
_WGRAD_SIDE_ONLY = int(os.environ.get("SAT_WGRAD_SIDE_ONLY", "0"))      # dev: which filters go to the side stream (0 = all)
#: the side stream is used from this many input pixels per batch on (N x H x W of the images).  Every fork / join costs the host ~40 us
#: (event record, stream wait, allocator bookkeeping of the tensors the side stream reads); with few images the weight-gradient kernels are
#: shorter than that and the step becomes host-bound: C1 (8 images of 256 x 256) steps in 4.2 ms with everything on the main stream and in
#: 4.9 ms with the side stream, C3's shard (32 images) in 15.5 vs 14.7 ms, C2 (128) in 22.0 vs 21.6 (tools/graph_step_time.py).
#: A version with ONE library call per fork and the tensors kept referenced until the join (no record_stream) measured SLOWER
#: (C3 14.7 -> 17.6 .. 21 ms: the allocator then cycles through far more live blocks per backward).
_WGRAD_SIDE_MIN_INPUT_PIXELS = int(os.environ.get("SAT_WGRAD_SIDE_MIN_INPUT_PIXELS", str(16 * 256 * 256)))
_side_streams = {}
_JOIN_LAG = None          # dev: a list collects (main-arrival, side-done) event pairs of every join


class _SideQueue:
    """Second stream for the weight gradients of one encoder backward.  ``launch(fn, *inputs)``: fn runs with the side stream current,
    after everything enqueued on the main stream so far; ``join()``: the main stream waits for everything launched here."""

    def __init__(self, device, enabled):
        self.enabled = bool(enabled)
        if not self.enabled:
            return
        self.main = torch.cuda.current_stream(device)
        key = device.index if device.index is not None else torch.cuda.current_device()
        n = max(1, _WGRAD_STREAMS)
        while len(_side_streams.setdefault(key, [])) < n:
            _side_streams[key].append(torch.cuda.Stream(device))
        self.sides = _side_streams[key][:n]
        self.side_ptrs = [C.c_void_p(sd.cuda_stream) for sd in self.sides]
        self.ev = torch.cuda.Event()
        self.turn = 0
        self.dirty = [False] * n

    def slot(self):
        """index of the side stream the next launch goes to (its split-K scratch is per stream)"""
        return self.turn % len(self.sides)

    def launch(self, fn, *inputs):
        """fn(stream) enqueues on the raw stream handle it is given (it must not allocate: its outputs are created by the caller)"""
        if not self.enabled:
            return fn(L.stream_ptr())
        i = self.slot(); side = self.sides[i]; self.turn += 1
        self.ev.record(self.main)              # one event serves every launch: a stream wait captures the record that precedes it
        side.wait_event(self.ev)
        out = fn(self.side_ptrs[i])            # no current-stream switch: the handle goes straight to the library (the switch was 10 us a launch)
        for t in inputs:                       # the caching allocator must not hand these blocks out again before the side stream is done
            if t is not None:
                t.record_stream(side)
        self.dirty[i] = True
        return out

    def join(self):
        if not self.enabled:
            return
        for i, side in enumerate(self.sides):
            if self.dirty[i]:
                timing = _JOIN_LAG is not None
                ev = torch.cuda.Event(enable_timing=timing); ev.record(side)
                if timing:          # dev (tools/join_lag.py): how long after the main stream arrived here the side stream finished
                    em = torch.cuda.Event(enable_timing=True); em.record(self.main)
                    _JOIN_LAG.append((em, ev))
                self.main.wait_event(ev)
                self.dirty[i] = False


_wgrad_slab_bytes = {}


def conv_wgrad(dy, x, w, stride, pad, stride_w=0, param=None, queue=None):
    """fp32 gradient of the (K,C,R,S) filter, whatever the activation storage.  ``param``: the filter parameter itself, when the
    gradient may be written to its data-parallel bucket slice (``_lib.grad_buffer``).  ``queue``: a ``_SideQueue`` - the launch then goes
    to its side stream (the result is allocated on the calling stream; the caller joins the queue before anything reads it)."""
    N, H, W, Cc = x.shape
    K, _, R, S = w.shape
    g = _geom(N, H, W, Cc, K, R, S, stride, pad, stride_w)
    lib = L.lib()
    gkey = (N, H, W, Cc, K, R, S, stride, pad, stride_w)
    nbytes = _wgrad_slab_bytes.get(gkey)
    if nbytes is None:
        nbytes = _wgrad_slab_bytes[gkey] = max(lib.sat_conv2d_wgrad_slab_bytes(C.byref(g)), 128 << 20)
    side = queue is not None and queue.enabled
    if side and _WGRAD_SIDE_ONLY:          # dev: 1 = only the 3x3 filters, 2 = only the 1x1 filters go to the side stream
        side = (R == 3) if _WGRAD_SIDE_ONLY == 1 else (R == 1)
    slab = _slab(x.device, nbytes, "side%d" % queue.slot() if side else "main")
    fn = lib.sat_conv2d_wgrad_bf16 if _is_bf(x) else lib.sat_conv2d_wgrad
    out = L.grad_buffer(param) if param is not None else None                 # (K,C,R,S), KRSC memory when the parameter is
    direct = out is not None and tuple(out.shape) == (K, Cc, R, S) and out.permute(0, 2, 3, 1).is_contiguous()
    dst = out if direct else torch.empty(K, R, S, Cc, dtype=torch.float32, device=x.device)       # KRSC

    def run(stream):
        L.check(fn(L.ptr(dy), L.ptr(x), L.ptr(dst), C.byref(g), L.ptr(slab), slab.numel(), stream), "sat_conv2d_wgrad")

    if side:
        queue.launch(run, dy, x, dst)
    else:
        run(L.stream_ptr())
    return out if direct else dst.permute(0, 3, 1, 2)                         # (K,C,R,S) view, channels_last memory


#: the stem's bn1 -> relu -> maxpool as one pass (tests switch it off to compare with the three separate kernels)
_FUSED_STEM_TAIL = os.environ.get("SAT_STEM_TAIL", "1") != "0"
#: bf16 stem as a 7x4 convolution over pixel pairs (include/sat_hip.h); off: the 8-channel layout with five zero channels
_STEM_PAIRS = os.environ.get("SAT_STEM_PAIRS", "1") != "0"

_tracked = []      # num_batches_tracked buffers touched by the running whole-encoder forward (bumped once, together)
_defer = [False]


_bn_scratch_bytes = {}
_bn_scratch_buf = {}


def _bn_scratch(device, rows, Cc):
    """BatchNorm partial-sum scratch: one buffer per (device, stream), grown on demand - consecutive BatchNorm calls on a stream are ordered,
    so they can share it (a ``torch.empty`` + a library call per BatchNorm call were ~0.6 ms of host time per C2 step)."""
    need = _bn_scratch_bytes.get((rows, Cc))
    if need is None:
        need = _bn_scratch_bytes[(rows, Cc)] = L.lib().sat_bn_scratch_bytes(rows, Cc) // 8 + 1
    key = (device.index, L.stream_ptr().value)
    buf = _bn_scratch_buf.get(key)
    if buf is None or buf.numel() < need:
        buf = _bn_scratch_buf[key] = torch.empty(max(need, 1 << 16), dtype=torch.float64, device=device)
    return buf


def bn_stats(x, bn, tiles):
    """Training-mode statistics of a BatchNorm from the producing convolution's row tiles, nothing normalised: (mean, invstd); the running
    statistics move as in nn.BatchNorm2d.  (bf16 storage; the caller normalises inside another kernel, see ``bn_fwd(res_bn=...)``.)"""
    lib = L.lib()
    Cc = x.shape[-1]; rows = x.numel() // Cc
    mean = torch.empty(Cc, dtype=torch.float32, device=x.device); invstd = torch.empty_like(mean)
    scratch = _bn_scratch(x.device, rows, Cc)
    mom = 0.1 if bn.momentum is None else float(bn.momentum)
    L.check(lib.sat_bn_train_fwd_tiles_bf16(L.ptr(x), rows, Cc, L.ptr(tiles[0]), int(tiles[1]), L.ptr(bn.weight), L.ptr(bn.bias), float(bn.eps),
                                            mom, L.ptr(bn.running_mean), L.ptr(bn.running_var), L.ptr(mean), L.ptr(invstd), None, 0, None, None,
                                            L.ptr(scratch), L.stream_ptr()), "sat_bn_train_fwd_tiles (statistics)")
    if _defer[0]:
        _tracked.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked += 1
    return mean, invstd


def bn_fwd(x, bn, residual=None, relu=True, training=True, want_mask=False, tiles=None, res_bn=None):
    """BatchNorm (+ residual) (+ ReLU) of a (..., C) NHWC tensor.  Returns (y, stats); in training mode stats =
    (mean, invstd[, relu_mask]) -- with ``want_mask`` the sign mask of the output (1 bit per element) that ``bn_bwd``
    reads instead of y.  ``res_bn`` = ((mean, invstd), module): ``residual`` is the RAW input of that train-mode BatchNorm (the projection
    shortcut), normalised on the fly (``sat_bn_train_fwd_tiles_bf16_resbn``)."""
    lib = L.lib()
    Cc = x.shape[-1]; rows = x.numel() // Cc
    y = torch.empty_like(x)
    dt = int(_is_bf(x))
    if training:
        mask = torch.empty(x.numel() // 8, dtype=torch.uint8, device=x.device) if (want_mask and relu and Cc % 8 == 0) else None
        mean = torch.empty(Cc, dtype=torch.float32, device=x.device); invstd = torch.empty_like(mean)
        scratch = _bn_scratch(x.device, rows, Cc)
        mom = 0.1 if bn.momentum is None else float(bn.momentum)
        if res_bn is not None:
            assert tiles is not None and dt == 1 and residual is not None
            (rmean, rinv), rmod = res_bn
            L.check(lib.sat_bn_train_fwd_tiles_bf16_resbn(L.ptr(x), rows, Cc, L.ptr(tiles[0]), int(tiles[1]), L.ptr(bn.weight), L.ptr(bn.bias), float(bn.eps),
                                                          mom, L.ptr(bn.running_mean), L.ptr(bn.running_var), L.ptr(mean), L.ptr(invstd), L.ptr(residual),
                                                          L.ptr(rmean), L.ptr(rinv), L.ptr(rmod.weight), L.ptr(rmod.bias), int(relu), L.ptr(y), L.ptr(mask),
                                                          L.ptr(scratch), L.stream_ptr()), "sat_bn_train_fwd_tiles_bf16_resbn")
        elif tiles is not None and dt == 1:          # statistics already reduced per row tile by the producing convolution
            L.check(lib.sat_bn_train_fwd_tiles_bf16(L.ptr(x), rows, Cc, L.ptr(tiles[0]), int(tiles[1]), L.ptr(bn.weight), L.ptr(bn.bias), float(bn.eps),
                                                    mom, L.ptr(bn.running_mean), L.ptr(bn.running_var), L.ptr(mean), L.ptr(invstd), L.ptr(residual),
                                                    int(relu), L.ptr(y), L.ptr(mask), L.ptr(scratch), L.stream_ptr()), "sat_bn_train_fwd_tiles")
        else:
            L.check(lib.sat_bn_train_fwd_t(dt, L.ptr(x), rows, Cc, L.ptr(bn.weight), L.ptr(bn.bias), float(bn.eps), mom, L.ptr(bn.running_mean),
                                           L.ptr(bn.running_var), L.ptr(mean), L.ptr(invstd), L.ptr(residual), int(relu), L.ptr(y), L.ptr(mask),
                                           L.ptr(scratch), L.stream_ptr()), "sat_bn_train_fwd")
        if _defer[0]:
            _tracked.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked += 1
        return y, ((mean, invstd) if mask is None else (mean, invstd, mask))
    L.check(lib.sat_bn_eval_fwd_t(dt, L.ptr(x), rows, Cc, L.ptr(bn.running_mean), L.ptr(bn.running_var), float(bn.eps), L.ptr(bn.weight),
                                  L.ptr(bn.bias), L.ptr(residual), int(relu), L.ptr(y), L.stream_ptr()), "sat_bn_eval_fwd")
    return y, None


def bn_bwd(dy, x, y, stats, bn, relu, dres=None, dres_accumulate=False, tiles=None):
    """``tiles``: backward statistics per row tile from the data-gradient launch that wrote ``dy`` (``conv_dgrad(..., bn=...)``)."""
    lib = L.lib()
    Cc = x.shape[-1]; rows = x.numel() // Cc
    dx = torch.empty_like(x)
    dgamma, dbeta = L.grad_buffer(bn.weight), L.grad_buffer(bn.bias)
    scratch = _bn_scratch(x.device, rows, Cc)
    if tiles is not None and _is_bf(x) and (not relu or len(stats) > 2):
        L.check(lib.sat_bn_train_bwd_tiles_bf16(L.ptr(dy), L.ptr(x), rows, Cc, L.ptr(tiles[0]), int(tiles[1]), L.ptr(stats[0]), L.ptr(stats[1]), L.ptr(bn.weight),
                                                int(relu), L.ptr(dx), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dres), int(dres_accumulate),
                                                L.ptr(stats[2] if len(stats) > 2 else None), L.ptr(scratch), L.stream_ptr()), "sat_bn_train_bwd_tiles_bf16")
        return dx, dgamma, dbeta
    L.check(lib.sat_bn_train_bwd_t(int(_is_bf(x)), L.ptr(dy), L.ptr(x), L.ptr(y), rows, Cc, L.ptr(stats[0]), L.ptr(stats[1]), L.ptr(bn.weight),
                                   int(relu), L.ptr(dx), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dres), int(dres_accumulate),
                                   L.ptr(stats[2] if len(stats) > 2 else None), L.ptr(scratch), L.stream_ptr()), "sat_bn_train_bwd")
    return dx, dgamma, dbeta


def stem_tail_fwd(x, bn, tiles=None):
    """Training-mode bn1 -> relu -> maxpool(3, 2, 1) of the stem convolution's NHWC output in one pass over x
    (``sat_stem_tail_fwd_t``): returns (pooled, (mean, invstd, argmax)).  The statistics come from the convolution's
    epilogue tiles when given, else from a pass over x; running statistics are updated like nn.BatchNorm2d."""
    lib = L.lib()
    N, H, W, Cc = x.shape
    rows = N * H * W
    dt = int(_is_bf(x))
    mean = torch.empty(Cc, dtype=torch.float32, device=x.device); invstd = torch.empty_like(mean)
    scratch = _bn_scratch(x.device, rows, Cc)
    mom = 0.1 if bn.momentum is None else float(bn.momentum)
    if tiles is not None and dt == 1:
        L.check(lib.sat_bn_train_fwd_tiles_bf16(L.ptr(x), rows, Cc, L.ptr(tiles[0]), int(tiles[1]), L.ptr(bn.weight), L.ptr(bn.bias), float(bn.eps),
                                                mom, L.ptr(bn.running_mean), L.ptr(bn.running_var), L.ptr(mean), L.ptr(invstd), None, 1, None, None,
                                                L.ptr(scratch), L.stream_ptr()), "sat_bn_train_fwd_tiles (statistics)")
    else:
        L.check(lib.sat_bn_train_fwd_t(dt, L.ptr(x), rows, Cc, L.ptr(bn.weight), L.ptr(bn.bias), float(bn.eps), mom, L.ptr(bn.running_mean),
                                       L.ptr(bn.running_var), L.ptr(mean), L.ptr(invstd), None, 1, None, None, L.ptr(scratch), L.stream_ptr()),
                "sat_bn_train_fwd (statistics)")
    if _defer[0]:
        _tracked.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked += 1
    P, Q = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty(N, P, Q, Cc, dtype=x.dtype, device=x.device)
    amax = torch.empty(N, P, Q, Cc, dtype=torch.uint8, device=x.device)
    L.check(lib.sat_stem_tail_fwd_t(dt, L.ptr(x), N, H, W, Cc, L.ptr(mean), L.ptr(invstd), L.ptr(bn.weight), L.ptr(bn.bias), L.ptr(y), L.ptr(amax),
                                    L.stream_ptr()), "sat_stem_tail_fwd")
    return y, (mean, invstd, amax)


def stem_tail_bwd(dy_pool, x, stats, bn):
    """gradient of ``stem_tail_fwd``: (dx, dgamma, dbeta) from the pooled gradient"""
    lib = L.lib()
    N, H, W, Cc = x.shape
    dx = torch.empty_like(x)
    dgamma, dbeta = L.grad_buffer(bn.weight), L.grad_buffer(bn.bias)
    scratch = _bn_scratch(x.device, N * H * W, Cc)
    L.check(lib.sat_stem_tail_bwd_t(int(_is_bf(x)), L.ptr(dy_pool), L.ptr(stats[2]), L.ptr(x), N, H, W, Cc, L.ptr(stats[0]), L.ptr(stats[1]), L.ptr(bn.weight),
                                    L.ptr(bn.bias), L.ptr(dx), L.ptr(dgamma), L.ptr(dbeta), L.ptr(scratch), L.stream_ptr()), "sat_stem_tail_bwd")
    return dx, dgamma, dbeta


def colsum(x2d, out=None):
    rows, cols = x2d.shape
    out = torch.empty(cols, dtype=torch.float32, device=x2d.device) if out is None else out
    scratch = torch.empty(((rows + 255) // 256) * cols, dtype=torch.float32, device=x2d.device)
    L.check(L.lib().sat_colsum(L.ptr(x2d), x2d.stride(0), rows, cols, L.ptr(out), L.ptr(scratch), L.stream_ptr()), "sat_colsum")
    return out


# ----------------------------------------------------------------------------- whole-network forward / backward
class _Rec:
    __slots__ = ("kind", "blk", "x", "c1", "a1", "s1", "c2", "a2", "s2", "c3", "s3", "cd", "sd", "idn", "out", "w2")


def _block_fwd(blk, x, training, W=None):
    """W maps a filter parameter to the tensor the kernels read (its bf16 copy in bf16 mode)."""
    W = W or (lambda p: p)
    r = _Rec(); r.kind, r.blk, r.x = blk.kind, blk, x
    r.cd = r.sd = None
    # training: the convolution's epilogue leaves the BatchNorm statistics of its output (bf16 storage), see conv_fwd_stats
    conv = conv_fwd_stats if training else (lambda *a: (conv_fwd(*a), None))
    if blk.kind == "basic":
        r.c1, tl = conv(x, W(blk.conv1.weight), blk.stride, 1); r.a1, r.s1 = bn_fwd(r.c1, blk.bn1, None, True, training, want_mask=True, tiles=tl)
        last, tl = conv(r.a1, W(blk.conv2.weight), 1, 1); r.c2 = last
    else:
        r.c1, tl = conv(x, W(blk.conv1.weight), 1, 0); r.a1, r.s1 = bn_fwd(r.c1, blk.bn1, None, True, training, want_mask=True, tiles=tl)
        r.w2 = W(blk.conv2.weight) if blk.groups == 1 else grouped_dense(W(blk.conv2.weight), blk.groups)      # resnext: block-diagonal dense filter
        r.c2, tl = conv(r.a1, r.w2, blk.stride, 1); r.a2, r.s2 = bn_fwd(r.c2, blk.bn2, None, True, training, want_mask=True, tiles=tl)
        last, tl = conv(r.a2, W(blk.conv3.weight), 1, 0); r.c3 = last
    res_bn = None
    if blk.downsample is not None:
        r.cd, tld = conv(x, W(blk.downsample[0].weight), blk.stride, 0)
        if _FWD_RES_BN and training and tl is not None and tld is not None and _is_bf(r.cd):
            # projection shortcut: only its statistics are taken here; the last BatchNorm's kernel normalises r.cd on the fly
            r.sd = bn_stats(r.cd, blk.downsample[1], tld)
            r.idn, res_bn = r.cd, (r.sd, blk.downsample[1])
        else:
            r.idn, r.sd = bn_fwd(r.cd, blk.downsample[1], None, False, training, tiles=tld)
    else:
        r.idn = x
    last_bn = blk.bn2 if blk.kind == "basic" else blk.bn3
    r.out, st = bn_fwd(last, last_bn, r.idn, True, training, want_mask=True, tiles=tl, res_bn=res_bn)
    if blk.kind == "basic":
        r.s2 = st
    else:
        r.s3 = st
    return r


def _block_bwd(r, dout, grads, need_dx, W=None, dout_tiles=None, prev=None, queue=None):
    """Backward of one residual block.  ``dout_tiles``: backward statistics of this block's last BatchNorm that came with ``dout``
    (produced by the block behind it); ``prev``: the record of the block in front, whose last BatchNorm's statistics the launch that
    writes this block's input gradient can produce.  Returns (dx, tiles for ``prev``'s last BatchNorm or None)."""
    W = W or (lambda p: p)
    blk = r.blk
    last_stats = r.s2 if blk.kind == "basic" else r.s3
    # identity blocks in bf16 storage: the masked gradient of the residual branch is not written out at all - the launch that produces the
    # block's input gradient reads dout and the ReLU sign bits itself (one pass over the block-wide tensor less)
    have_mask = _DGRAD_JOIN and _is_bf(dout) and last_stats is not None and len(last_stats) > 2 and last_stats[2] is not None and dout.is_contiguous()
    join = have_mask and blk.downsample is None and need_dx
    # projection blocks: the shortcut's BatchNorm backward masks dout with the same sign bits on the fly (its "ReLU" is the block's)
    mask_ds = have_mask and blk.downsample is not None
    g = None if (join or mask_ds) else torch.empty_like(r.out)    # gradient of the residual branch (= dout masked by the final ReLU)
    if blk.kind == "basic":
        dx2, grads[blk.bn2.weight], grads[blk.bn2.bias] = bn_bwd(dout, r.c2, r.out, r.s2, blk.bn2, True, dres=g, tiles=dout_tiles)
        grads[blk.conv2.weight] = conv_wgrad(dx2, r.a1, blk.conv2.weight, 1, 1, param=blk.conv2.weight, queue=queue)
        da1, t1 = conv_dgrad(dx2, W(blk.conv2.weight), r.a1.shape, 1, 1, bn=(r.c1, r.s1))
        first_w, first_stride, first_pad = blk.conv1.weight, blk.stride, 1
    else:
        dx3, grads[blk.bn3.weight], grads[blk.bn3.bias] = bn_bwd(dout, r.c3, r.out, r.s3, blk.bn3, True, dres=g, tiles=dout_tiles)
        grads[blk.conv3.weight] = conv_wgrad(dx3, r.a2, blk.conv3.weight, 1, 0, param=blk.conv3.weight, queue=queue)
        da2, t2 = conv_dgrad(dx3, W(blk.conv3.weight), r.a2.shape, 1, 0, bn=(r.c2, r.s2))
        dx2, grads[blk.bn2.weight], grads[blk.bn2.bias] = bn_bwd(da2, r.c2, r.a2, r.s2, blk.bn2, True, tiles=t2)
        if blk.groups == 1:
            grads[blk.conv2.weight] = conv_wgrad(dx2, r.a1, blk.conv2.weight, blk.stride, 1, param=blk.conv2.weight, queue=queue)
        else:          # resnext: gradient of the dense block-diagonal filter (main stream), then its diagonal blocks
            grads[blk.conv2.weight] = grouped_grad(conv_wgrad(dx2, r.a1, r.w2, blk.stride, 1), blk.conv2.weight, blk.groups)
        da1, t1 = conv_dgrad(dx2, r.w2, r.a1.shape, blk.stride, 1, bn=(r.c1, r.s1))
        first_w, first_stride, first_pad = blk.conv1.weight, 1, 0
    dx1, grads[blk.bn1.weight], grads[blk.bn1.bias] = bn_bwd(da1, r.c1, r.a1, r.s1, blk.bn1, True, tiles=t1)
    grads[first_w] = conv_wgrad(dx1, r.x, first_w, first_stride, first_pad, param=first_w, queue=queue)
    if blk.downsample is not None:
        if mask_ds:
            dxd, grads[blk.downsample[1].weight], grads[blk.downsample[1].bias] = bn_bwd(dout, r.cd, None, (r.sd[0], r.sd[1], last_stats[2]), blk.downsample[1], True)
        else:
            dxd, grads[blk.downsample[1].weight], grads[blk.downsample[1].bias] = bn_bwd(g, r.cd, None, r.sd, blk.downsample[1], False)
        grads[blk.downsample[0].weight] = conv_wgrad(dxd, r.x, blk.downsample[0].weight, blk.stride, 0, param=blk.downsample[0].weight, queue=queue)
        if not need_dx:
            return None, None
        dx = conv_dgrad(dx1, W(first_w), r.x.shape, first_stride, first_pad)
        # the strided 1x1 shortcut only reaches the even pixels: accumulate it on top (the other parity classes are skipped)
        return conv_dgrad(dxd, W(blk.downsample[0].weight), r.x.shape, blk.stride, 0, out=dx, accumulate=True), None
    if not need_dx:
        return None, None
    # identity path + conv path; the launch that writes the sum also leaves the statistics of the previous block's last BatchNorm
    pbn = None
    if prev is not None and prev.out is r.x:
        pbn = (prev.c3, prev.s3) if prev.kind == "bottleneck" else (prev.c2, prev.s2)
    if join:
        if pbn is not None:
            return conv_dgrad(dx1, W(first_w), r.x.shape, first_stride, first_pad, bn=pbn, add=(dout, last_stats[2]))
        return conv_dgrad(dx1, W(first_w), r.x.shape, first_stride, first_pad, add=(dout, last_stats[2])), None
    if pbn is not None:
        return conv_dgrad(dx1, W(first_w), r.x.shape, first_stride, first_pad, out=g, accumulate=True, bn=pbn)
    return conv_dgrad(dx1, W(first_w), r.x.shape, first_stride, first_pad, out=g, accumulate=True), None


#: the residual-block loop inside the library (csrc/encoder_loop.hip): one call per direction for the whole trunk instead of ~24 calls per block.
#: Training, bf16 storage, one process (the data-parallel exchange wants a notification per ResNet stage: it keeps the per-layer driver).
_BLOCK_LOOP = os.environ.get("SAT_BLOCK_LOOP", "1") != "0"


def _loop_blocks(enc):
    return [blk for li in (5, 6, 7, 8) for blk in enc[li]]


def _loop_eligible(enc, x, training, bf):
    if not (_BLOCK_LOOP and training and bf and _is_bf(x) and x.is_contiguous()):
        return False
    blocks = _loop_blocks(enc)
    if len(blocks) > 64:
        return False
    for blk in blocks:
        if blk.groups != 1 or blk.conv1.in_channels % 8 or blk.cout % 8 or blk.conv1.out_channels % 8:
            return False
    return True


def _bn_ptrs(d, i, bn):
    d.gamma[i], d.beta[i] = bn.weight.data_ptr(), bn.bias.data_ptr()
    d.running_mean[i], d.running_var[i] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
    d.eps[i], d.momentum[i] = float(bn.eps), 0.1 if bn.momentum is None else float(bn.momentum)


def _loop_descs(enc, x, Wt):
    """descriptor table of the trunk's blocks for an input map x (N, H, W, C): geometry, filters as the kernels read them, BatchNorm pointers"""
    blocks = _loop_blocks(enc)
    arr = (L.BlockDesc * len(blocks))()
    N, H, W, _ = x.shape
    bns = []
    for d, blk in zip(arr, blocks):
        d.kind = 0 if blk.kind == "basic" else 1
        d.stride, d.cin, d.cout, d.has_ds = blk.stride, blk.conv1.in_channels, blk.cout, int(blk.downsample is not None)
        d.mid = blk.conv1.out_channels if blk.kind != "basic" else blk.cout
        d.N, d.H, d.W = N, H, W
        d.fwd_res_bn, d.dgrad_join, d.bn_bwd_epilogue = int(_FWD_RES_BN), int(_DGRAD_JOIN), int(_BN_BWD_EPILOGUE)
        d.w1, d.w2 = Wt(blk.conv1.weight).data_ptr(), Wt(blk.conv2.weight).data_ptr()
        _bn_ptrs(d, 0, blk.bn1); _bn_ptrs(d, 1, blk.bn2); bns += [blk.bn1, blk.bn2]
        if blk.kind != "basic":
            d.w3 = Wt(blk.conv3.weight).data_ptr(); _bn_ptrs(d, 2, blk.bn3); bns.append(blk.bn3)
        if blk.downsample is not None:
            d.wd = Wt(blk.downsample[0].weight).data_ptr(); _bn_ptrs(d, 3, blk.downsample[1]); bns.append(blk.downsample[1])
        H, W = _out_hw(H, W, 3, 3, blk.stride, 1)
    return arr, blocks, bns, (N, H, W, blocks[-1].cout)


def _loop_scratch(device, arr, blocks):
    """BatchNorm scratch for the largest map of the trunk"""
    best = None
    for d, blk in zip(arr, blocks):
        for rows, Cc in ((d.N * d.H * d.W, max(d.cin, d.mid)), (d.N * d.H * d.W, d.cout)):
            need = L.lib().sat_bn_scratch_bytes(rows, Cc)
            if best is None or need > best[0]:
                best = (need, rows, Cc)
    return _bn_scratch(device, best[1], best[2])


def _loop_slab_bytes(arr):
    """split-K scratch that serves every weight-gradient launch of the trunk"""
    lib = L.lib()
    need = 128 << 20
    for d in arr:
        P, Q = _out_hw(d.H, d.W, 3, 3, d.stride, 1)
        if d.kind:
            geoms = [(d.N, d.H, d.W, d.cin, d.mid, 1, 1, 1, 0), (d.N, d.H, d.W, d.mid, d.mid, 3, 3, d.stride, 1), (d.N, P, Q, d.mid, d.cout, 1, 1, 1, 0)]
        else:
            geoms = [(d.N, d.H, d.W, d.cin, d.cout, 3, 3, d.stride, 1), (d.N, P, Q, d.cout, d.cout, 3, 3, 1, 1)]
        if d.has_ds:
            geoms.append((d.N, d.H, d.W, d.cin, d.cout, 1, 1, d.stride, 0))
        for g in geoms:
            need = max(need, lib.sat_conv2d_wgrad_slab_bytes(C.byref(_geom(*g))))
    return need


def _arena_view(arena, ptr, shape):
    off = ptr - arena.data_ptr()
    n = 1
    for v in shape:
        n *= v
    return arena[off:off + 2 * n].view(BF16).view(*shape)


def _krsc_dest(param):
    """(destination pointer tensor in KRSC fp32 memory, gradient tensor to hand to autograd) for a filter parameter"""
    K, Cc, R, S = param.shape
    out = L.grad_buffer(param)
    if tuple(out.shape) == (K, Cc, R, S) and out.permute(0, 2, 3, 1).is_contiguous():
        return out, out
    dst = torch.empty(K, R, S, Cc, dtype=torch.float32, device=param.device)
    return dst, dst.permute(0, 3, 1, 2)


def _weight_reader(bf):
    """Wt(p): the tensor the kernels read for filter parameter p (its bf16 copy in bf16 mode), remembered for one forward / backward pair"""
    if not bf:
        return lambda p: p
    wcopy = {}

    def Wt(p):
        c = wcopy.get(p)
        if c is None:
            # the copy lives on the parameter: FusedOptimizer rewrites it in its update kernel, so a training loop casts
            # each filter once; any other in-place change of p bumps p._version and the copy is remade here
            c = getattr(p, "_sat_bf16_shadow", None)
            if c is None or getattr(p, "_sat_shadow_version", -1) != p._version or c.device != p.device:
                c = cast_bf16(p)
                p._sat_bf16_shadow, p._sat_shadow_version = c, p._version
            wcopy[p] = c
        return c
    return Wt


def _head_fwd(enc, x, t, Wt, bf):
    """what follows the trunk: the optional 1x1 projection to ``encoder_dim`` (model.py:53) and the optional ``encoder_size`` resize; x NHWC"""
    lib = L.lib()
    st = L.stream_ptr()
    t["trunk"] = x
    if enc.proj is not None:
        Nn, Hh, Ww, Cc = x.shape
        D = enc.proj.out_channels
        if bf:        # 1x1 projection: bf16 x bf16 -> fp32 annotations (+bias) on the bf16 MFMA kernel
            from .decoder import gemm
            y = torch.empty(Nn, Hh, Ww, D, dtype=torch.float32, device=x.device)
            gemm(x.view(-1, Cc), Wt(enc.proj.weight).view(D, Cc), out=y.view(-1, D), bias=enc.proj.bias, epi=1, bf16_mfma=True)
            x = y
        else:
            x = conv_fwd(x, enc.proj.weight, 1, 0, enc.proj.bias)
    elif bf:          # no projection (encoder_dim None or equal to the trunk width, model.py:56-57): the annotations are the trunk output as fp32
        y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        L.check(lib.sat_cast_bf16_to_f32(L.ptr(x), L.ptr(y), x.numel(), st), "sat_cast_bf16_to_f32")
        x = y
    t["proj_out"] = x
    if enc.out_size is not None and enc.out_size != x.shape[1]:
        Nn, Hh, Ww, Cc = x.shape
        y = torch.empty(Nn, enc.out_size, enc.out_size, Cc, dtype=torch.float32, device=x.device)
        L.check(lib.sat_resize_fwd(L.ptr(x), L.ptr(y), Nn, Hh, Ww, Cc, enc.out_size, enc.out_size, st), "sat_resize_fwd")
        x = y
    return x


def _head_bwd(enc, t, dann, grads, Wt, bf):
    """gradient of ``_head_fwd``: fills the projection's gradients, returns the gradient of the trunk output (NHWC, the trunk's storage type) or
    None when the trunk is frozen"""
    lib = L.lib()
    st = L.stream_ptr()
    d = dann.permute(0, 2, 3, 1).contiguous()                                 # NHWC fp32 (no copy when it already is)
    if enc.out_size is not None and enc.out_size != t["proj_out"].shape[1]:
        Nn, Hh, Ww, Cc = t["proj_out"].shape
        dx = torch.empty_like(t["proj_out"])
        L.check(lib.sat_resize_bwd(L.ptr(d), L.ptr(dx), Nn, Hh, Ww, Cc, enc.out_size, enc.out_size, st), "sat_resize_bwd")
        d = dx
    if enc.proj is not None:
        grads[enc.proj.bias] = colsum(d.reshape(-1, d.shape[-1]), out=L.grad_buffer(enc.proj.bias))
        if bf:
            from .decoder import gemm
            D = enc.proj.out_channels; Cc = t["trunk"].shape[-1]
            db = cast_bf16(d.reshape(-1, D))
            dw = L.grad_buffer(enc.proj.weight)                                   # (D, Cc, 1, 1): D x Cc row-major underneath
            gemm(db, t["trunk"].view(-1, Cc), amode=1, bmode=1, out=dw.view(D, Cc), slab=_slab(d.device, 128 << 20), bf16_mfma=True)
            grads[enc.proj.weight] = dw
            if enc.trunk_trainable:
                dtr = torch.empty(t["trunk"].shape, dtype=BF16, device=d.device)
                gemm(db, Wt(enc.proj.weight).view(D, Cc), amode=0, bmode=1, out=dtr.view(-1, Cc), bf16_mfma=True)
                d = dtr
        else:
            grads[enc.proj.weight] = conv_wgrad(d, t["trunk"], enc.proj.weight, 1, 0, param=enc.proj.weight)
            d = conv_dgrad(d, enc.proj.weight, t["trunk"].shape, 1, 0) if enc.trunk_trainable else None
    elif bf and enc.trunk_trainable:
        d = cast_bf16(d.reshape(-1, d.shape[-1])).view(d.shape)
    return d


class EncoderFn(torch.autograd.Function):
    """img (B,3,H,W) fp32 in [0,1] -> annotations (B,D,h,w) fp32 (NHWC memory).  ``enc.precision``:
    "fp32" = fp32 activations on the exact fp32 MFMA kernel (parity mode); "bf16" = bf16 activations and
    filter copies on the bf16 MFMA kernel, fp32 statistics / gradients of parameters / master weights."""

    @staticmethod
    def forward(ctx, img, enc, *params):
        try:
            return EncoderFn._forward(ctx, img, enc, *params)
        finally:
            _defer[0] = False

    @staticmethod
    def _forward(ctx, img, enc, *params):
        lib = L.lib()
        L.require_gpu(img, *params)
        if img.dim() != 4 or img.shape[1] != 3 or img.dtype != torch.float32:
            raise ValueError("encoder input must be (B,3,H,W) fp32 in [0,1]")
        img = img.contiguous()
        training = enc.training
        _defer[0] = True; del _tracked[:]
        bf = enc.precision == "bf16"
        adt = BF16 if bf else torch.float32
        N, _, H, W = img.shape
        st = L.stream_ptr()
        t = {}
        Wt = _weight_reader(bf)

        mean = (C.c_float * 3)(*enc[0].mean); std = (C.c_float * 3)(*enc[0].std)
        conv1 = enc[1]
        w3 = _krsc(conv1.weight)                                                   # (64,3,7,7), memory 64,7,7,3
        cpad = 8 if bf else 4
        pairs = bf and _STEM_PAIRS and W % 2 == 0 and tuple(conv1.weight.shape[1:]) == (3, 7, 7)
        t["stem_pairs"] = pairs
        if pairs:
            # two horizontally adjacent pixels of the zero-padded 4-channel image = one 16-byte "pixel" of 8 channels: the 7x7
            # stride-2 stem becomes a 7x4 convolution with strides (2, 1) and no padding - 224 products per output, not 392
            x0 = torch.empty(N, H + 6, (W + 6) // 2, 8, dtype=adt, device=img.device)
            wp = torch.empty(conv1.out_channels, 8, 7, 4, dtype=adt, device=img.device).contiguous(memory_format=torch.channels_last)
            L.check(lib.sat_image_normalize_nhwc4_padded_bf16(L.ptr(img), L.ptr(x0), N, H, W, mean, std, st), "sat_image_normalize_nhwc4_padded_bf16")
            L.check(lib.sat_stem_filter_pairs(L.ptr(w3), L.ptr(wp), conv1.out_channels, st), "sat_stem_filter_pairs")
            stem = dict(stride=2, pad=0, stride_w=1)
        else:
            x0 = torch.empty(N, H, W, cpad, dtype=adt, device=img.device)
            wp = torch.empty(conv1.out_channels, cpad, 7, 7, dtype=adt, device=img.device).contiguous(memory_format=torch.channels_last)
            if bf:
                L.check(lib.sat_image_normalize_nhwc8_bf16(L.ptr(img), L.ptr(x0), N, H, W, mean, std, st), "sat_image_normalize_nhwc8_bf16")
                L.check(lib.sat_stem_filter_pad(L.ptr(w3), L.ptr(wp), conv1.out_channels * 49, st), "sat_stem_filter_pad")
            else:
                L.check(lib.sat_image_normalize_nhwc4(L.ptr(img), L.ptr(x0), N, H, W, mean, std, st), "sat_image_normalize_nhwc4")
                L.check(lib.sat_pad_channels_3to4(L.ptr(w3), L.ptr(wp), conv1.out_channels * 49, 0, st), "sat_pad_channels_3to4")
            stem = dict(stride=2, pad=3, stride_w=0)
        t["x0"], t["wp"] = x0, wp
        t["c0"], tl = (conv_fwd_stats(x0, wp, stem["stride"], stem["pad"], stride_w=stem["stride_w"]) if training
                       else (conv_fwd(x0, wp, stem["stride"], stem["pad"], stride_w=stem["stride_w"]), None))
        if training and _FUSED_STEM_TAIL:      # bn1 + relu + maxpool in one pass: the full-size activation is never written
            t["p0"], t["s0"] = stem_tail_fwd(t["c0"], enc[2], tiles=tl)
        else:
            t["a0"], t["s0"] = bn_fwd(t["c0"], enc[2], None, True, training, want_mask=True, tiles=tl)
            Nn, Hh, Ww, Cc = t["a0"].shape
            P, Q = (Hh + 2 - 3) // 2 + 1, (Ww + 2 - 3) // 2 + 1
            t["p0"] = torch.empty(Nn, P, Q, Cc, dtype=adt, device=img.device)
            t["amax"] = torch.empty(Nn, P, Q, Cc, dtype=torch.uint8, device=img.device)
            L.check(lib.sat_maxpool3x3s2_fwd_t(int(bf), L.ptr(t["a0"]), L.ptr(t["p0"]), L.ptr(t["amax"]), Nn, Hh, Ww, Cc, st), "sat_maxpool3x3s2_fwd")
        x = t["p0"]
        recs = []
        if _loop_eligible(enc, x, training, bf):
            # the whole trunk in one library call over one arena (csrc/encoder_loop.hip); bit-identical to the per-layer driver below
            arr, blocks, bns, oshape = _loop_descs(enc, x, Wt)
            nbytes = lib.sat_encoder_blocks_arena_bytes(arr, len(blocks))
            arena = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            outp = C.c_void_p(0)
            L.check(lib.sat_encoder_blocks_fwd(arr, len(blocks), L.ptr(x), L.ptr(arena), nbytes, L.ptr(_loop_scratch(x.device, arr, blocks)), C.byref(outp), st),
                    "sat_encoder_blocks_fwd")
            _tracked.extend(bn.num_batches_tracked for bn in bns)
            t["loop"] = (arr, blocks, arena, nbytes, x)
            x = _arena_view(arena, outp.value, oshape)
        else:
            for li in (5, 6, 7, 8):
                for blk in enc[li]:
                    r = _block_fwd(blk, x, training, Wt)
                    recs.append(r); x = r.out
        x = _head_fwd(enc, x, t, Wt, bf)
        _defer[0] = False
        if _tracked:
            torch._foreach_add_(_tracked, 1)
            del _tracked[:]
        ctx.t, ctx.recs, ctx.enc, ctx.Wt, ctx.bf = t, recs, enc, Wt, bf
        ctx.params = params
        return x.permute(0, 3, 1, 2)            # (B, D, h, w) view over NHWC memory

    @staticmethod
    def backward(ctx, dann):
        lib = L.lib()
        enc, t, recs, Wt, bf = ctx.enc, ctx.t, ctx.recs, ctx.Wt, ctx.bf
        st = L.stream_ptr()
        grads = {}
        cb = getattr(enc, "grad_ready", None)          # optional hook: called with {param: grad} as stages finish
        d = _head_bwd(enc, t, dann, grads, Wt, bf)
        if enc.trunk_trainable:
            n_per_stage = [len(enc[li]) for li in (5, 6, 7, 8)]
            bounds, acc = [], 0
            for n in n_per_stage:
                acc += n; bounds.append(acc)
            tiles = None
            x0s = t["x0"].shape
            queue = _SideQueue(d.device, _WGRAD_STREAM and bf and x0s[0] * x0s[1] * x0s[2] * (2 if t.get("stem_pairs") else 1) >= _WGRAD_SIDE_MIN_INPUT_PIXELS)
            if t.get("loop") is not None:
                arr, blocks, arena, nbytes, xin = t["loop"]
                dests, per_block = [], []
                for dsc, blk in zip(arr, blocks):          # where the parameter gradients go (bucket slices of the data-parallel exchange, or new memory)
                    mine = {}
                    convs = [(blk.conv1, "dw1"), (blk.conv2, "dw2")] + ([(blk.conv3, "dw3")] if blk.kind != "basic" else []) + \
                            ([(blk.downsample[0], "dwd")] if blk.downsample is not None else [])
                    for conv, field in convs:
                        dst, gr = _krsc_dest(conv.weight)
                        setattr(dsc, field, dst.data_ptr()); mine[conv.weight] = gr; dests.append(dst)
                    bnl = [(0, blk.bn1), (1, blk.bn2)] + ([(2, blk.bn3)] if blk.kind != "basic" else []) + ([(3, blk.downsample[1])] if blk.downsample is not None else [])
                    for i, bn in bnl:
                        gw, gb = L.grad_buffer(bn.weight), L.grad_buffer(bn.bias)
                        dsc.dgamma[i], dsc.dbeta[i] = gw.data_ptr(), gb.data_ptr(); mine[bn.weight], mine[bn.bias] = gw, gb
                    per_block.append(mine)
                slab_b = _loop_slab_bytes(arr)
                slab_m = _slab(d.device, slab_b, "main")
                side_ptr = ev_ptr = None; slab_s = None
                if queue.enabled:
                    slab_s = _slab(d.device, slab_b, "side0")
                    queue.ev.record(queue.main)
                    side_ptr, ev_ptr = queue.side_ptrs[0], C.c_void_p(queue.ev.cuda_event)
                scratch = _loop_scratch(d.device, arr, blocks)
                # one call for the whole trunk, or one per ResNet stage when the data-parallel exchange wants to start a bucket per stage
                starts = [0] + bounds[:3] if cb is not None else [0]
                ranges = [(starts[k], (starts[k + 1] if k + 1 < len(starts) else len(blocks)) - 1) for k in range(len(starts))]
                dcur = d.contiguous(); dptr = dcur.data_ptr()
                tptr, trows = C.c_void_p(0), C.c_int32(0)
                for first, last in reversed(ranges):
                    dxp = C.c_void_p(0)
                    L.check(lib.sat_encoder_blocks_bwd(arr, len(blocks), first, last, L.ptr(xin), L.ptr(arena), nbytes, dptr, tptr, trows.value, L.ptr(scratch),
                                                       L.ptr(slab_m), slab_m.numel(), L.ptr(slab_s), 0 if slab_s is None else slab_s.numel(), side_ptr, ev_ptr,
                                                       C.byref(dxp), C.byref(tptr), C.byref(trows), st), "sat_encoder_blocks_bwd")
                    dptr = dxp.value
                    for i in range(first, last + 1):
                        grads.update(per_block[i])
                    if cb is not None and first > 0:          # a ResNet stage just finished (its weight gradients are joined)
                        cb(dict(grads))
                d = _arena_view(arena, dptr, tuple(xin.shape))
                t["loop_keep"] = (dests, dcur)
            for idx in range(len(recs) - 1, -1, -1):
                d, tiles = _block_bwd(recs[idx], d, grads, True, Wt, dout_tiles=tiles, prev=(recs[idx - 1] if idx > 0 else None), queue=queue)
                if cb is not None and idx in (bounds[2], bounds[1], bounds[0]):      # a ResNet stage just finished
                    queue.join()
                    cb(dict(grads))
            if "a0" not in t:
                dc0, grads[enc[2].weight], grads[enc[2].bias] = stem_tail_bwd(d, t["c0"], t["s0"], enc[2])
            else:
                Nn, Hh, Ww, Cc = t["a0"].shape
                da0 = torch.empty_like(t["a0"])
                L.check(lib.sat_maxpool3x3s2_bwd_t(int(bf), L.ptr(d), L.ptr(t["amax"]), L.ptr(da0), Nn, Hh, Ww, Cc, st), "sat_maxpool3x3s2_bwd")
                dc0, grads[enc[2].weight], grads[enc[2].bias] = bn_bwd(da0, t["c0"], t["a0"], t["s0"], enc[2], True)
            if t.get("stem_pairs"):
                dwp = conv_wgrad(dc0, t["x0"], t["wp"], 2, 0, stride_w=1)               # (64,8,7,4) view of K,7,4,8 fp32 memory
            else:
                dwp = conv_wgrad(dc0, t["x0"], t["wp"], 2, 3)                          # (64,cpad,7,7) view of KRS{4,8} fp32 memory
            dw3 = L.grad_buffer(enc[1].weight)
            if t.get("stem_pairs"):
                L.check(lib.sat_stem_filter_grad_unpairs(L.ptr(dwp), L.ptr(dw3), enc[1].out_channels, st), "sat_stem_filter_grad_unpairs")
            elif bf:
                L.check(lib.sat_stem_filter_grad_unpad(L.ptr(dwp), L.ptr(dw3), enc[1].out_channels * 49, st), "sat_stem_filter_grad_unpad")
            else:
                L.check(lib.sat_pad_channels_3to4(L.ptr(dwp), L.ptr(dw3), enc[1].out_channels * 49, 1, st), "sat_pad_channels_3to4")
            grads[enc[1].weight] = dw3
            queue.join()
        ctx.t = ctx.recs = ctx.Wt = None
        return (None, None, *[grads.get(p) if p.requires_grad else None for p in ctx.params])


class HipEncoder(nn.Sequential):
    Fn = EncoderFn

    def __init__(self, norm, conv1, bn1, layers, proj, out_size):
        mods = [norm, conv1, bn1, nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1), *layers]
        if proj is not None:
            mods.append(proj)
        super().__init__(*mods)
        self.__dict__["proj"] = proj              # not registered twice: index 9 already owns it
        self.out_size = out_size
        self.precision = "fp32"                   # or "bf16" (see EncoderFn)

    @property
    def trunk_trainable(self):
        return any(p.requires_grad for p in self[1].parameters())

    def forward(self, img):
        params = self.__dict__.get("_plist")
        if params is None:          # the module tree is walked once, not every step (1 ms of host time per resnet50 forward)
            params = self.__dict__["_plist"] = list(self.parameters())
        return EncoderFn.apply(img, self, *params)

    def _apply(self, fn, *a, **k):                # .cuda() / .to() / .half() may replace the parameter objects
        self.__dict__.pop("_plist", None)
        return super()._apply(fn, *a, **k)


def _pretrained_file(arch, pretrained):
    """``pretrained`` (model.py:18): the reference downloads torchvision's ImageNet weights.  There is no network here: a path
    (``pretrained="/x/resnet50.pth"``) or, for ``True``, ``$SAT_PRETRAINED_DIR`` / ``$TORCH_HOME/hub/checkpoints`` /
    ``~/.cache/torch/hub/checkpoints`` searched for torchvision's file name (``<arch>-<hash>.pth``) or ``<arch>.pth``."""
    if not pretrained:
        return None
    if isinstance(pretrained, (str, os.PathLike)):
        if not os.path.isfile(pretrained):
            raise FileNotFoundError("pretrained=%r: no such file" % (pretrained,))
        return os.fspath(pretrained)
    dirs = [os.environ.get("SAT_PRETRAINED_DIR"), os.path.join(os.environ.get("TORCH_HOME", os.path.expanduser("~/.cache/torch")), "hub", "checkpoints")]
    for d in dirs:
        if d and os.path.isdir(d):
            hits = sorted(glob.glob(os.path.join(d, arch + "-*.pth")) + glob.glob(os.path.join(d, arch + ".pth")))
            if hits:
                return hits[0]
    raise RuntimeError("pretrained=True: no local torchvision checkpoint for %s (searched %s); there is no download here - pass pretrained=<path> "
                       "or set SAT_PRETRAINED_DIR" % (arch, [d for d in dirs if d]))


def _load_torchvision_trunk(path, conv1, bn1, layers):
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    sd = {k: v for k, v in sd.items() if not k.startswith("fc.")}          # model.py:29 drops pooling and fc
    holder = nn.Module()
    holder.conv1, holder.bn1 = conv1, bn1
    for i, layer in enumerate(layers):
        setattr(holder, "layer%d" % (i + 1), layer)
    missing, unexpected = holder.load_state_dict(sd, strict=False)
    missing = [k for k in missing if not k.endswith("num_batches_tracked")]          # absent from very old torchvision files
    if missing or unexpected:
        raise RuntimeError("pretrained checkpoint %s does not fit: missing %s, unexpected %s" % (path, missing[:5], list(unexpected)[:5]))


def _probe_zero_image(conv1, bn1, layers, size):
    """model.py:46-48 pushes one all-zero image through the train-mode trunk: its only lasting effect is on the BatchNorm buffers.
    Initialisation-time host arithmetic on a single image (torch CPU ops), not part of the step."""
    with torch.no_grad():
        def bn(x, m):
            return F.batch_norm(x, m.running_mean, m.running_var, m.weight, m.bias, True, m.momentum, m.eps)

        def cv(x, m):
            return F.conv2d(x, m.weight, None, m.stride, m.padding, 1, m.groups)
        x = torch.zeros(1, 3, size, size)
        x = F.max_pool2d(F.relu(bn(cv(x, conv1), bn1)), 3, 2, 1); bn1.num_batches_tracked += 1
        for layer in layers:
            for blk in layer:
                idn = x
                if blk.downsample is not None:
                    idn = bn(cv(x, blk.downsample[0]), blk.downsample[1]); blk.downsample[1].num_batches_tracked += 1
                y = F.relu(bn(cv(x, blk.conv1), blk.bn1)); blk.bn1.num_batches_tracked += 1
                if blk.kind == "basic":
                    y = bn(cv(y, blk.conv2), blk.bn2); blk.bn2.num_batches_tracked += 1
                else:
                    y = F.relu(bn(cv(y, blk.conv2), blk.bn2)); blk.bn2.num_batches_tracked += 1
                    y = bn(cv(y, blk.conv3), blk.bn3); blk.bn3.num_batches_tracked += 1
                x = F.relu(y + idn)


def get_encoder(args):
    """Reference get_encoder (model.py:16-63) for the resnet / wide_resnet / resnext archs shufflenet_v2 (``encoder_shuffle.py``) and mobilenet_v2 (``encoder_mobilenet.py``); adds the README's
    ``encoder_size`` resize (readme.md:118-121) when ``args.encoder_size`` is set."""
    arch = args.encoder_arch
    if arch.startswith("shufflenet_v2"):          # model.py:30-31 (the CLI default, train.py:43)
        from . import encoder_shuffle
        if arch in encoder_shuffle.SHUFFLENETS:
            return encoder_shuffle.get_shuffle_encoder(args)
    if arch == "mobilenet_v2":                    # model.py:38-39
        from . import encoder_mobilenet
        return encoder_mobilenet.get_mobilenet_encoder(args)
    if arch not in RESNETS:
        raise ValueError("Encoder not supported : {}".format(arch))
    ckpt = _pretrained_file(arch, getattr(args, "pretrained", False))
    kind, depths, wpg = RESNETS[arch][:3]
    groups = RESNETS[arch][3] if len(RESNETS[arch]) > 3 else 1
    conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False); bn1 = nn.BatchNorm2d(64)
    layers, cin = [], 64
    for si, (planes, nblk) in enumerate(zip((64, 128, 256, 512), depths)):
        blocks = []
        for bi in range(nblk):
            blk = Block(kind, cin, planes, 2 if (si > 0 and bi == 0) else 1, wpg, groups)
            cin = blk.cout; blocks.append(blk)
        layers.append(nn.Sequential(*blocks))
    final_dim = cin
    nn.Linear(final_dim, 1000)        # torchvision builds (and the reference then drops, model.py:29) an fc layer: keep the RNG stream aligned
    for mod in [conv1, *layers]:
        for sub in mod.modules():
            if isinstance(sub, nn.Conv2d):
                nn.init.kaiming_normal_(sub.weight, mode="fan_out", nonlinearity="relu")
    # model.py:46-48 pushes a zero image through the train-mode trunk to read its output shape.  With zero
    # biases every BatchNorm sees an all-zero batch: running_mean stays 0, running_var becomes 0.9, one batch tracked.
    if ckpt is None:
        for mod in [bn1, *layers]:
            for sub in mod.modules():
                if isinstance(sub, nn.BatchNorm2d):
                    sub.running_var.fill_(0.9); sub.num_batches_tracked.fill_(1)
    else:
        # model.py:18-24: torchvision weights, every trunk parameter frozen (the 1x1 projection added below stays trainable);
        # then the same zero-image probe, which with real weights moves every running statistic by one momentum step
        _load_torchvision_trunk(ckpt, conv1, bn1, layers)
        for mod in [conv1, bn1, *layers]:
            for prm in mod.parameters():
                prm.requires_grad = False
        _probe_zero_image(conv1, bn1, layers, int(args.input_size))
    inp = int(args.input_size)
    s = (inp + 2 * 3 - 7) // 2 + 1; s = (s + 2 - 3) // 2 + 1
    for _ in range(3):
        s = (s + 2 - 3) // 2 + 1
    final_size = s
    proj = None
    if getattr(args, "encoder_dim", None) is not None and args.encoder_dim != final_dim:
        proj = nn.Conv2d(final_dim, args.encoder_dim, kernel_size=1, stride=1, bias=True)      # model.py:53
    else:
        args.encoder_dim = final_dim
    es = getattr(args, "encoder_size", None)
    enc = HipEncoder(Normalize(args.mean, args.std, inplace=True), conv1, bn1, layers, proj, es if (es is not None and es != final_size) else None)
    _channels_last_(enc)
    _shadow_(enc)
    return enc
