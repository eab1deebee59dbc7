"""ShuffleNetV2 encoders: the reference's CLI default ``--encoder_arch shufflenet_v2_x0_5`` (train.py:43).  ``get_encoder`` keeps every child of
torchvision's model but the classifier (model.py:30-31): conv1 (3x3 stride 2 + BatchNorm + ReLU), maxpool, stage2-4, conv5 (1x1 + BatchNorm +
ReLU).  Same conventions as ``encoder.py``: the children only hold parameters under torchvision's state-dict keys (``1.0.weight`` conv1,
``3.0.branch1.0.weight`` ..., ``6.0.weight`` conv5, ``7.*`` the optional 1x1 projection), the layers run in ``libsat_hip.so`` on NHWC
activations (fp32, or bf16 storage with fp32 statistics / parameter gradients / master weights).

A unit of the network (Ma et al. 2018, fig. 3 c/d):

    stride 1:  x1, x2 = halves of the channels;  out = shuffle(cat(x1, branch2(x2)))
    stride 2:  out = shuffle(cat(branch1(x), branch2(x)))
    branch1 = depthwise 3x3 (stride 2) - BN - 1x1 - BN - ReLU;   branch2 = 1x1 - BN - ReLU - depthwise 3x3 - BN - 1x1 - BN - ReLU

The 1x1 convolutions are the library's implicit-GEMM kernels (with the BatchNorm statistics in their epilogue in bf16 mode), the depthwise
convolutions and the shuffle are streaming kernels (``csrc/depthwise.hip``).  The shuffle writes what its consumer reads: the two halves as
separate dense tensors in front of a stride-1 unit (so ``x.chunk(2, dim=1)`` costs nothing and the pass-through half is never copied on its
own), the whole tensor in front of a stride-2 unit / conv5.

Widths: the kernels move 16-byte vectors (8 bf16 channels).  x0_5 (branches of 24 / 48 / 96 channels) and x1_5 (88 / 176 / 352) fit; x1_0
(58 / 116 / 232) and x2_0 (122 / 244 / 488) do not: their parameters and activations are HELD zero-padded to the next multiple of 8 (58 -> 64,
116 -> 120, 122 -> 128, 244 -> 248).  The padding is exact, not an approximation: a padded output channel has zero filter rows and a zero
BatchNorm scale / shift, so it is exactly 0 forward, its gradients are exactly 0 backward (the BatchNorm backward multiplies by the zero scale;
a padded INPUT channel meets zero filter columns), and an optimizer step of a zero gradient on a zero weight with zero moments leaves zero -
also with weight decay.  ``state_dict()`` / ``load_state_dict()`` slice / pad at the module boundary (``_hold_padded_``), so checkpoints have
the reference's shapes; ``parameters()`` are the padded tensors.
"""
import ctypes as C

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib as L
from . import encoder as E

#: arch -> (units per stage, output channels of conv1, stage2, stage3, stage4, conv5); torchvision's table
SHUFFLENETS = {
    "shufflenet_v2_x0_5": ((4, 8, 4), (24, 48, 96, 192, 1024)),
    "shufflenet_v2_x1_0": ((4, 8, 4), (24, 116, 232, 464, 1024)),
    "shufflenet_v2_x1_5": ((4, 8, 4), (24, 176, 352, 704, 1024)),
    "shufflenet_v2_x2_0": ((4, 8, 4), (24, 244, 488, 976, 2048)),
}


def _pad8(c):
    return (c + 7) // 8 * 8


def _slice_to(t, shape):
    return t[tuple(slice(0, n) for n in shape)]


def _hold_padded_(mod, shapes):
    """Replace the parameters / buffers of leaf module ``mod`` named in ``shapes`` (name -> padded shape) by zero-padded tensors holding the old
    values in their leading corner, and make the module's state dict speak the ORIGINAL shapes (save: slice; load: pad)."""
    real = {}
    for name, shape in shapes.items():
        old = getattr(mod, name)
        if tuple(old.shape) == tuple(shape):
            continue
        real[name] = tuple(old.shape)
        new = torch.zeros(shape, dtype=old.dtype, device=old.device)
        _slice_to(new, old.shape).copy_(old.detach())
        if isinstance(old, nn.Parameter):
            setattr(mod, name, nn.Parameter(new, requires_grad=old.requires_grad))
        else:
            mod.register_buffer(name, new)
    if not real:
        return
    mod.__dict__["_sat_real_shapes"] = real

    def save_hook(m, sd, prefix, local_metadata):
        for name, shape in real.items():
            if prefix + name in sd:
                sd[prefix + name] = _slice_to(sd[prefix + name], shape).clone()

    def load_hook(sd, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for name, shape in real.items():
            t = sd.get(prefix + name)
            if t is not None and tuple(t.shape) == shape:
                full = torch.zeros(getattr(mod, name).shape, dtype=t.dtype, device=t.device)
                _slice_to(full, shape).copy_(t)
                sd[prefix + name] = full

    mod._register_state_dict_hook(save_hook)
    mod._register_load_state_dict_pre_hook(load_hook)


def _pad_conv_(conv, cin_p, cout_p):
    """(K, C / groups, R, S) filter -> (cout_p, cin_p or 1, R, S), zeros outside the original block"""
    dw = conv.groups > 1
    _hold_padded_(conv, {"weight": (cout_p, 1 if dw else cin_p) + tuple(conv.weight.shape[2:])})
    conv.in_channels, conv.out_channels = cin_p, cout_p
    if dw:
        conv.groups = cout_p


def _pad_bn_(bn, c_p):
    if bn.num_features == c_p:
        return
    # the padded channels get scale 0 and shift 0 (``_hold_padded_`` fills zeros): their output and every gradient are exactly 0
    _hold_padded_(bn, {"weight": (c_p,), "bias": (c_p,), "running_mean": (c_p,), "running_var": (c_p,)})
    bn.num_features = c_p


class ShuffleUnit(nn.Module):
    """Parameter holder of one torchvision InvertedResidual (same child indices inside branch1 / branch2)."""

    def __init__(self, inp, oup, stride):
        super().__init__()
        self.stride = stride
        bfeat = oup // 2
        self.bfeat = bfeat          # the real branch width (``pad_()`` may hold the tensors wider)
        if stride == 1 and inp != 2 * bfeat:
            raise ValueError("a stride-1 unit keeps its width")

        def dw(c, s):
            return nn.Conv2d(c, c, 3, s, 1, bias=False, groups=c)

        if stride > 1:
            self.branch1 = nn.Sequential(dw(inp, stride), nn.BatchNorm2d(inp), nn.Conv2d(inp, bfeat, 1, 1, 0, bias=False), nn.BatchNorm2d(bfeat), nn.ReLU(inplace=True))
        else:
            self.branch1 = nn.Sequential()
        self.branch2 = nn.Sequential(nn.Conv2d(inp if stride > 1 else bfeat, bfeat, 1, 1, 0, bias=False), nn.BatchNorm2d(bfeat), nn.ReLU(inplace=True),
                                     dw(bfeat, stride), nn.BatchNorm2d(bfeat), nn.Conv2d(bfeat, bfeat, 1, 1, 0, bias=False), nn.BatchNorm2d(bfeat),
                                     nn.ReLU(inplace=True))

    def pad_(self, inp_p):
        """hold every tensor of the unit at multiples of 8 channels: ``inp_p`` = the width of the unit's input in memory (stride 2; the halves of
        a stride-1 unit are branch-wide).  Returns the width of the unit's whole output in memory."""
        bp = _pad8(self.bfeat)
        if self.stride > 1:
            b1 = list(self.branch1)
            _pad_conv_(b1[0], inp_p, inp_p); _pad_bn_(b1[1], inp_p); _pad_conv_(b1[2], inp_p, bp); _pad_bn_(b1[3], bp)
        b2 = list(self.branch2)
        _pad_conv_(b2[0], inp_p if self.stride > 1 else bp, bp); _pad_bn_(b2[1], bp); _pad_conv_(b2[3], bp, bp); _pad_bn_(b2[4], bp)
        _pad_conv_(b2[5], bp, bp); _pad_bn_(b2[6], bp)
        return _pad8(2 * self.bfeat)

    def branches(self):
        """(branch1, branch2) as plain lists: ``nn.Sequential.__getitem__`` walks an OrderedDict per lookup (~440 lookups per x0_5 step)"""
        b = self.__dict__.get("_branch_lists")
        if b is None:
            b = self.__dict__["_branch_lists"] = (list(self.branch1), list(self.branch2))
        return b


# ----------------------------------------------------------------------------- raw layer calls
def _dw_weight(conv):
    w = conv.weight
    assert w.dtype == torch.float32 and w.shape[1] == 1 and tuple(w.shape[2:]) == (3, 3)
    return w if (w.is_contiguous() or w.is_contiguous(memory_format=torch.channels_last)) else w.contiguous()      # (C, 1, 3, 3): [C][9] either way


def dw_fwd(x, conv):
    N, H, W, Cc = x.shape
    s = conv.stride[0]
    y = torch.empty(N, (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1, Cc, dtype=x.dtype, device=x.device)
    L.check(L.lib().sat_dwconv3x3_fwd_t(int(E._is_bf(x)), L.ptr(x), L.ptr(_dw_weight(conv)), L.ptr(y), N, H, W, Cc, s, L.stream_ptr()), "sat_dwconv3x3_fwd")
    return y


def dw_dgrad(dy, conv, x_shape):
    N, H, W, Cc = x_shape
    dx = torch.empty(N, H, W, Cc, dtype=dy.dtype, device=dy.device)
    L.check(L.lib().sat_dwconv3x3_dgrad_t(int(E._is_bf(dy)), L.ptr(dy), L.ptr(_dw_weight(conv)), L.ptr(dx), N, H, W, Cc, conv.stride[0], L.stream_ptr()),
            "sat_dwconv3x3_dgrad")
    return dx


_dw_scratch = {}


def dw_wgrad(dy, x, conv):
    """fp32 gradient of the (C, 1, 3, 3) filter, written to the parameter's gradient buffer"""
    lib = L.lib()
    N, H, W, Cc = x.shape
    s = conv.stride[0]
    need = lib.sat_dwconv3x3_wgrad_scratch_bytes(N, H, W, Cc, s) // 4
    key = (x.device.index, L.stream_ptr().value)
    buf = _dw_scratch.get(key)
    if buf is None or buf.numel() < need:
        buf = _dw_scratch[key] = torch.empty(max(need, 1 << 18), dtype=torch.float32, device=x.device)
    out = L.grad_buffer(conv.weight)
    dense = out.is_contiguous() or out.is_contiguous(memory_format=torch.channels_last)
    dst = out if dense else torch.empty(Cc, 1, 3, 3, dtype=torch.float32, device=x.device)
    L.check(lib.sat_dwconv3x3_wgrad_t(int(E._is_bf(x)), L.ptr(dy), L.ptr(x), L.ptr(dst), N, H, W, Cc, s, L.ptr(buf), L.stream_ptr()), "sat_dwconv3x3_wgrad")
    return dst


def shuffle_join(a, b, halves, ch=None):
    """channel_shuffle(cat(a, b), 2) of two NHWC branches: the full tensor, or (x1, x2) = its channel halves as separate tensors.  ``ch``: the real
    branch width when the tensors are held wider (zero channels past it, see the module docstring)"""
    N, H, W, Chp = a.shape
    ch = Chp if ch is None else ch
    assert tuple(b.shape) == tuple(a.shape) and a.dtype == b.dtype and a.is_contiguous() and b.is_contiguous()
    rows = N * H * W
    if halves:
        x1, x2 = torch.empty_like(a), torch.empty_like(a)
        L.check(L.lib().sat_shuffle_join_t(int(E._is_bf(a)), L.ptr(a), L.ptr(b), None, L.ptr(x1), L.ptr(x2), rows, ch, Chp, L.stream_ptr()), "sat_shuffle_join")
        return x1, x2
    full = torch.empty(N, H, W, _pad8(2 * ch), dtype=a.dtype, device=a.device)
    L.check(L.lib().sat_shuffle_join_t(int(E._is_bf(a)), L.ptr(a), L.ptr(b), L.ptr(full), None, None, rows, ch, Chp, L.stream_ptr()), "sat_shuffle_join")
    return full


def shuffle_split(d, ch=None, chp=None):
    """backward of ``shuffle_join``: d = the full gradient or the pair of its halves -> (da, db); ``ch`` / ``chp``: real / held branch width"""
    if isinstance(d, tuple):
        d1, d2 = d
        N, H, W, Chp = d1.shape
        ch = Chp if ch is None else ch
        da, db = torch.empty_like(d1), torch.empty_like(d1)
        L.check(L.lib().sat_shuffle_split_t(int(E._is_bf(d1)), None, L.ptr(d1), L.ptr(d2), L.ptr(da), L.ptr(db), N * H * W, ch, Chp, L.stream_ptr()), "sat_shuffle_split")
        return da, db
    N, H, W, Fp = d.shape
    ch = Fp // 2 if ch is None else ch
    chp = ch if chp is None else chp
    assert Fp == _pad8(2 * ch) or (Fp == 2 * ch and ch == chp)
    d = d.contiguous()
    da = torch.empty(N, H, W, chp, dtype=d.dtype, device=d.device); db = torch.empty_like(da)
    L.check(L.lib().sat_shuffle_split_t(int(E._is_bf(d)), L.ptr(d), None, None, L.ptr(da), L.ptr(db), N * H * W, ch, chp, L.stream_ptr()), "sat_shuffle_split")
    return da, db


# ----------------------------------------------------------------------------- units
class _URec:
    __slots__ = ("u", "src", "d1", "sd1", "e1", "c1", "a", "sa", "c2", "a2", "s2", "d2", "sd2", "e2", "c3", "b", "s3")


def _unit_fwd(u, xin, training, Wt, halves):
    """xin: the whole NHWC tensor (stride 2) or the pair of its channel halves (stride 1).  Returns (record, output as ``halves`` asks)."""
    conv = E.conv_fwd_stats if training else (lambda *a: (E.conv_fwd(*a), None))
    r = _URec(); r.u = u
    b1, b2 = u.branches()
    if u.stride == 1:
        a, r.src = xin
    else:
        r.src = xin
        r.d1 = dw_fwd(xin, b1[0]); r.e1, r.sd1 = E.bn_fwd(r.d1, b1[1], None, False, training)
        r.c1, tl = conv(r.e1, Wt(b1[2].weight), 1, 0); r.a, r.sa = E.bn_fwd(r.c1, b1[3], None, True, training, want_mask=True, tiles=tl)
        a = r.a
    r.c2, tl = conv(r.src, Wt(b2[0].weight), 1, 0); r.a2, r.s2 = E.bn_fwd(r.c2, b2[1], None, True, training, want_mask=True, tiles=tl)
    r.d2 = dw_fwd(r.a2, b2[3]); r.e2, r.sd2 = E.bn_fwd(r.d2, b2[4], None, False, training)
    r.c3, tl = conv(r.e2, Wt(b2[5].weight), 1, 0); r.b, r.s3 = E.bn_fwd(r.c3, b2[6], None, True, training, want_mask=True, tiles=tl)
    return r, shuffle_join(a, r.b, halves, u.bfeat)


def _bn_g(grads, bn, res):
    dx, grads[bn.weight], grads[bn.bias] = res
    return dx


def _unit_bwd(r, dout, grads, Wt, need_dx=True):
    """dout: gradient of the unit's output (whole, or its halves).  Returns the gradient of the unit's input in the form the input had."""
    u = r.u
    b1, b2 = u.branches()
    da, db = shuffle_split(dout, u.bfeat, r.b.shape[-1])
    dc3 = _bn_g(grads, b2[6], E.bn_bwd(db, r.c3, r.b, r.s3, b2[6], True))
    grads[b2[5].weight] = E.conv_wgrad(dc3, r.e2, b2[5].weight, 1, 0, param=b2[5].weight)
    de2, tl = E.conv_dgrad(dc3, Wt(b2[5].weight), r.e2.shape, 1, 0, bn=(r.d2, r.sd2))
    dd2 = _bn_g(grads, b2[4], E.bn_bwd(de2, r.d2, None, r.sd2, b2[4], False, tiles=tl))
    grads[b2[3].weight] = dw_wgrad(dd2, r.a2, b2[3])
    da2 = dw_dgrad(dd2, b2[3], r.a2.shape)
    dc2 = _bn_g(grads, b2[1], E.bn_bwd(da2, r.c2, r.a2, r.s2, b2[1], True))
    grads[b2[0].weight] = E.conv_wgrad(dc2, r.src, b2[0].weight, 1, 0, param=b2[0].weight)
    if u.stride == 1:
        return da, E.conv_dgrad(dc2, Wt(b2[0].weight), r.src.shape, 1, 0)
    dc1 = _bn_g(grads, b1[3], E.bn_bwd(da, r.c1, r.a, r.sa, b1[3], True))
    grads[b1[2].weight] = E.conv_wgrad(dc1, r.e1, b1[2].weight, 1, 0, param=b1[2].weight)
    de1, tl = E.conv_dgrad(dc1, Wt(b1[2].weight), r.e1.shape, 1, 0, bn=(r.d1, r.sd1))
    dd1 = _bn_g(grads, b1[1], E.bn_bwd(de1, r.d1, None, r.sd1, b1[1], False, tiles=tl))
    grads[b1[0].weight] = dw_wgrad(dd1, r.src, b1[0])
    if not need_dx:
        return None
    dx = dw_dgrad(dd1, b1[0], r.src.shape)
    return E.conv_dgrad(dc2, Wt(b2[0].weight), r.src.shape, 1, 0, out=dx, accumulate=True)


def stem3x3_fwd(norm, conv1, img, bf, training):
    """Normalize + NCHW -> NHWC with the 3 image channels padded to 4 (fp32) / 8 (bf16), then the 3x3 stride-2 stem convolution of shufflenet_v2 /
    mobilenet_v2.  Returns (padded image, padded filter, convolution output, its BatchNorm statistics tiles or None)."""
    lib = L.lib()
    st = L.stream_ptr()
    N, _, H, W = img.shape
    adt = E.BF16 if bf else torch.float32
    mean = (C.c_float * 3)(*norm.mean); std = (C.c_float * 3)(*norm.std)
    K = conv1.out_channels
    w3 = E._krsc(conv1.weight)                                                  # (K,3,3,3), memory K,3,3,3(c)
    cpad = 8 if bf else 4
    x0 = torch.empty(N, H, W, cpad, dtype=adt, device=img.device)
    wp = torch.empty(K, cpad, 3, 3, dtype=adt, device=img.device).contiguous(memory_format=torch.channels_last)
    if bf:
        L.check(lib.sat_image_normalize_nhwc8_bf16(L.ptr(img), L.ptr(x0), N, H, W, mean, std, st), "sat_image_normalize_nhwc8_bf16")
        L.check(lib.sat_stem_filter_pad(L.ptr(w3), L.ptr(wp), K * 9, st), "sat_stem_filter_pad")
    else:
        L.check(lib.sat_image_normalize_nhwc4(L.ptr(img), L.ptr(x0), N, H, W, mean, std, st), "sat_image_normalize_nhwc4")
        L.check(lib.sat_pad_channels_3to4(L.ptr(w3), L.ptr(wp), K * 9, 0, st), "sat_pad_channels_3to4")
    c0, tl = E.conv_fwd_stats(x0, wp, 2, 1) if training else (E.conv_fwd(x0, wp, 2, 1), None)
    return x0, wp, c0, tl


def stem3x3_wgrad(dc0, x0, wp, conv1, bf):
    """gradient of the (K,3,3,3) stem filter from the gradient of the convolution output (through the padded filter's gradient)"""
    lib = L.lib()
    st = L.stream_ptr()
    dwp = E.conv_wgrad(dc0, x0, wp, 2, 1)                          # (K,cpad,3,3) view of K,3,3,{4,8} fp32 memory
    dw3 = L.grad_buffer(conv1.weight)
    dst = dw3 if dw3.permute(0, 2, 3, 1).is_contiguous() else torch.empty(conv1.weight.shape, dtype=torch.float32, device=dc0.device).contiguous(memory_format=torch.channels_last)
    if bf:
        L.check(lib.sat_stem_filter_grad_unpad(L.ptr(dwp), L.ptr(dst), conv1.out_channels * 9, st), "sat_stem_filter_grad_unpad")
    else:
        L.check(lib.sat_pad_channels_3to4(L.ptr(dwp), L.ptr(dst), conv1.out_channels * 9, 1, st), "sat_pad_channels_3to4")
    return dst


class ShuffleEncoderFn(torch.autograd.Function):
    """img (B,3,H,W) fp32 in [0,1] -> annotations (B,D,h,w) fp32 (NHWC memory); ``enc.precision`` as in ``encoder.EncoderFn``."""

    @staticmethod
    def forward(ctx, img, enc, *params):
        try:
            return ShuffleEncoderFn._forward(ctx, img, enc, *params)
        finally:
            E._defer[0] = False

    @staticmethod
    def _forward(ctx, img, enc, *params):
        lib = L.lib()
        L.require_gpu(img, *params)
        if img.dim() != 4 or img.shape[1] != 3 or img.dtype != torch.float32:
            raise ValueError("encoder input must be (B,3,H,W) fp32 in [0,1]")
        img = img.contiguous()
        training = enc.training
        E._defer[0] = True; del E._tracked[:]
        bf = enc.precision == "bf16"
        adt = E.BF16 if bf else torch.float32
        st = L.stream_ptr()
        t = {}
        Wt = E._weight_reader(bf)
        conv1, bn1, conv5, bn5, units = enc.layers()
        t["x0"], t["wp"], t["c0"], tl = stem3x3_fwd(enc[0], conv1, img, bf, training)
        if training:          # bn + relu + maxpool in one pass
            x, t["s0"] = E.stem_tail_fwd(t["c0"], bn1, tiles=tl)
        else:
            a0, _ = E.bn_fwd(t["c0"], bn1, None, True, False)
            Nn, Hh, Ww, Cc = a0.shape
            x = torch.empty(Nn, (Hh + 2 - 3) // 2 + 1, (Ww + 2 - 3) // 2 + 1, Cc, dtype=adt, device=img.device)
            amax = torch.empty(x.shape, dtype=torch.uint8, device=img.device)
            L.check(lib.sat_maxpool3x3s2_fwd_t(int(bf), L.ptr(a0), L.ptr(x), L.ptr(amax), Nn, Hh, Ww, Cc, st), "sat_maxpool3x3s2_fwd")
        recs = []
        for i, u in enumerate(units):
            halves = i + 1 < len(units) and units[i + 1].stride == 1
            r, x = _unit_fwd(u, x, training, Wt, halves)
            recs.append(r)
        t["x5"] = x
        t["c5"], tl = E.conv_fwd_stats(x, Wt(conv5.weight), 1, 0) if training else (E.conv_fwd(x, Wt(conv5.weight), 1, 0), None)
        t["a5"], t["s5"] = E.bn_fwd(t["c5"], bn5, None, True, training, want_mask=True, tiles=tl)
        x = E._head_fwd(enc, t["a5"], t, Wt, bf)
        E._defer[0] = False
        if E._tracked:
            torch._foreach_add_(E._tracked, 1)
            del E._tracked[:]
        ctx.t, ctx.recs, ctx.enc, ctx.Wt, ctx.bf = t, recs, enc, Wt, bf
        ctx.params = params
        return x.permute(0, 3, 1, 2)            # (B, D, h, w) view over NHWC memory

    @staticmethod
    def backward(ctx, dann):
        enc, t, recs, Wt, bf = ctx.enc, ctx.t, ctx.recs, ctx.Wt, ctx.bf
        grads = {}
        d = E._head_bwd(enc, t, dann, grads, Wt, bf)
        if enc.trunk_trainable:
            conv1, bn1, conv5, bn5, _ = enc.layers()
            dc5 = _bn_g(grads, bn5, E.bn_bwd(d, t["c5"], t["a5"], t["s5"], bn5, True))
            grads[conv5.weight] = E.conv_wgrad(dc5, t["x5"], conv5.weight, 1, 0, param=conv5.weight)
            d = E.conv_dgrad(dc5, Wt(conv5.weight), t["x5"].shape, 1, 0)
            for r in reversed(recs):
                d = _unit_bwd(r, d, grads, Wt)
            dc0 = _bn_g(grads, bn1, E.stem_tail_bwd(d, t["c0"], t["s0"], bn1))
            grads[conv1.weight] = stem3x3_wgrad(dc0, t["x0"], t["wp"], conv1, bf)
        ctx.t = ctx.recs = ctx.Wt = None
        return (None, None, *[grads.get(p) if p.requires_grad else None for p in ctx.params])


class HipShuffleEncoder(nn.Sequential):
    Fn = ShuffleEncoderFn
    single_bucket = True          # data-parallel exchange: one bucket for the whole trunk (0.3 - 2.5 M parameters)

    def __init__(self, norm, conv1, stages, conv5, proj, out_size):
        mods = [norm, conv1, nn.MaxPool2d(3, 2, 1), *stages, conv5]
        if proj is not None:
            mods.append(proj)
        super().__init__(*mods)
        self.__dict__["proj"] = proj              # not registered twice: index 7 already owns it
        self.out_size = out_size
        self.precision = "fp32"

    @property
    def trunk_trainable(self):
        return any(p.requires_grad for p in self[1].parameters())

    def layers(self):
        """(conv1, bn1, conv5, bn5, [units of stage2..4]) looked up once"""
        ls = self.__dict__.get("_layers")
        if ls is None:
            ls = self.__dict__["_layers"] = (self[1][0], self[1][1], self[6][0], self[6][1], [u for li in (3, 4, 5) for u in self[li]])
        return ls

    def forward(self, img):
        params = self.__dict__.get("_plist")
        if params is None:
            params = self.__dict__["_plist"] = list(self.parameters())
        return ShuffleEncoderFn.apply(img, self, *params)

    def _apply(self, fn, *a, **k):
        self.__dict__.pop("_plist", None)
        return super()._apply(fn, *a, **k)


def _load_torchvision_trunk(path, conv1, stages, conv5):
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    sd = {k: v for k, v in sd.items() if not k.startswith("fc.")}          # model.py:31 drops the classifier
    holder = nn.Module()
    holder.conv1, holder.conv5 = conv1, conv5
    for i, stage in enumerate(stages):
        setattr(holder, "stage%d" % (i + 2), stage)
    missing, unexpected = holder.load_state_dict(sd, strict=False)
    missing = [k for k in missing if not k.endswith("num_batches_tracked")]
    if missing or unexpected:
        raise RuntimeError("pretrained checkpoint %s does not fit: missing %s, unexpected %s" % (path, missing[:5], list(unexpected)[:5]))


def _probe_zero_image(conv1, stages, conv5, size):
    """model.py:46-48 pushes one all-zero image through the train-mode trunk: its only lasting effect is on the BatchNorm buffers.
    Initialisation-time host arithmetic on a single image (torch CPU ops), not part of the step."""
    with torch.no_grad():
        def seq(x, mods):
            for m in mods:
                if isinstance(m, nn.BatchNorm2d):
                    x = F.batch_norm(x, m.running_mean, m.running_var, m.weight, m.bias, True, m.momentum, m.eps); m.num_batches_tracked += 1
                elif isinstance(m, nn.Conv2d):
                    x = F.conv2d(x, m.weight, None, m.stride, m.padding, 1, m.groups)
                else:
                    x = F.relu(x)
            return x
        x = F.max_pool2d(seq(torch.zeros(1, 3, size, size), conv1), 3, 2, 1)
        for stage in stages:
            for u in stage:
                if u.stride == 1:
                    x1, x2 = x.chunk(2, dim=1)
                    out = torch.cat((x1, seq(x2, u.branch2)), 1)
                else:
                    out = torch.cat((seq(x, u.branch1), seq(x, u.branch2)), 1)
                B, Cc, Hh, Ww = out.shape
                x = out.view(B, 2, Cc // 2, Hh, Ww).transpose(1, 2).reshape(B, Cc, Hh, Ww)
        seq(x, conv5)


def get_shuffle_encoder(args):
    """Reference get_encoder (model.py:16-63) for the shufflenet_v2 archs (called by ``encoder.get_encoder``)."""
    arch = args.encoder_arch
    repeats, chans = SHUFFLENETS[arch]
    ckpt = E._pretrained_file(arch, getattr(args, "pretrained", False))
    # construction order = torchvision's, default initialisation: the RNG stream stays aligned with the reference's
    conv1 = nn.Sequential(nn.Conv2d(3, chans[0], 3, 2, 1, bias=False), nn.BatchNorm2d(chans[0]), nn.ReLU(inplace=True))
    stages, cin = [], chans[0]
    for rep, cout in zip(repeats, chans[1:4]):
        stages.append(nn.Sequential(ShuffleUnit(cin, cout, 2), *[ShuffleUnit(cout, cout, 1) for _ in range(rep - 1)]))
        cin = cout
    conv5 = nn.Sequential(nn.Conv2d(cin, chans[4], 1, 1, 0, bias=False), nn.BatchNorm2d(chans[4]), nn.ReLU(inplace=True))
    final_dim = chans[4]
    nn.Linear(final_dim, 1000)        # the classifier the reference drops (model.py:31): built for the RNG stream only
    if ckpt is None:
        # model.py:46-48: the zero image of the shape probe.  Zero biases: every BatchNorm sees an all-zero batch
        for mod in [conv1, *stages, conv5]:
            for sub in mod.modules():
                if isinstance(sub, nn.BatchNorm2d):
                    sub.running_var.fill_(0.9); sub.num_batches_tracked.fill_(1)
    else:
        _load_torchvision_trunk(ckpt, conv1, stages, conv5)
        for mod in [conv1, *stages, conv5]:
            for prm in mod.parameters():
                prm.requires_grad = False
        _probe_zero_image(conv1, stages, conv5, int(args.input_size))
    if any((c // 2) % 8 for c in chans[1:4]):          # x1_0 / x2_0: hold everything at multiples of 8 channels (after initialisation / loading / the probe,
        width = chans[0]                               # which ran on the original shapes)
        for stage in stages:
            for u in stage:
                width = u.pad_(width)
        assert width == chans[3], "the last stage's width is a multiple of 8 in every torchvision variant"
    s = int(args.input_size)
    for _ in range(5):                # conv1, maxpool, three stride-2 units: 3x3 windows, stride 2, pad 1
        s = (s + 2 - 3) // 2 + 1
    proj = None
    if getattr(args, "encoder_dim", None) is not None and args.encoder_dim != final_dim:
        proj = nn.Conv2d(final_dim, args.encoder_dim, kernel_size=1, stride=1, bias=True)      # model.py:53
    else:
        args.encoder_dim = final_dim
    es = getattr(args, "encoder_size", None)
    enc = HipShuffleEncoder(E.Normalize(args.mean, args.std, inplace=True), conv1, stages, conv5, proj, es if (es is not None and es != s) else None)
    E._channels_last_(enc)
    E._shadow_(enc)
    return enc
