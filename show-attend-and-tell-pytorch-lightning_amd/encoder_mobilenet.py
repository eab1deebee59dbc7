"""MobileNetV2 encoder (readme.md:104, model.py:38-39: ``get_encoder`` keeps ``m.features`` of torchvision's model, i.e. everything but the
classifier).  Same conventions as ``encoder.py`` / ``encoder_shuffle.py``: the children hold parameters under torchvision's state-dict keys
(``1.0.0.weight`` the stem convolution, ``1.<i>.conv.*`` the inverted residuals, ``1.18.0.weight`` the last 1x1, ``2.*`` the optional projection),
the layers run in ``libsat_hip.so`` on NHWC activations (fp32, or bf16 storage with fp32 statistics / parameter gradients / master weights).

An inverted residual (Sandler et al. 2018): ``[1x1 expand - BN - ReLU6] - depthwise 3x3 - BN - ReLU6 - 1x1 project - BN (+ x when the block keeps
shape)``.  1x1 convolutions: the implicit-GEMM kernels (BatchNorm statistics in their epilogue in bf16 mode); depthwise 3x3: ``csrc/depthwise.hip``;
ReLU6 = ``relu = 2`` of the BatchNorm apply kernels (clamp to [0, 6]; the mask bit the backward reads says "0 < v < 6", hardtanh's rule); the
residual add happens inside the last BatchNorm's apply kernel and, backward, inside the data-gradient launch of the expand convolution
(accumulating into the block's output gradient in place).  Every channel count of the width-1.0 network is a multiple of 8.
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import _lib as L
from . import encoder as E
from .encoder_shuffle import dw_dgrad, dw_fwd, dw_wgrad, stem3x3_fwd, stem3x3_wgrad

#: torchvision's inverted_residual_setting: (expand ratio t, output channels c, repeats n, stride s)
SETTING = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))
RELU6 = 2


def conv_bn_relu6(cin, cout, k=3, stride=1, groups=1):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False), nn.BatchNorm2d(cout), nn.ReLU6(inplace=True))


class InvertedResidual(nn.Module):
    """Parameter holder with torchvision's layout: ``conv`` = [ConvBNReLU expand (t != 1)], ConvBNReLU depthwise, Conv2d project, BatchNorm2d."""

    def __init__(self, inp, oup, stride, expand_ratio):
        super().__init__()
        hidden = int(round(inp * expand_ratio))
        self.stride = stride
        self.use_res_connect = stride == 1 and inp == oup
        layers = []
        if expand_ratio != 1:
            layers.append(conv_bn_relu6(inp, hidden, k=1))
        layers.extend([conv_bn_relu6(hidden, hidden, stride=stride, groups=hidden), nn.Conv2d(hidden, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup)])
        self.conv = nn.Sequential(*layers)

    def parts(self):
        """((expand conv, bn) or None, (depthwise conv, bn), (project conv, bn)) as plain tuples, looked up once"""
        p = self.__dict__.get("_parts")
        if p is None:
            mods = list(self.conv)
            ex = (mods[0][0], mods[0][1]) if len(mods) == 4 else None
            dwm = mods[-3]
            p = self.__dict__["_parts"] = (ex, (dwm[0], dwm[1]), (mods[-2], mods[-1]))
        return p


class _BRec:
    __slots__ = ("blk", "x", "ce", "ae", "se", "d", "ad", "sd", "cp", "sp")


def _block_fwd(blk, x, training, Wt):
    conv = E.conv_fwd_stats if training else (lambda *a: (E.conv_fwd(*a), None))
    ex, (dwc, dbn), (pc, pbn) = blk.parts()
    r = _BRec(); r.blk, r.x = blk, x
    h = x
    if ex is not None:
        r.ce, tl = conv(x, Wt(ex[0].weight), 1, 0)
        r.ae, r.se = E.bn_fwd(r.ce, ex[1], None, RELU6, training, want_mask=True, tiles=tl)
        h = r.ae
    r.d = dw_fwd(h, dwc)
    r.ad, r.sd = E.bn_fwd(r.d, dbn, None, RELU6, training, want_mask=True)
    r.cp, tl = conv(r.ad, Wt(pc.weight), 1, 0)
    out, r.sp = E.bn_fwd(r.cp, pbn, x if blk.use_res_connect else None, False, training, tiles=tl)
    return r, out


def _bn_g(grads, bn, res):
    dx, grads[bn.weight], grads[bn.bias] = res
    return dx


def _block_bwd(r, dout, grads, Wt, need_dx=True):
    """dout: gradient of the block's output (owned by the caller chain: it is overwritten when the block has the identity path)."""
    blk = r.blk
    ex, (dwc, dbn), (pc, pbn) = blk.parts()
    dcp = _bn_g(grads, pbn, E.bn_bwd(dout, r.cp, None, r.sp, pbn, False))
    grads[pc.weight] = E.conv_wgrad(dcp, r.ad, pc.weight, 1, 0, param=pc.weight)
    dad, tl = E.conv_dgrad(dcp, Wt(pc.weight), r.ad.shape, 1, 0, bn=(r.d, r.sd))
    dd = _bn_g(grads, dbn, E.bn_bwd(dad, r.d, r.ad, r.sd, dbn, RELU6, tiles=tl))
    h = r.ae if ex is not None else r.x
    grads[dwc.weight] = dw_wgrad(dd, h, dwc)
    if ex is None:
        return dw_dgrad(dd, dwc, h.shape) if need_dx else None          # t == 1 (the first block): 32 -> 16 channels, no identity path
    dae = dw_dgrad(dd, dwc, h.shape)
    dce = _bn_g(grads, ex[1], E.bn_bwd(dae, r.ce, r.ae, r.se, ex[1], RELU6))
    grads[ex[0].weight] = E.conv_wgrad(dce, r.x, ex[0].weight, 1, 0, param=ex[0].weight)
    if not need_dx:
        return None
    if blk.use_res_connect:          # dx = data gradient + dout: accumulated onto dout in place
        return E.conv_dgrad(dce, Wt(ex[0].weight), r.x.shape, 1, 0, out=dout, accumulate=True)
    return E.conv_dgrad(dce, Wt(ex[0].weight), r.x.shape, 1, 0)


class MobileNetEncoderFn(torch.autograd.Function):
    """img (B,3,H,W) fp32 in [0,1] -> annotations (B,D,h,w) fp32 (NHWC memory); ``enc.precision`` as in ``encoder.EncoderFn``."""

    @staticmethod
    def forward(ctx, img, enc, *params):
        try:
            return MobileNetEncoderFn._forward(ctx, img, enc, *params)
        finally:
            E._defer[0] = False

    @staticmethod
    def _forward(ctx, img, enc, *params):
        L.require_gpu(img, *params)
        if img.dim() != 4 or img.shape[1] != 3 or img.dtype != torch.float32:
            raise ValueError("encoder input must be (B,3,H,W) fp32 in [0,1]")
        img = img.contiguous()
        training = enc.training
        E._defer[0] = True; del E._tracked[:]
        bf = enc.precision == "bf16"
        t = {}
        Wt = E._weight_reader(bf)
        conv = E.conv_fwd_stats if training else (lambda *a: (E.conv_fwd(*a), None))
        (conv1, bn1), blocks, (convL, bnL) = enc.layers()
        t["x0"], t["wp"], t["c0"], tl = stem3x3_fwd(enc[0], conv1, img, bf, training)
        t["a0"], t["s0"] = E.bn_fwd(t["c0"], bn1, None, RELU6, training, want_mask=True, tiles=tl)
        x = t["a0"]
        recs = []
        for blk in blocks:
            r, x = _block_fwd(blk, x, training, Wt)
            recs.append(r)
        t["xL"] = x
        t["cL"], tl = conv(x, Wt(convL.weight), 1, 0)
        t["aL"], t["sL"] = E.bn_fwd(t["cL"], bnL, None, RELU6, training, want_mask=True, tiles=tl)
        x = E._head_fwd(enc, t["aL"], t, Wt, bf)
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
            (conv1, bn1), _, (convL, bnL) = enc.layers()
            dcL = _bn_g(grads, bnL, E.bn_bwd(d, t["cL"], t["aL"], t["sL"], bnL, RELU6))
            grads[convL.weight] = E.conv_wgrad(dcL, t["xL"], convL.weight, 1, 0, param=convL.weight)
            d = E.conv_dgrad(dcL, Wt(convL.weight), t["xL"].shape, 1, 0)
            for r in reversed(recs):
                d = _block_bwd(r, d, grads, Wt)
            dc0 = _bn_g(grads, bn1, E.bn_bwd(d, t["c0"], t["a0"], t["s0"], bn1, RELU6))
            grads[conv1.weight] = stem3x3_wgrad(dc0, t["x0"], t["wp"], conv1, bf)
        ctx.t = ctx.recs = ctx.Wt = None
        return (None, None, *[grads.get(p) if p.requires_grad else None for p in ctx.params])


class HipMobileNetEncoder(nn.Sequential):
    Fn = MobileNetEncoderFn
    single_bucket = True          # data-parallel exchange: one bucket for the whole trunk (2.2 M parameters)

    def __init__(self, norm, features, proj, out_size):
        mods = [norm, features] + ([proj] if proj is not None else [])
        super().__init__(*mods)
        self.__dict__["proj"] = proj              # not registered twice: index 2 already owns it
        self.out_size = out_size
        self.precision = "fp32"

    @property
    def trunk_trainable(self):
        return any(p.requires_grad for p in self[1][0].parameters())

    def layers(self):
        """((stem conv, bn), [inverted residuals], (last conv, bn)) looked up once"""
        ls = self.__dict__.get("_layers")
        if ls is None:
            f = list(self[1])
            ls = self.__dict__["_layers"] = ((f[0][0], f[0][1]), f[1:-1], (f[-1][0], f[-1][1]))
        return ls

    def forward(self, img):
        params = self.__dict__.get("_plist")
        if params is None:
            params = self.__dict__["_plist"] = list(self.parameters())
        return MobileNetEncoderFn.apply(img, self, *params)

    def _apply(self, fn, *a, **k):
        self.__dict__.pop("_plist", None)
        return super()._apply(fn, *a, **k)


def _load_torchvision_trunk(path, features):
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    sd = {k: v for k, v in sd.items() if not k.startswith("classifier.")}          # model.py:39 drops the classifier
    holder = nn.Module()
    holder.features = features
    missing, unexpected = holder.load_state_dict(sd, strict=False)
    missing = [k for k in missing if not k.endswith("num_batches_tracked")]
    if missing or unexpected:
        raise RuntimeError("pretrained checkpoint %s does not fit: missing %s, unexpected %s" % (path, missing[:5], list(unexpected)[:5]))


def _probe_zero_image(features, size):
    """model.py:46-48 pushes one all-zero image through the train-mode trunk: its only lasting effect is on the BatchNorm buffers.
    Initialisation-time host arithmetic on a single image (torch CPU ops), not part of the step."""
    with torch.no_grad():
        def seq(x, mods):
            for m in mods:
                if isinstance(m, nn.BatchNorm2d):
                    x = F.batch_norm(x, m.running_mean, m.running_var, m.weight, m.bias, True, m.momentum, m.eps); m.num_batches_tracked += 1
                elif isinstance(m, nn.Conv2d):
                    x = F.conv2d(x, m.weight, None, m.stride, m.padding, 1, m.groups)
                elif isinstance(m, nn.Sequential):
                    x = seq(x, m)
                else:
                    x = F.relu6(x)
            return x
        x = torch.zeros(1, 3, size, size)
        for m in features:
            x = (x + seq(x, m.conv) if m.use_res_connect else seq(x, m.conv)) if isinstance(m, InvertedResidual) else seq(x, m)


def get_mobilenet_encoder(args):
    """Reference get_encoder (model.py:16-63) for mobilenet_v2 (called by ``encoder.get_encoder``)."""
    ckpt = E._pretrained_file("mobilenet_v2", getattr(args, "pretrained", False))
    # construction order and initialisers = torchvision's: the RNG stream stays aligned with the reference's
    cin, last = 32, 1280
    feats = [conv_bn_relu6(3, cin, stride=2)]
    for tt, c, n, s in SETTING:
        for i in range(n):
            feats.append(InvertedResidual(cin, c, s if i == 0 else 1, tt))
            cin = c
    feats.append(conv_bn_relu6(cin, last, k=1))
    features = nn.Sequential(*feats)
    fc = nn.Linear(last, 1000)        # the classifier the reference drops (model.py:39): built (and initialised) for the RNG stream only
    for mod in features.modules():
        if isinstance(mod, nn.Conv2d):
            nn.init.kaiming_normal_(mod.weight, mode="fan_out")
    nn.init.normal_(fc.weight, 0, 0.01)
    if ckpt is None:
        # model.py:46-48: the zero image of the shape probe.  Zero biases: every BatchNorm sees an all-zero batch
        for sub in features.modules():
            if isinstance(sub, nn.BatchNorm2d):
                sub.running_var.fill_(0.9); sub.num_batches_tracked.fill_(1)
    else:
        _load_torchvision_trunk(ckpt, features)
        for prm in features.parameters():
            prm.requires_grad = False
        _probe_zero_image(features, int(args.input_size))
    s = int(args.input_size)
    for _ in range(5):                # the stem and four stride-2 blocks: 3x3 windows, stride 2, pad 1
        s = (s + 2 - 3) // 2 + 1
    proj = None
    if getattr(args, "encoder_dim", None) is not None and args.encoder_dim != last:
        proj = nn.Conv2d(last, args.encoder_dim, kernel_size=1, stride=1, bias=True)      # model.py:53
    else:
        args.encoder_dim = last
    es = getattr(args, "encoder_size", None)
    enc = HipMobileNetEncoder(E.Normalize(args.mean, args.std, inplace=True), features, proj, es if (es is not None and es != s) else None)
    E._channels_last_(enc)
    E._shadow_(enc)
    return enc
