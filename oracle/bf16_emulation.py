"""TEST INFRASTRUCTURE ONLY -- the CPU oracle with bf16 ROUNDING AT THE HIP PATH'S STORAGE POINTS.

The product's ``hip_precision="bf16"`` mode (BASELINE configs[1]) has no reference counterpart (the reference
only has fp16 AMP).  Its arithmetic is: fp32 accumulation everywhere, bf16 STORAGE of the encoder's activations and of
the operand copies the matrix cores read.  Comparing it with the fp32 oracle mixes two things -- the rounding the mode is
*designed* to make and any real bug -- and on a randomly initialised ResNet the first is large (ReLU decisions flip, the
flips compound through 16 residual blocks).  This module restates the SAME arithmetic on the CPU with ``x.bfloat16()``
applied where the HIP path stores bf16 (show-attend-and-tell-pytorch-lightning_amd/encoder.py, csrc/decoder.hip), so that ReLU / max-pool
decisions agree and what remains is accumulation-order noise:

  encoder  normalised image -> bf16;  filter copies -> bf16 (gradients of the fp32 master weights stay unrounded);
           every convolution output -> bf16 (its gradient, the BatchNorm dx, is stored bf16 too);
           every BatchNorm(+residual)(+ReLU) output -> bf16 (its gradient, a dgrad output, is stored bf16 too);
           statistics from the rounded convolution output, in fp32; stem tail = bn1 -> relu -> maxpool, pooled map -> bf16;
           1x1 projection: bf16 x bf16 -> fp32 annotations + fp32 bias, the incoming gradient is cast to bf16 for its GEMMs;
  decoder  every Linear / LSTM product rounds both operands to bf16 (fp32 accumulate) in forward and in both backward GEMMs;
           the attention context (and its backward, dalpha) reads a bf16 copy of the annotations; everything else pointwise
           (scores / softmax, cell, losses) is fp32.

Only tests import this file."""
import contextlib

import torch
import torch.nn.functional as F
from torch import nn

from . import sat_oracle as O


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


class _RoundBoth(torch.autograd.Function):
    """value stored as bf16 in the forward pass, its gradient stored as bf16 in the backward pass"""

    @staticmethod
    def forward(ctx, x):
        return bf(x)

    @staticmethod
    def backward(ctx, g):
        return bf(g)


class _RoundFwd(torch.autograd.Function):
    """bf16 copy of an fp32 master tensor: the gradient of the master stays fp32"""

    @staticmethod
    def forward(ctx, x):
        return bf(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    """fp32 value whose gradient is cast to bf16 before it is used"""

    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return bf(g)


rb, rf, rg = _RoundBoth.apply, _RoundFwd.apply, _RoundBwd.apply


def _conv(x, conv):
    return rb(F.conv2d(x, rf(conv.weight), None, conv.stride, conv.padding, 1, conv.groups))


def _bn(x, bn, residual=None, relu=True):
    y = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training, 0.1 if bn.momentum is None else bn.momentum, bn.eps)
    if bn.training:
        bn.num_batches_tracked += 1
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def _block(blk, x):
    skip = x if blk.downsample is None else rb(_bn(_conv(x, blk.downsample[0]), blk.downsample[1], None, False))
    y = rb(_bn(_conv(x, blk.conv1), blk.bn1))
    if blk.kind == "basic":
        return rb(_bn(_conv(y, blk.conv2), blk.bn2, skip))
    y = rb(_bn(_conv(y, blk.conv2), blk.bn2))
    return rb(_bn(_conv(y, blk.conv3), blk.bn3, skip))


def _dwconv(x, conv):
    """depthwise 3x3 of the shufflenet units: the kernel reads the fp32 master filter (csrc/depthwise.hip), only the activations are bf16"""
    return rb(F.conv2d(x, conv.weight, None, conv.stride, conv.padding, 1, conv.groups))


def _shuffle_branch(x, mods, residual=None):
    """conv / BatchNorm / ReLU | ReLU6 chain (nested Sequentials flattened); bf16 storage after every convolution and after every
    BatchNorm(+ activation); ``residual`` is added inside the LAST BatchNorm's kernel (fp32, before the one rounding)"""
    flat = []

    def walk(ms):
        for m in ms:
            walk(m) if isinstance(m, nn.Sequential) else flat.append(m)
    walk(mods)
    last_bn = max(i for i, m in enumerate(flat) if isinstance(m, nn.BatchNorm2d))
    for i, m in enumerate(flat):
        if isinstance(m, nn.Conv2d):
            x = _dwconv(x, m) if m.groups > 1 else _conv(x, m)
        elif isinstance(m, nn.BatchNorm2d):
            act = flat[i + 1] if i + 1 < len(flat) else None
            y = _bn(x, m, residual if i == last_bn else None, isinstance(act, nn.ReLU))
            x = rb(torch.clamp(y, 0.0, 6.0) if isinstance(act, nn.ReLU6) else y)
    return x


def _mobilenet_forward(mods, x):
    """children: Normalize, features (stem ConvBNReLU6, inverted residuals, last 1x1 ConvBNReLU6)[, 1x1 conv][, resize]"""
    feats = list(mods[1])
    conv1, bn1 = feats[0][0], feats[0][1]
    x = rb(F.conv2d(x, rf(conv1.weight), None, conv1.stride, conv1.padding))
    x = rb(torch.clamp(_bn(x, bn1, None, False), 0.0, 6.0))
    for blk in feats[1:-1]:
        x = _shuffle_branch(x, blk.conv, residual=x if blk.use_res_connect else None)
    return _shuffle_branch(x, feats[-1]), mods[2:]


def _shufflenet_forward(mods, x):
    """children: Normalize, conv1 (conv, bn, relu), maxpool, stage2..4, conv5 (conv, bn, relu)[, 1x1 conv][, resize]"""
    conv1, bn1 = mods[1][0], mods[1][1]
    x = rb(F.conv2d(x, rf(conv1.weight), None, conv1.stride, conv1.padding))
    x = rb(F.max_pool2d(_bn(x, bn1), 3, 2, 1))
    for stage in mods[3:6]:
        for u in stage:
            if u.stride == 1:
                x1, x2 = x.chunk(2, dim=1)
                out = torch.cat((x1, _shuffle_branch(x2, u.branch2)), 1)
            else:
                out = torch.cat((_shuffle_branch(x, u.branch1), _shuffle_branch(x, u.branch2)), 1)
            x = O.channel_shuffle(out, 2)
    x = _shuffle_branch(x, mods[6])
    return x, mods[7:]


def encoder_forward(enc, img):
    """``enc`` = oracle.build_encoder(hp) (children: Normalize, conv1, bn1, relu, maxpool, layer1..4[, 1x1 conv][, resize]; or the
    shufflenet_v2 children); img (B, 3, H, W) fp32 in [0, 1].  Returns the annotations (B, D, h, w) fp32."""
    mods = list(enc.children())
    if isinstance(mods[1], nn.Sequential):
        norm = mods[0]
        m = torch.as_tensor(norm.mean, dtype=torch.float32).view(1, -1, 1, 1); s = torch.as_tensor(norm.std, dtype=torch.float32).view(1, -1, 1, 1)
        mobilenet = len(mods[1]) > 3                    # mobilenet_v2: one ``features`` child; shufflenet: conv1 (conv, bn, relu) first
        x, rest = (_mobilenet_forward if mobilenet else _shufflenet_forward)(mods, bf((img - m) / s))
        for mod in rest:
            if isinstance(mod, nn.Conv2d):
                x = rg(F.conv2d(x, rf(mod.weight), None)) + mod.bias.view(1, -1, 1, 1)
            else:
                x = mod(x)
        return x
    norm, conv1, bn1 = mods[0], mods[1], mods[2]
    m = torch.as_tensor(norm.mean, dtype=torch.float32).view(1, -1, 1, 1); s = torch.as_tensor(norm.std, dtype=torch.float32).view(1, -1, 1, 1)
    x = bf((img - m) / s)
    x = rb(F.conv2d(x, rf(conv1.weight), None, conv1.stride, conv1.padding))
    x = rb(F.max_pool2d(_bn(x, bn1), 3, 2, 1))
    for layer in mods[5:9]:
        for blk in layer:
            x = _block(blk, x)
    for mod in mods[9:]:
        if isinstance(mod, nn.Conv2d):
            x = rg(F.conv2d(x, rf(mod.weight), None)) + mod.bias.view(1, -1, 1, 1)
        else:
            x = mod(x)                                  # encoder_size resize, fp32
    return x


class _LinearBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        xb, wb = bf(x), bf(w)
        ctx.save_for_backward(xb, wb)
        ctx.has_b = b is not None
        y = xb @ wb.t()
        return y + b if b is not None else y

    @staticmethod
    def backward(ctx, dy):
        xb, wb = ctx.saved_tensors
        dyb = bf(dy)
        dx = dyb @ wb
        dw = dyb.reshape(-1, dyb.shape[-1]).t() @ xb.reshape(-1, xb.shape[-1])
        db = dy.reshape(-1, dy.shape[-1]).sum(0) if ctx.has_b else None
        return dx, dw, db


class _F:
    """stand-in for ``torch.nn.functional`` inside oracle.sat_oracle: ``linear`` rounds its operands like the bf16 matrix cores"""

    def __getattr__(self, k):
        return getattr(F, k)

    @staticmethod
    def linear(x, w, b=None):
        if w.shape[0] == 1:                              # attention.f_att: part of the fp32 score kernel, not a GEMM
            return F.linear(x, w, b)
        return _LinearBF16.apply(x, w, b)


def _soft_attention_bf16(sd, ann, h):
    """``O.soft_attention`` (model.py:94-109) with the annotations of the CONTEXT product rounded to bf16: the HIP path keeps a bf16 copy of the
    annotations for the two kernels that stream them every time step (context forward, dalpha backward); scores, softmax, sums are fp32 and the
    gradient that reaches the annotations through the context (alpha x dz) is not rounded."""
    N, D, H, W = ann.shape
    a = ann.reshape(N, D, H * W).permute(0, 2, 1)
    u = O.F.linear(a, sd["attention.encoder_att.weight"])
    q = O.F.linear(h, sd["attention.decoder_att.weight"]).unsqueeze(1)
    s = O.F.linear(torch.tanh(u + q), sd["attention.f_att.weight"]) * a.shape[1] ** -0.5
    alpha = F.softmax(s, dim=1)
    z = (rf(a) * alpha).sum(dim=1)
    return z, alpha.permute(0, 2, 1).reshape(N, H, W)


@contextlib.contextmanager
def decoder_bf16_products():
    old, old_att = O.F, O.soft_attention
    O.F = _F(); O.soft_attention = _soft_attention_bf16
    try:
        yield
    finally:
        O.F = old; O.soft_attention = old_att


def step_loss(oracle, img, caps, lengths, epsilon=1.0, draw=None):
    """``OracleSAT.step_loss`` with the rounding points above (the LSTM step is taken through its written-out form so that its
    products can be rounded)."""
    ann = encoder_forward(oracle.encoder, img)
    with decoder_bf16_products():
        out = O.decode_train(oracle.sd, oracle.hp, ann, caps, lengths, epsilon, draw, lstm_fn=O.lstm_step_math)
    lp, bs = O.pack_time_major(out["logits"], out["lengths"])
    tp, _ = O.pack_time_major(out["targets"], out["lengths"])
    ce = O.label_smoothing_ce(lp, tp, oracle.hp.label_smoothing)
    ds = O.doubly_stochastic(out["alphas"], oracle.hp.att_gamma)
    out.update(logits_packed=lp, targets_packed=tp, batch_sizes=bs, ce=ce, ds=ds, loss=ce + ds, annotations=ann)
    return ce + ds, out
