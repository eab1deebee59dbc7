"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the Show-Attend-and-Tell
train-step hot path.

This file restates, in plain fp32 PyTorch-CPU ops over a *state dict*, what the
reference computes.  Every function cites the reference lines it follows
(paths are under /root/reference).  It is the checker for the HIP path and the
"port" CPU baseline of bench.py; it is never imported by the product package.

Pinning: ``tests/golden/*.npz`` were produced by importing the reference's own
``model.py`` in the build container (``tests/golden/make_golden.py``) and
``tests/test_oracle_golden.py`` checks this restatement against them
(decoder rows a2-a10: pinned).  The torchvision ResNet arithmetic (row a1) is
third party and absent from the reference tree: **encoder parity is unpinned
at the reference level**; it is pinned only structurally (parameter counts /
feature dims from dev/encoder_summaries.txt:2-18 and the 256px -> 8x8 map).

Dropout is not restated: parity runs use dropout = embedding_dropout = 0
(train.py:140-143 defaults), where ``nn.Dropout`` is the identity.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F
from torch import nn

# --------------------------------------------------------------------------
# a1. Encoder: torchvision-style ResNet trunk (third-party arithmetic; the
#     reference only slices it: model.py:19-29).  Key names follow torchvision
#     so real weights could be loaded from a local file.
# --------------------------------------------------------------------------

#: arch -> (block kind, blocks per stage, width per group[, groups of the 3x3 convolution: the resnext archs, model.py:28])
RESNET_TABLE = {
    "resnet18": ("basic", (2, 2, 2, 2), 64),
    "resnet34": ("basic", (3, 4, 6, 3), 64),
    "resnet50": ("bottleneck", (3, 4, 6, 3), 64),
    "resnet101": ("bottleneck", (3, 4, 23, 3), 64),
    "resnet152": ("bottleneck", (3, 8, 36, 3), 64),
    "wide_resnet50_2": ("bottleneck", (3, 4, 6, 3), 128),
    "wide_resnet101_2": ("bottleneck", (3, 4, 23, 3), 128),
    "resnext50_32x4d": ("bottleneck", (3, 4, 6, 3), 4, 32),
    "resnext101_32x8d": ("bottleneck", (3, 4, 23, 3), 8, 32),
}


class _Residual(nn.Module):
    """One residual unit.  ``kind`` selects 3x3-3x3 or 1x1-3x3-1x1 (stride on
    the 3x3, i.e. the "v1.5" layout torchvision uses)."""

    def __init__(self, kind, cin, planes, stride, width_per_group, groups=1):
        super().__init__()
        self.kind = kind
        if kind == "basic":
            cout = planes
            self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes)
        else:
            cout = planes * 4
            mid = int(planes * (width_per_group / 64.0)) * groups          # torchvision Bottleneck: width = int(planes * base_width / 64) * groups
            self.conv1 = nn.Conv2d(cin, mid, 1, 1, 0, bias=False)
            self.bn1 = nn.BatchNorm2d(mid)
            self.conv2 = nn.Conv2d(mid, mid, 3, stride, 1, groups=groups, bias=False)
            self.bn2 = nn.BatchNorm2d(mid)
            self.conv3 = nn.Conv2d(mid, cout, 1, 1, 0, bias=False)
            self.bn3 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, cout, 1, stride, 0, bias=False), nn.BatchNorm2d(cout))
        self.cout = cout

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        if self.kind == "basic":
            y = self.bn2(self.conv2(y))
        else:
            y = self.relu(self.bn2(self.conv2(y)))
            y = self.bn3(self.conv3(y))
        y += skip
        return self.relu(y)


class ResNetOracle(nn.Module):
    """Child order conv1,bn1,relu,maxpool,layer1..4,avgpool,fc so that the
    reference's ``list(m.children())[:-2]`` (model.py:29) keeps the trunk."""

    def __init__(self, arch, num_classes=1000):
        super().__init__()
        kind, depths, wpg = RESNET_TABLE[arch][:3]
        groups = RESNET_TABLE[arch][3] if len(RESNET_TABLE[arch]) > 3 else 1
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for si, (planes, nblk) in enumerate(zip((64, 128, 256, 512), depths)):
            blocks = []
            for bi in range(nblk):
                blk = _Residual(kind, cin, planes, (2 if si > 0 and bi == 0 else 1), wpg, groups)
                cin = blk.cout
                blocks.append(blk)
            setattr(self, "layer%d" % (si + 1), nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(cin, num_classes)
        self.feature_dim = cin
        for mod in self.modules():
            if isinstance(mod, nn.Conv2d):
                nn.init.kaiming_normal_(mod.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(mod, nn.BatchNorm2d):
                nn.init.ones_(mod.weight)
                nn.init.zeros_(mod.bias)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


# torchvision's ShuffleNetV2 (third party, absent from the reference tree like the ResNets): the reference's CLI default encoder_arch is
# shufflenet_v2_x0_5 (train.py:43) and get_encoder keeps every child but the classifier (model.py:30-31).  Restated from the published
# architecture (Ma et al. 2018, "ShuffleNet V2", fig. 3 c/d and table 5); pinned structurally by dev/encoder_summaries.txt:28-35
# (features 1024 / 2048, 0.34 / 1.25 / 2.48 / 5.34 M parameters without the classifier).
#: arch -> (units per stage, output channels of conv1, stage2, stage3, stage4, conv5)
SHUFFLENET_TABLE = {
    "shufflenet_v2_x0_5": ((4, 8, 4), (24, 48, 96, 192, 1024)),
    "shufflenet_v2_x1_0": ((4, 8, 4), (24, 116, 232, 464, 1024)),
    "shufflenet_v2_x1_5": ((4, 8, 4), (24, 176, 352, 704, 1024)),
    "shufflenet_v2_x2_0": ((4, 8, 4), (24, 244, 488, 976, 2048)),
}


def channel_shuffle(x, groups):
    """(B, g * n, H, W): channel j * n + i -> channel i * g + j (the transpose of the (g, n) channel grid)"""
    B, Cc, H, W = x.shape
    return x.view(B, groups, Cc // groups, H, W).transpose(1, 2).reshape(B, Cc, H, W)


class _ShuffleUnit(nn.Module):
    """One ShuffleNetV2 unit.  stride 1: the first half of the channels passes through, the second goes through 1x1 -> depthwise 3x3 -> 1x1;
    stride 2: both branches see the whole input (the left one is depthwise 3x3 -> 1x1).  Then the halves are concatenated and shuffled."""

    def __init__(self, inp, oup, stride):
        super().__init__()
        self.stride = stride
        bfeat = oup // 2
        assert stride != 1 or inp == bfeat << 1

        def dw(c, s):
            return nn.Conv2d(c, c, 3, s, 1, bias=False, groups=c)

        if stride > 1:
            self.branch1 = nn.Sequential(dw(inp, stride), nn.BatchNorm2d(inp), nn.Conv2d(inp, bfeat, 1, 1, 0, bias=False), nn.BatchNorm2d(bfeat),
                                         nn.ReLU(inplace=True))
        else:
            self.branch1 = nn.Sequential()
        self.branch2 = nn.Sequential(nn.Conv2d(inp if stride > 1 else bfeat, bfeat, 1, 1, 0, bias=False), nn.BatchNorm2d(bfeat), nn.ReLU(inplace=True),
                                     dw(bfeat, stride), nn.BatchNorm2d(bfeat), nn.Conv2d(bfeat, bfeat, 1, 1, 0, bias=False), nn.BatchNorm2d(bfeat),
                                     nn.ReLU(inplace=True))

    def forward(self, x):
        if self.stride == 1:
            x1, x2 = x.chunk(2, dim=1)
            out = torch.cat((x1, self.branch2(x2)), dim=1)
        else:
            out = torch.cat((self.branch1(x), self.branch2(x)), dim=1)
        return channel_shuffle(out, 2)


class ShuffleNetOracle(nn.Module):
    """Child order conv1, maxpool, stage2, stage3, stage4, conv5, fc so that the reference's ``list(m.children())[:-1]`` (model.py:31) keeps
    the trunk.  Default PyTorch initialisation (the architecture defines none of its own)."""

    def __init__(self, arch, num_classes=1000):
        super().__init__()
        repeats, chans = SHUFFLENET_TABLE[arch]
        self.conv1 = nn.Sequential(nn.Conv2d(3, chans[0], 3, 2, 1, bias=False), nn.BatchNorm2d(chans[0]), nn.ReLU(inplace=True))
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = chans[0]
        for name, rep, cout in zip(("stage2", "stage3", "stage4"), repeats, chans[1:4]):
            units = [_ShuffleUnit(cin, cout, 2)] + [_ShuffleUnit(cout, cout, 1) for _ in range(rep - 1)]
            setattr(self, name, nn.Sequential(*units))
            cin = cout
        self.conv5 = nn.Sequential(nn.Conv2d(cin, chans[4], 1, 1, 0, bias=False), nn.BatchNorm2d(chans[4]), nn.ReLU(inplace=True))
        self.fc = nn.Linear(chans[4], num_classes)
        self.feature_dim = chans[4]

    def forward(self, x):
        x = self.conv5(self.stage4(self.stage3(self.stage2(self.maxpool(self.conv1(x))))))
        return self.fc(x.mean([2, 3]))


# torchvision's MobileNetV2 (third party, absent from the reference tree): get_encoder keeps ``m.features`` (model.py:38-39 drops the classifier).
# Restated from the published architecture (Sandler et al. 2018, table 2) with torchvision's module layout (ConvBNReLU = Sequential(conv, bn,
# ReLU6); InvertedResidual.conv = [expand 1x1 ConvBNReLU unless t == 1, depthwise 3x3 ConvBNReLU, 1x1 Conv2d, BatchNorm2d]) and initialisers
# (kaiming_normal fan_out, BatchNorm ones / zeros, Linear normal(0, 0.01)); pinned structurally by dev/encoder_summaries.txt:36-37 (1280
# features, 2.22 M parameters without the classifier).
MOBILENET_V2_SETTING = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))      # t, c, n, s


def _conv_bn_relu6(cin, cout, k=3, stride=1, groups=1):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False), nn.BatchNorm2d(cout), nn.ReLU6(inplace=True))


class _InvertedResidual(nn.Module):
    def __init__(self, inp, oup, stride, expand_ratio):
        super().__init__()
        hidden = int(round(inp * expand_ratio))
        self.stride = stride
        self.use_res_connect = stride == 1 and inp == oup
        layers = []
        if expand_ratio != 1:
            layers.append(_conv_bn_relu6(inp, hidden, k=1))
        layers.extend([_conv_bn_relu6(hidden, hidden, stride=stride, groups=hidden), nn.Conv2d(hidden, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup)])
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        return x + self.conv(x) if self.use_res_connect else self.conv(x)


class MobileNetV2Oracle(nn.Module):
    """Children: features, classifier - the reference's ``list(m.children())[:-1]`` (model.py:39) keeps ``features``."""

    def __init__(self, num_classes=1000):
        super().__init__()
        cin, last = 32, 1280
        feats = [_conv_bn_relu6(3, cin, stride=2)]
        for t, c, n, s in MOBILENET_V2_SETTING:
            for i in range(n):
                feats.append(_InvertedResidual(cin, c, s if i == 0 else 1, t))
                cin = c
        feats.append(_conv_bn_relu6(cin, last, k=1))
        self.features = nn.Sequential(*feats)
        self.classifier = nn.Sequential(nn.Dropout(0.2), nn.Linear(last, num_classes))
        self.feature_dim = last
        for mod in self.modules():
            if isinstance(mod, nn.Conv2d):
                nn.init.kaiming_normal_(mod.weight, mode="fan_out")
            elif isinstance(mod, nn.BatchNorm2d):
                nn.init.ones_(mod.weight); nn.init.zeros_(mod.bias)
            elif isinstance(mod, nn.Linear):
                nn.init.normal_(mod.weight, 0, 0.01); nn.init.zeros_(mod.bias)

    def forward(self, x):
        return self.classifier(self.features(x).mean([2, 3]))


class NormalizeInplace(nn.Module):
    """torchvision.transforms.Normalize(mean, std, inplace=True) on a batch:
    ``x.sub_(mean).div_(std)`` (used at model.py:59; mutates its input, F9)."""

    def __init__(self, mean, std, inplace=True):
        super().__init__()
        self.mean, self.std, self.inplace = list(mean), list(std), inplace

    def forward(self, x):
        if not self.inplace:
            x = x.clone()
        m = torch.as_tensor(self.mean, dtype=x.dtype, device=x.device).view(-1, 1, 1)
        s = torch.as_tensor(self.std, dtype=x.dtype, device=x.device).view(-1, 1, 1)
        return x.sub_(m).div_(s)


def resnet_factory(arch):
    """Callable with torchvision's ``models.<arch>(pretrained=...)`` signature."""

    def make(pretrained=False, **_):
        if pretrained:
            raise RuntimeError("no pretrained weights offline (SURVEY 8c)")
        return ResNetOracle(arch)

    make.__name__ = arch
    return make


def build_encoder(hp):
    """Restates get_encoder (model.py:16-63) for the resnet, shufflenet_v2 and mobilenet_v2 families, plus the
    ``encoder_size`` resize the README documents (readme.md:118-121, F2).

    Returns nn.Sequential whose state-dict keys equal the reference's
    (``1.weight`` conv1, ``2.*`` bn1, ``5..8`` layer1..4, ``9.*`` 1x1 proj)."""
    if hp.encoder_arch in SHUFFLENET_TABLE:          # model.py:30-31 (keys: 1.* conv1, 3..5 stage2..4, 6.* conv5, 7.* 1x1 projection)
        net = ShuffleNetOracle(hp.encoder_arch)
        trunk = [net.conv1, net.maxpool, net.stage2, net.stage3, net.stage4, net.conv5]
    elif hp.encoder_arch == "mobilenet_v2":           # model.py:38-39 (keys: 1.<i>.* features, 2.* 1x1 projection)
        net = MobileNetV2Oracle()
        trunk = [net.features]
    elif hp.encoder_arch in RESNET_TABLE:
        net = ResNetOracle(hp.encoder_arch)
        trunk = [net.conv1, net.bn1, net.relu, net.maxpool, net.layer1, net.layer2, net.layer3, net.layer4]
    else:
        raise ValueError("Encoder not supported : {}".format(hp.encoder_arch))
    # model.py:46-48: a zero image is pushed through the (train-mode) trunk to read the
    # feature dim; as a side effect every BatchNorm's running stats see one batch.
    probe = nn.Sequential(*trunk)(torch.zeros(1, 3, hp.input_size, hp.input_size))
    final_dim, final_size = probe.shape[1], probe.shape[-1]
    if getattr(hp, "encoder_dim", None) is not None and hp.encoder_dim != final_dim:
        trunk.append(nn.Conv2d(final_dim, hp.encoder_dim, kernel_size=1, stride=1, bias=True))
    else:
        hp.encoder_dim = final_dim
    es = getattr(hp, "encoder_size", None)
    if es is not None:
        if es < final_size:
            trunk.append(nn.AdaptiveAvgPool2d((es, es)))
        elif es > final_size:
            trunk.append(nn.Upsample((es, es), mode="bilinear", align_corners=False))
    return nn.Sequential(NormalizeInplace(hp.mean, hp.std, inplace=True), *trunk)


def trunk_param_count(arch):
    """Parameters of the trunk without fc (dev/encoder_summaries.txt:2-18)."""
    net = ShuffleNetOracle(arch) if arch in SHUFFLENET_TABLE else (MobileNetV2Oracle() if arch == "mobilenet_v2" else ResNetOracle(arch))
    return sum(p.numel() for n, p in net.named_parameters() if not n.startswith(("fc.", "classifier."))), net.feature_dim


# --------------------------------------------------------------------------
# a2-a7. Decoder pieces over a state dict ``sd`` (reference key names,
#        SURVEY 8b).  All tensors fp32 CPU.
# --------------------------------------------------------------------------

def init_state(sd, ann, layers, n, mean_mask=None):
    """InitLSTM.forward (model.py:76-81).  ann (N,D,h,w) -> h0,c0 (layers,N,n).
    The ``reshape`` is a raw reinterpretation of the (N, 2*layers*n) buffer --
    no permute -- so rows mix across the batch.  ``mean_mask`` (N,D) stands
    for ``self.dropout`` (model.py:78): kept elements carry 1/(1-p)."""
    mean = ann.mean((2, 3))
    if mean_mask is not None:
        mean = mean * mean_mask
    v = F.linear(F.linear(mean, sd["init_lstm.factorize.weight"], sd["init_lstm.factorize.bias"]),
                 sd["init_lstm.init.weight"], sd["init_lstm.init.bias"])
    flat = v.reshape(2 * layers, mean.shape[0], n)
    return flat[:layers], flat[layers:]


def soft_attention(sd, ann, h):
    """SoftAttention.forward (model.py:94-109).  ann (N,D,h,w), h (N,n) ->
    z (N,D), alpha (N,h,w).  Scores are scaled by L**-0.5 (F8)."""
    N, D, H, W = ann.shape
    a = ann.reshape(N, D, H * W).permute(0, 2, 1)                      # (N,L,D)
    u = F.linear(a, sd["attention.encoder_att.weight"])                # (N,L,A)  F4: loop invariant
    q = F.linear(h, sd["attention.decoder_att.weight"]).unsqueeze(1)   # (N,1,A)
    s = F.linear(torch.tanh(u + q), sd["attention.f_att.weight"]) * a.shape[1] ** -0.5
    alpha = F.softmax(s, dim=1)                                        # (N,L,1)
    z = (a * alpha).sum(dim=1)
    return z, alpha.permute(0, 2, 1).reshape(N, H, W)


def beta_gate(sd, h):
    """self.beta = Sigmoid(Linear(n->D)) (model.py:187-192, used at 538)."""
    return torch.sigmoid(F.linear(h, sd["beta.0.weight"], sd["beta.0.bias"]))


def embed(sd, idx, max_norm=None, padding_idx=0):
    """self.embedding (model.py:158-163).  ``max_norm`` renormalises the
    looked-up rows of the weight in place, as nn.Embedding does."""
    return F.embedding(idx, sd["embedding.weight"], padding_idx=padding_idx, max_norm=max_norm)


def lstm_weights(sd, layers):
    out = []
    for k in range(layers):
        out += [sd["lstm.weight_ih_l%d" % k], sd["lstm.weight_hh_l%d" % k],
                sd["lstm.bias_ih_l%d" % k], sd["lstm.bias_hh_l%d" % k]]
    return out


def lstm_step(sd, x, h, c, layers):
    """One time step of nn.LSTM (model.py:175-180, calls at 326/544) through the
    same ATen entry point nn.LSTM.forward uses.  x (1,N,m+D); h,c (layers,N,n)."""
    _, hn, cn = torch._VF.lstm(x, (h, c), lstm_weights(sd, layers), True, layers, 0.0, False, False, False)
    return hn, cn


def lstm_step_math(sd, x, h, c, layers):
    """The same step written out (gate order i,f,g,o; two biases; F1):
    g = x W_ih^T + b_ih + h W_hh^T + b_hh; c' = s(f) c + s(i) tanh(g); h' = s(o) tanh(c')."""
    hs, cs = [], []
    inp = x[0]
    for k in range(layers):
        g = (F.linear(inp, sd["lstm.weight_ih_l%d" % k], sd["lstm.bias_ih_l%d" % k])
             + F.linear(h[k], sd["lstm.weight_hh_l%d" % k], sd["lstm.bias_hh_l%d" % k]))
        i, f, gg, o = g.chunk(4, dim=1)
        cn = torch.sigmoid(f) * c[k] + torch.sigmoid(i) * torch.tanh(gg)
        hn = torch.sigmoid(o) * torch.tanh(cn)
        hs.append(hn)
        cs.append(cn)
        inp = hn
    return torch.stack(hs), torch.stack(cs)


def deep_output(sd, y, h, z, deep=True, x_mask=None):
    """DeepOutput.forward (model.py:125-131): un-gated context z and the *new*
    hidden state (F8).  Tied weights simply alias output.output.weight.
    ``x_mask`` stands for ``self.dropout`` (model.py:130)."""
    if deep:
        x = torch.tanh(y + F.linear(h, sd["output.hidden.weight"]) + F.linear(z, sd["output.context.weight"]))
    else:
        x = F.linear(h, sd["output.hidden.weight"])
    if x_mask is not None:
        x = x * x_mask
    return F.linear(x, sd["output.output.weight"], sd.get("output.output.bias"))


# --------------------------------------------------------------------------
# a8. train_batch (model.py:474-557)
# --------------------------------------------------------------------------

def decode_train(sd, hp, ann_img, caps, lengths, epsilon=0, draw=None, lstm_fn=lstm_step, masks=None):
    """Decoder half of train_batch: everything after ``self.encoder(img)``.

    ann_img (B,D,h,w); caps (B,R,T) int64; lengths (B,R) int64.
    Returns dict(logits (N,T-1,V), alphas (N,T-1,L), targets (N,T-1), lengths (N,)).
    ``draw()`` supplies the per-step uniform sample of model.py:518 (default:
    the global CPU generator, consumed exactly as the reference does, F7)."""
    if draw is None:
        draw = lambda: float(torch.rand(1))
    layers, n, V = hp.decoder_layers, hp.decoder_dim, hp.vocab_size
    R = lengths.size(1)
    ann = ann_img.repeat_interleave(R, dim=0)                 # model.py:487 (F5)
    _, _, Hh, Ww = ann.shape
    caps = caps.reshape(-1, caps.size(2))
    lens = lengths.reshape(-1)
    # masks (dropout restated with explicit masks): init (N,D), emb (T-1,N,m), out (T-1,N,m); None = no dropout
    masks = masks or {}
    h, c = init_state(sd, ann, layers, n, masks.get("init"))   # model.py:498
    h, c = h.clone(), c.clone()
    N, T = caps.shape
    logits = torch.zeros(N, T - 1, V)
    alphas = torch.zeros(N, T - 1, Hh * Ww)
    for step in range(T - 1):
        live = lens > step                                    # model.py:512 (F6)
        if not bool(live.any()):
            break
        if step <= 2 or draw() <= epsilon:                    # model.py:518 (F7)
            tok = caps[live, step]
        else:
            tok = torch.argmax(logits[live, step - 1, :], dim=1)
        y = embed(sd, tok, getattr(hp, "embed_norm", None))
        if masks.get("emb") is not None:
            y = y * masks["emb"][step][live]                  # embedding_dropout (model.py:526)
        z, alpha = soft_attention(sd, ann[live], h[-1, live])
        alphas[live, step, :] = alpha.reshape(int(live.sum()), Hh * Ww)
        gate = beta_gate(sd, h[-1, live])
        x = torch.cat([y, gate * z], dim=1).unsqueeze(0)
        hn, cn = lstm_fn(sd, x, h[:, live], c[:, live], layers)
        h = h.clone(); c = c.clone()
        h[:, live] = hn
        c[:, live] = cn
        xm = masks["out"][step][live] if masks.get("out") is not None else None
        logits[live, step, :] = deep_output(sd, y, h[-1, live], z, hp.deep_output, xm).float()
    return dict(logits=logits, alphas=alphas, targets=caps[:, 1:], lengths=lens)


def pack_time_major(x, lens):
    """``pack_padded_sequence(x, lens, batch_first=True, enforce_sorted=False)``
    (model.py:553-554): returns (.data, .batch_sizes as list).  Rows are ordered
    time-major over the batch sorted by length descending; ties follow
    ``torch.sort`` (not stable), so the library call itself is used."""
    packed = torch.nn.utils.rnn.pack_padded_sequence(x, lens.tolist(), batch_first=True, enforce_sorted=False)
    return packed.data, packed.batch_sizes.tolist()


# --------------------------------------------------------------------------
# a9/a10. Losses (util.py:105-112, model.py:592-597)
# --------------------------------------------------------------------------

def label_smoothing_ce(x, target, smoothing=0.0):
    """LabelSmoothing.forward (util.py:105-112)."""
    logp = F.log_softmax(x, dim=-1)
    nll = -logp.gather(dim=-1, index=target.unsqueeze(1)).squeeze(1)
    smooth = -logp.mean(dim=-1)
    return ((1.0 - smoothing) * nll + smoothing * smooth).mean()


def doubly_stochastic(alphas, gamma):
    """model.py:594: gamma * ((1 - sum_t alpha)^2).mean() over (N, L)."""
    return gamma * ((1 - alphas.sum(dim=1)) ** 2).mean()


def training_loss(sd, hp, ann_img, caps, lengths, epsilon=0, draw=None, masks=None):
    """model.py:589-597 without the logging: returns (loss, parts)."""
    out = decode_train(sd, hp, ann_img, caps, lengths, epsilon, draw, masks=masks)
    lp, bs = pack_time_major(out["logits"], out["lengths"])
    tp, _ = pack_time_major(out["targets"], out["lengths"])
    ce = label_smoothing_ce(lp, tp, hp.label_smoothing)
    ds = doubly_stochastic(out["alphas"], hp.att_gamma)
    loss = ce + ds
    pred = torch.argmax(lp, dim=1)
    acc = torch.sum(pred == tp) / pred.shape[0]
    out.update(logits_packed=lp, targets_packed=tp, batch_sizes=bs, ce=ce, ds=ds, loss=loss, acc=acc)
    return loss, out


# --------------------------------------------------------------------------
# Convenience: default hyper-parameters (train.py:16-165 defaults for the
# hot-path keys listed in SURVEY section 5) and random state dicts.
# --------------------------------------------------------------------------

def default_hparams(**over):
    V = over.get("vocab_size", 64)
    stoi = {"<PAD>": 0, "<UNK>": V - 3, "<START>": V - 2, "<END>": V - 1}
    hp = dict(encoder_arch="resnet18", pretrained=False, input_size=256, encoder_dim=256, encoder_size=None,
              mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225], embed_dim=256, embed_norm=None,
              attention_dim=128, decoder_dim=512, decoder_layers=1, dropout=0.0, embedding_dropout=0.0,
              label_smoothing=0.0, weight_tying=False, deep_output=True, att_gamma=1.0, vocab_size=V,
              vocab_stoi=stoi, vocab_itos={v: k for k, v in stoi.items()}, pretrained_embedding=None,
              decoder_tf="always", decoder_tf_min=0.5, epochs=10, encoder_finetune_after=-1)
    hp.update(over)
    return SimpleNamespace(**hp)


def random_decoder_state(hp, seed=0):
    """Decoder state dict with the reference's shapes and torch's
    default initialisers' scale; used when no reference module is at hand."""
    g = torch.Generator().manual_seed(seed)
    D, m, A, n, V, K = hp.encoder_dim, hp.embed_dim, hp.attention_dim, hp.decoder_dim, hp.vocab_size, hp.decoder_layers

    def U(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * b

    sd = {"embedding.weight": torch.randn(V, m, generator=g)}
    sd["embedding.weight"][0].zero_()
    sd["init_lstm.factorize.weight"] = U((m, D), D); sd["init_lstm.factorize.bias"] = U((m,), D)
    sd["init_lstm.init.weight"] = U((2 * n * K, m), m); sd["init_lstm.init.bias"] = U((2 * n * K,), m)
    for k in range(K):
        cin = m + D if k == 0 else n
        sd["lstm.weight_ih_l%d" % k] = U((4 * n, cin), n); sd["lstm.weight_hh_l%d" % k] = U((4 * n, n), n)
        sd["lstm.bias_ih_l%d" % k] = U((4 * n,), n); sd["lstm.bias_hh_l%d" % k] = U((4 * n,), n)
    sd["attention.encoder_att.weight"] = U((A, D), D)
    sd["attention.decoder_att.weight"] = U((A, n), n)
    sd["attention.f_att.weight"] = U((1, A), A)
    sd["beta.0.weight"] = U((D, n), n); sd["beta.0.bias"] = torch.full((D,), 1.0 / n)
    sd["output.hidden.weight"] = U((m, n), n)
    if hp.deep_output:
        sd["output.context.weight"] = U((m, D), D)
    if hp.weight_tying and hp.deep_output:
        sd["output.output.weight"] = sd["embedding.weight"]
    else:
        sd["output.output.weight"] = U((V, m), m)
    if not hp.weight_tying:
        sd["output.output.bias"] = U((V,), m)
    return sd


# --------------------------------------------------------------------------
# Whole model: encoder + decoder + losses = SAT.training_step's arithmetic
# (model.py:589-597).  Used by the parity tests and as bench.py's "port" CPU baseline.
# --------------------------------------------------------------------------

class OracleSAT:
    """``state_dict`` uses the reference's keys (``encoder.*`` + decoder keys)."""

    def __init__(self, hp, state_dict=None, seed=42):
        self.hp = hp
        torch.manual_seed(seed)
        self.encoder = build_encoder(hp)
        if state_dict is None:
            self.sd = random_decoder_state(hp, seed)
        else:
            enc = {k[len("encoder."):]: v for k, v in state_dict.items() if k.startswith("encoder.")}
            self.encoder.load_state_dict(enc)
            self.sd = {k: v.detach().clone() for k, v in state_dict.items() if not k.startswith("encoder.")}
        for v in self.sd.values():
            v.requires_grad_()
        if hp.weight_tying and hp.deep_output:
            self.sd["output.output.weight"] = self.sd["embedding.weight"]

    def parameters(self):
        seen, out = set(), []
        for p in list(self.encoder.parameters()) + list(self.sd.values()):
            if id(p) not in seen:
                seen.add(id(p)); out.append(p)
        return out

    def named_grads(self):
        g = {"encoder." + k: p.grad for k, p in self.encoder.named_parameters()}
        g.update({k: p.grad for k, p in self.sd.items()})
        return g

    def step_loss(self, img, caps, lengths, epsilon=1.0, draw=None):
        ann = self.encoder(img.clone())                    # clone: Normalize is in place (F9)
        return training_loss(self.sd, self.hp, ann, caps, lengths, epsilon, draw)


# --------------------------------------------------------------------------
# a11. Inference: SAT.forward / caption (model.py:237-472): beam search, the two sampled variants, decoder noise.
# --------------------------------------------------------------------------

def gumbel_topk(probs, k, gumbel):
    """An ordered sample of k distinct categories from the (unnormalised) distribution ``probs`` - what
    ``torch.multinomial(probs, k)`` (without replacement) draws, model.py:364,377 - taken as the top k of
    log p + Gumbel(0,1) noise (the Gumbel-top-k / Plackett-Luce identity).  ``gumbel``: one variate per category.
    The batched HIP search draws this way; tests/test_oracle_golden.py checks the identity against torch.multinomial."""
    return torch.topk(torch.log(probs) + gumbel, k).indices


def beam_search(sd, hp, ann_img, beamk=3, max_gen_length=32, temperature=1.0, rescore_method=None, rescore_reward=0.5,
                return_all=False, lstm_fn=lstm_step, sample_method="beam", sample_topk=3, decoder_noise=None,
                multinomial=torch.multinomial, randn=torch.randn):
    """Per-image beam search exactly as the reference runs it (one image at a time, the beam is the batch).
    ann_img (B,D,h,w).  Returns (captions, scores, alphas, perplexities) lists like model.py:472.
    sample_method "multinomial" / "topk" follow model.py:360-379 and decoder_noise model.py:322-324; their random draws
    come from ``multinomial(probs, k)`` / ``randn(shape)`` (the torch samplers unless a test supplies its own)."""
    assert sample_method in ("beam", "multinomial", "topk")
    stoi = hp.vocab_stoi
    START, PAD, END, UNK = int(stoi["<START>"]), int(stoi["<PAD>"]), int(stoi["<END>"]), int(stoi["<UNK>"])
    V, layers, n = hp.vocab_size, hp.decoder_layers, hp.decoder_dim
    temps = temperature if isinstance(temperature, list) else [temperature]
    captions, cap_scores, cap_alphas, cap_ppl = [], [], [], []
    _, _, Hh, Ww = ann_img.shape
    for idx in range(ann_img.shape[0]):
        k = beamk
        annots = ann_img[idx].expand(k, *ann_img[idx].shape)
        h, c = init_state(sd, annots, layers, n)                          # F3: odd beams start with h/c swapped
        top_preds = torch.full((1, k), START, dtype=torch.long)
        top_scores = torch.zeros(k)
        alphas = torch.zeros(1, k, Hh, Ww)
        fin_caps, fin_alphas, fin_scores, fin_ppl = [], [], [], []

        def rescore(s, step):
            if rescore_method == "LN":
                return s / step
            if rescore_method == "WR":
                return s + rescore_reward * step
            if rescore_method == "BAR":
                return s + rescore_reward * (-torch.mean(top_scores))
            return s

        step = 0
        while True:
            T = temps[step % len(temps)]
            y = embed(sd, top_preds[step])
            z, alpha = soft_attention(sd, annots, h[-1])
            x = torch.cat([y, beta_gate(sd, h[-1]) * z], dim=1).unsqueeze(0)
            if decoder_noise is not None and decoder_noise != 0.0:               # model.py:322-324: after attention / gate
                h = h + randn(h.size()) * decoder_noise / (step + 1)
            h, c = lstm_fn(sd, x, h, c, layers)
            scores = F.log_softmax(deep_output(sd, y, h[-1], z, hp.deep_output) / T, dim=1)
            scores[:, [START, PAD]] = float("-inf")                      # model.py:333
            if step == 0:
                scores[:, [END, UNK]] = float("-inf")                    # model.py:340
                top_scores, pred_idx = torch.topk(scores[0], k)
                top_preds = torch.cat([top_preds, pred_idx.unsqueeze(0)], 0)
                alphas = torch.cat([alphas, alpha.unsqueeze(0)], 0)
            else:
                seq = scores + top_scores.unsqueeze(1)
                if sample_method == "beam":
                    _, pred_idx = torch.topk(seq.reshape(-1), k, dim=0)      # model.py:359
                elif sample_method == "multinomial":                         # model.py:360-364
                    pred_idx = multinomial(F.softmax(20 * seq / step, dim=1).reshape(-1), k)
                else:                                                        # model.py:365-379
                    _, cand = torch.topk(seq, sample_topk, dim=1)
                    cand = (cand + torch.tensor([i * V for i in range(k)]).unsqueeze(1).to(cand)).reshape(-1)
                    choice = multinomial(F.softmax(seq.reshape(-1)[cand] / step, dim=0), k)
                    pred_idx = cand[choice]
                top_scores = seq.reshape(-1)[pred_idx]
                keep = torch.div(pred_idx, V, rounding_mode="floor")
                word = torch.remainder(pred_idx, V).unsqueeze(0)
                top_preds = torch.cat([top_preds[:, keep], word], 0)
                alphas = torch.cat([alphas[:, keep], alpha.unsqueeze(0)[:, keep]], 0)
                h, c = h[:, keep], c[:, keep]
                annots = annots[keep]
            complete = top_preds[step + 1] == END
            if bool(complete.any()):
                for i in torch.nonzero(complete).flatten().tolist():
                    fin_caps.append(top_preds[:, i][1:-1].tolist())
                    fin_alphas.append(alphas[:, i][1:-1].clone())
                    fin_scores.append(float(rescore(top_scores[i], step)))
                    fin_ppl.append(float(torch.exp(-top_scores[i] / step)))
                inc = ~complete
                top_preds, alphas, top_scores = top_preds[:, inc], alphas[:, inc], top_scores[inc]
                h, c, annots = h[:, inc], c[:, inc], annots[inc]
                k = int(inc.sum())
                if k == 0:
                    break
            if step >= max_gen_length:
                for i in range(top_preds.shape[1]):
                    fin_caps.append(top_preds[:, i][1:-1].tolist())
                    fin_alphas.append(alphas[:, i][1:-1].clone())
                    fin_scores.append(float(rescore(top_scores[i], step)))
                    fin_ppl.append(float(torch.exp(-top_scores[i] / step)))
                break
            step += 1
        if return_all:
            order = [i for _, i in sorted([[fin_scores[i], i] for i in range(len(fin_scores))], reverse=True)]
            captions.append([fin_caps[i] for i in order]); cap_alphas.append([fin_alphas[i] for i in order])
            cap_scores.append([fin_scores[i] for i in order]); cap_ppl.append([fin_ppl[i] for i in order])
        else:
            best = fin_scores.index(max(fin_scores))
            captions.append(fin_caps[best]); cap_alphas.append(fin_alphas[best])
            cap_scores.append(fin_scores[best]); cap_ppl.append(fin_ppl[best])
    return captions, cap_scores, cap_alphas, cap_ppl
