"""GPU parity of the encoder layers (C ABI: sat_conv2d_*, sat_bn_*, sat_maxpool*, sat_resize_*) against torch
fp32 on the CPU, and of the whole ResNet-18 get_encoder against the oracle's build_encoder with the same weights.
Encoder parity is unpinned at the reference level (torchvision is absent): the oracle is the
reference's own get_encoder logic running over the repo's ResNet definition (tests/golden/g_encoder.npz)."""
import numpy as np
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def close(a, b, tol, what=""):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, "%s: max|d|=%.3e (scale %.3g)" % (what, err, scale)


@pytest.fixture(scope="module")
def E():
    import sat_amd  # noqa: F401
    from sat_amd import encoder
    return encoder


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2)


CONVS = [  # N,H,W,C,K,R,stride,pad
    (2, 9, 9, 8, 12, 3, 1, 1), (2, 10, 11, 8, 16, 3, 2, 1), (3, 8, 8, 16, 8, 1, 1, 0), (2, 9, 9, 8, 12, 1, 2, 0),
    (2, 20, 20, 4, 8, 7, 2, 3), (1, 16, 16, 64, 64, 3, 1, 1), (4, 14, 14, 32, 128, 3, 2, 1),
]


@pytest.mark.parametrize("N,H,W,C,K,R,stride,pad", CONVS)
def test_conv_fwd_dgrad_wgrad(E, N, H, W, C, K, R, stride, pad):
    g = torch.Generator().manual_seed(N * 100 + H + C + K + R)
    x = torch.randn(N, C, H, W, generator=g, requires_grad=True)
    w = torch.randn(K, C, R, R, generator=g, requires_grad=True)
    y = F.conv2d(x, w, None, stride, pad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xd = nhwc(x.detach()).cuda(); wd = w.detach().cuda().contiguous(memory_format=torch.channels_last)
    yd = E.conv_fwd(xd, wd, stride, pad)
    tol = 1e-5 * (C * R * R) ** 0.5
    close(nchw(yd), y, tol, "conv fwd")
    dyd = nhwc(dy).cuda()
    close(nchw(E.conv_dgrad(dyd, wd, xd.shape, stride, pad)), x.grad, 1e-5 * (K * R * R) ** 0.5, "conv dgrad")
    close(E.conv_wgrad(dyd, xd, wd, stride, pad), w.grad, 1e-5 * (N * y.shape[2] * y.shape[3]) ** 0.5, "conv wgrad")
    # accumulate form
    base = torch.randn(x.shape, generator=g)
    acc = E.conv_dgrad(dyd, wd, xd.shape, stride, pad, out=nhwc(base).cuda(), accumulate=True)
    close(nchw(acc), x.grad + base, 1e-5 * (K * R * R) ** 0.5, "conv dgrad accumulate")


def test_conv_bias(E):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 16, 4, 4, generator=g); w = torch.randn(8, 16, 1, 1, generator=g); b = torch.randn(8, generator=g)
    close(nchw(E.conv_fwd(nhwc(x).cuda(), w.cuda(), 1, 0, b.cuda())), F.conv2d(x, w, b), 2e-5, "1x1 conv + bias")


@pytest.mark.parametrize("mask", [False, True])
@pytest.mark.parametrize("relu,res", [(True, False), (True, True), (False, False)])
@pytest.mark.parametrize("N,H,W,C", [(4, 5, 5, 8), (2, 16, 16, 64), (3, 7, 7, 12), (8, 8, 8, 128), (8, 16, 16, 64), (8, 4, 4, 256), (8, 2, 2, 512)])
def test_batchnorm_train_fwd_bwd(E, N, H, W, C, relu, res, mask):
    """mask=True: the forward writes the 1-bit ReLU sign mask and the backward reads it instead of the output tensor."""
    g = torch.Generator().manual_seed(C + N)
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5); bn.bias.copy_(torch.randn(C, generator=g) * 0.1)
    x = (torch.randn(N, C, H, W, generator=g) * 2 + 0.5).requires_grad_()
    r = torch.randn(N, C, H, W, generator=g).requires_grad_() if res else None
    y = bn(x)
    if res: y = y + r
    if relu: y = F.relu(y)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    bnd = torch.nn.BatchNorm2d(C).cuda()
    with torch.no_grad():
        bnd.weight.copy_(bn.weight); bnd.bias.copy_(bn.bias)
    xd = nhwc(x.detach()).cuda(); rd = nhwc(r.detach()).cuda() if res else None
    yd, stats = E.bn_fwd(xd, bnd, rd, relu, True, want_mask=mask)
    close(nchw(yd), y, 1e-5, "bn fwd")
    y_for_bwd = yd
    if mask and relu and C % 8 == 0:
        assert len(stats) == 3
        bits = np.unpackbits(stats[2].cpu().numpy(), bitorder="little").astype(bool)
        assert np.array_equal(bits, (yd.cpu().numpy().reshape(-1) > 0))
        y_for_bwd = None                      # the backward must not need the output tensor any more
    else:
        assert len(stats) == 2
    close(bnd.running_mean, bn.running_mean, 1e-5, "running_mean"); close(bnd.running_var, bn.running_var, 1e-5, "running_var")
    assert int(bnd.num_batches_tracked) == 1
    dres = torch.empty_like(xd) if res else None
    dx, dgam, dbet = E.bn_bwd(nhwc(dy).cuda(), xd, y_for_bwd, stats, bnd, relu, dres=dres)
    close(nchw(dx), x.grad, 2e-5, "bn dx"); close(dgam, bn.weight.grad, 2e-5, "bn dgamma"); close(dbet, bn.bias.grad, 2e-5, "bn dbeta")
    if res:
        close(nchw(dres), r.grad, 1e-6, "residual grad")
    # eval mode
    bn.eval(); bnd.eval()
    ye = bn(x.detach()); ye = F.relu(ye) if relu else ye
    yde, _ = E.bn_fwd(xd, bnd, None, relu, False)
    close(nchw(yde), ye, 1e-5, "bn eval")


def test_maxpool_fwd_bwd_with_ties(E):
    import sat_amd  # noqa
    from sat_amd import _lib as L
    g = torch.Generator().manual_seed(9)
    x = F.relu(torch.randn(2, 8, 11, 12, generator=g)).requires_grad_()        # post-ReLU: many exact ties at 0
    y = F.max_pool2d(x, 3, 2, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xd = nhwc(x.detach()).cuda()
    N, H, W, C = xd.shape
    P, Q = y.shape[2], y.shape[3]
    yd = torch.empty(N, P, Q, C, device="cuda"); am = torch.empty(N, P, Q, C, dtype=torch.uint8, device="cuda")
    L.check(L.lib().sat_maxpool3x3s2_fwd(L.ptr(xd), L.ptr(yd), L.ptr(am), N, H, W, C, L.stream_ptr()), "maxpool fwd")
    assert torch.equal(nchw(yd).cpu(), y.detach())
    dx = torch.empty_like(xd)
    L.check(L.lib().sat_maxpool3x3s2_bwd(L.ptr(nhwc(dy).cuda()), L.ptr(am), L.ptr(dx), N, H, W, C, L.stream_ptr()), "maxpool bwd")
    close(nchw(dx), x.grad, 1e-6, "maxpool bwd (first-max tie rule)")


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 11, 12, 16), (3, 16, 16, 64), (1, 7, 9, 8)])
def test_stem_tail_one_pass_equals_bn_relu_maxpool(E, dtype, shape):
    """bn1 -> relu -> maxpool as one pass (forward and backward): fp32 against torch autograd; both storage types against the
    three separate kernels - the pooled map, the argmax and the input gradient bit for bit (the same values are compared and
    the gathered gradient is rounded where the separate path stores it), dgamma / dbeta within summation-order noise."""
    N, H, W, Cc = shape
    g = torch.Generator().manual_seed(N * 100 + H)
    x = torch.randn(N, Cc, H, W, generator=g) * 1.5 + 0.2
    bn = torch.nn.BatchNorm2d(Cc)
    with torch.no_grad():
        bn.weight.copy_(torch.randn(Cc, generator=g)); bn.bias.copy_(torch.randn(Cc, generator=g) * 0.3)      # negative gammas too
    import copy
    bnd1, bnd2 = copy.deepcopy(bn).cuda(), copy.deepcopy(bn).cuda()
    adt = torch.bfloat16 if dtype == "bf16" else torch.float32
    xd = nhwc(x).cuda().to(adt).contiguous()
    # separate kernels
    a0, s0 = E.bn_fwd(xd, bnd1, None, True, True, want_mask=(Cc % 8 == 0))
    from sat_amd import _lib as L
    P, Q = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    p0 = torch.empty(N, P, Q, Cc, dtype=adt, device="cuda"); am = torch.empty(N, P, Q, Cc, dtype=torch.uint8, device="cuda")
    L.check(L.lib().sat_maxpool3x3s2_fwd_t(int(dtype == "bf16"), L.ptr(a0), L.ptr(p0), L.ptr(am), N, H, W, Cc, L.stream_ptr()), "maxpool")
    # one pass
    p1, s1 = E.stem_tail_fwd(xd, bnd2)
    assert torch.equal(p1, p0) and torch.equal(s1[2], am)
    assert torch.equal(s1[0], s0[0]) and torch.equal(s1[1], s0[1])
    assert torch.equal(bnd1.running_mean, bnd2.running_mean) and torch.equal(bnd1.running_var, bnd2.running_var) and int(bnd2.num_batches_tracked) == 1
    dy = torch.randn(N, P, Q, Cc, generator=g)
    dyd = dy.cuda().to(adt)
    da0 = torch.empty_like(a0)
    L.check(L.lib().sat_maxpool3x3s2_bwd_t(int(dtype == "bf16"), L.ptr(dyd), L.ptr(am), L.ptr(da0), N, H, W, Cc, L.stream_ptr()), "maxpool bwd")
    dx0, dg0, db0 = E.bn_bwd(da0, xd, a0, s0, bnd1, True)
    dx1, dg1, db1 = E.stem_tail_bwd(dyd, xd, s1, bnd2)
    close(dg1, dg0, 1e-5, "dgamma vs separate kernels"); close(db1, db0, 1e-5, "dbeta vs separate kernels")
    if dtype == "fp32":
        close(dx1, dx0, 1e-5, "dx vs separate kernels")
        xr = x.clone().requires_grad_()
        bn.train()
        y = F.max_pool2d(F.relu(bn(xr)), 3, 2, 1)
        y.backward(dy.permute(0, 3, 1, 2))
        close(nchw(p1), y, 1e-5, "pooled vs torch"); close(nchw(dx1), xr.grad, 3e-5, "dx vs torch")
        close(dg1, bn.weight.grad, 3e-5, "dgamma vs torch"); close(db1, bn.bias.grad, 3e-5, "dbeta vs torch")
    else:
        diff = (dx1.float() - dx0.float()).abs()
        assert float(diff.max()) <= 0.02 * float(dx0.float().abs().max()) + 1e-6, "dx differs beyond a bf16 rounding of the statistics: %g" % float(diff.max())


@pytest.mark.parametrize("C", [12, 10])            # 16-byte and 4-byte forms of the pooling kernel
@pytest.mark.parametrize("out", [7, 14, 5, 3, 1])
def test_resize_fwd_bwd(E, out, C):
    import sat_amd  # noqa
    from sat_amd import _lib as L
    g = torch.Generator().manual_seed(out)
    x = torch.randn(2, C, 8, 8, generator=g, requires_grad=True)
    mod = torch.nn.AdaptiveAvgPool2d((out, out)) if out < 8 else torch.nn.Upsample((out, out), mode="bilinear", align_corners=False)
    y = mod(x); dy = torch.randn(y.shape, generator=g); y.backward(dy)
    xd = nhwc(x.detach()).cuda(); yd = torch.empty(2, out, out, C, device="cuda"); dx = torch.empty_like(xd)
    L.check(L.lib().sat_resize_fwd(L.ptr(xd), L.ptr(yd), 2, 8, 8, C, out, out, L.stream_ptr()), "resize fwd")
    L.check(L.lib().sat_resize_bwd(L.ptr(nhwc(dy).cuda()), L.ptr(dx), 2, 8, 8, C, out, out, L.stream_ptr()), "resize bwd")
    close(nchw(yd), y, 1e-6, "resize fwd"); close(nchw(dx), x.grad, 1e-6, "resize bwd")


@pytest.mark.parametrize("kind,cin,planes,stride", [("basic", 16, 16, 1), ("basic", 16, 32, 2), ("bottleneck", 64, 16, 1),
                                                     ("bottleneck", 32, 16, 2), ("bottleneck", 16, 8, 1), ("basic", 128, 256, 2),
                                                     ("basic", 64, 128, 2), ("bottleneck", 256, 128, 2)])
def test_residual_block_fwd_bwd(E, kind, cin, planes, stride):
    """One BasicBlock / Bottleneck with healthy batch statistics: HIP block forward/backward against the oracle's block."""
    from oracle import sat_oracle as O
    g = torch.Generator().manual_seed(cin + planes + stride)
    torch.manual_seed(cin * 7 + planes)
    ref = O._Residual(kind, cin, planes, stride, 64)
    blk = E.Block(kind, cin, planes, stride, 64)
    with torch.no_grad():
        for p in ref.parameters():
            if p.dim() == 1:
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
    blk.load_state_dict(ref.state_dict()); E._channels_last_(blk); blk = blk.cuda().train()
    x = torch.randn(8 if cin >= 64 else 6, cin, 8 if cin >= 64 else 12, 8 if cin >= 64 else 12, generator=g).requires_grad_()
    y = ref(x.clone()); dy = torch.randn(y.shape, generator=g); y.backward(dy)
    xd = nhwc(x.detach()).cuda()
    rec = E._block_fwd(blk, xd, True)
    close(nchw(rec.out), y, 2e-5, "block out")
    grads = {}
    dx, _ = E._block_bwd(rec, nhwc(dy).cuda(), grads, True)
    close(nchw(dx), x.grad, 1e-4, "block dx")
    refp = dict(ref.named_parameters())
    for k, p in blk.named_parameters():
        close(grads[p], refp[k].grad, 2e-4, k)


def _damp(enc_like, value=0.25):
    """scale the last BatchNorm of every residual branch: the well-conditioned variant of the net (a freshly initialised ResNet
    doubles its activation scale every block; ReLU decisions then flip between fp32 and fp64 and no tight comparison is possible)"""
    with torch.no_grad():
        for mod in enc_like.modules():
            if hasattr(mod, "conv1") and hasattr(mod, "bn2"):
                (mod.bn3 if hasattr(mod, "bn3") else mod.bn2).weight.fill_(value)


@pytest.mark.parametrize("arch,es,px,D,nb", [("resnet18", None, 64, 32, 8), ("resnet18", 3, 64, 32, 8), ("resnet50", None, 128, 32, 8), ("resnet50", 7, 256, 32, 8),
                                             ("wide_resnet50_2", 14, 64, 32, 8), ("resnet34", None, 64, 32, 8),
                                             # resnext (model.py:28: same branch of get_encoder): grouped 3x3 convolutions, 32 groups
                                             ("resnext50_32x4d", 7, 256, 64, 8), ("resnext101_32x8d", None, 64, 32, 8),
                                             # BASELINE configs[2] and [3]: the encoders at their real resolution and annotation shape
                                             ("resnet101", 14, 256, 512, 4), ("wide_resnet101_2", 14, 256, 1024, 4)])
def test_whole_encoder_against_oracle(E, arch, es, px, D, nb):
    """Whole encoders, fp32 parity mode, forward + every gradient against the CPU oracle; the fp64 run of the same oracle is the
    yardstick for what fp32 itself costs.  ("resnet50", 7, 256) is BASELINE configs[1]'s encoder at its real resolution
    (32768-row stage-1 maps: the tile sizes of the real step); the last two are configs[2] / [3] (L = 196, D = 512 / 1024)."""
    from oracle import prng, sat_oracle as O
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    hp = O.default_hparams(encoder_arch=arch, encoder_dim=D, input_size=px, encoder_size=es)
    torch.manual_seed(3)
    ref = O.build_encoder(hp)                                   # CPU, train mode
    _damp(ref)
    hp2 = O.default_hparams(encoder_arch=arch, encoder_dim=D, input_size=px, encoder_size=es)
    enc = E.get_encoder(hp2)
    assert list(enc.state_dict().keys()) == list(ref.state_dict().keys())
    enc.load_state_dict(ref.state_dict())
    enc = enc.cuda().train()
    img = torch.from_numpy(prng.uniform((nb, 3, px, px), 77, 0.0, 1.0))
    import copy
    ref64 = copy.deepcopy(ref).double()                         # fp64 run of the same oracle: the yardstick for fp32 noise
    y_ref = ref(img.clone())
    dy = torch.from_numpy(prng.uniform(tuple(y_ref.shape), 78))
    y_ref.backward(dy)
    y64 = ref64(img.double().clone()); y64.backward(dy.double())
    y = enc(img.cuda())
    assert y.shape == y_ref.shape
    close(y, y_ref, 2e-4, "annotations")
    y.backward(dy.cuda())
    gref = dict(ref.named_parameters()); g64 = dict(ref64.named_parameters())
    worst = (0.0, 0.0, "")
    for k, p in enc.named_parameters():
        assert p.grad is not None, k
        exact = g64[k].grad
        nrm = max(1e-12, float(exact.norm()))
        err_gpu = float((p.grad.cpu().double() - exact).norm()) / nrm
        err_cpu = float((gref[k].grad.double() - exact).norm()) / nrm      # what fp32 itself costs on this net
        worst = max(worst, (err_gpu, err_cpu, k))
        # The HIP path must be as close to fp64 as the fp32 CPU reference is: twice its error plus `slack`.  What the slack covers: the
        # exact-fp32 MFMA sums in another order than the CPU, so a pre-activation within fp32 rounding of zero can take the other side of a
        # ReLU; ONE such flip moves a channel's gradients by ~1 / (samples per channel).  At the BASELINE resolution (256 px: >= 512 samples
        # per channel in every layer) that is below 5e-3 per tensor; the 64 px toy nets end with 2x2 / 4x4 maps at batch 8 (32 .. 128
        # samples per channel), where a single flip is 1e-2: 2e-2 there.
        slack = 5e-3 if px >= 128 else 2e-2
        assert err_gpu <= 2 * err_cpu + slack, "%s: HIP %.3e vs CPU-fp32 %.3e (relative L2 to fp64)" % (k, err_gpu, err_cpu)
    print("worst relative grad error vs fp64 (HIP, CPU fp32, tensor):", worst)
    # running statistics follow nn.BatchNorm2d
    sd, sr = enc.state_dict(), ref.state_dict()
    for k in sd:
        if "running" in k:
            close(sd[k], sr[k], 1e-4, k)
        if "num_batches" in k:
            assert int(sd[k]) == int(sr[k]), k
    # eval mode uses the running statistics
    enc.eval(); ref.eval()
    with torch.no_grad():
        close(enc(img.cuda()), ref(img.clone()), 2e-4, "eval annotations")


# ----------------------------------------------------------------------------- bf16 storage convolutions
BCONVS = [(2, 9, 9, 8, 16, 3, 1, 1), (2, 10, 11, 16, 24, 3, 2, 1), (3, 8, 8, 16, 8, 1, 1, 0), (2, 9, 9, 8, 16, 1, 2, 0),
          (2, 20, 20, 8, 16, 7, 2, 3), (1, 16, 16, 64, 64, 3, 1, 1), (4, 14, 14, 32, 128, 3, 2, 1), (8, 16, 16, 64, 256, 1, 1, 0),
          # channel counts that are multiples of 64 take the direct-to-LDS kernel (csrc/gemm_glds.hip): several k-tiles per
          # tap, stride-2 parity classes, ragged pixel counts and filter counts, split-K weight gradients on 128-wide tiles
          (2, 12, 12, 128, 64, 3, 1, 1), (2, 13, 11, 64, 128, 3, 2, 1), (3, 9, 9, 192, 64, 1, 1, 0), (2, 8, 8, 64, 72, 3, 1, 1),
          (8, 32, 32, 128, 128, 3, 1, 1), (5, 14, 14, 256, 512, 1, 2, 0), (4, 16, 16, 128, 256, 3, 1, 1),
          # >= 8192 pixels with 64 filters or 64 input channels: the 128x64 tiles (forward, data gradient, k-major weight gradient)
          (2, 64, 64, 64, 64, 3, 1, 1), (3, 61, 59, 256, 64, 1, 1, 0), (2, 64, 66, 64, 256, 1, 1, 0),
          # >= 192 tiles of 128x128 (what the C2 step mostly runs): 3x3 and 1x1, ragged row count
          (8, 64, 64, 128, 128, 3, 1, 1), (7, 63, 65, 256, 128, 1, 1, 0), (8, 64, 64, 128, 512, 1, 1, 0)]


@pytest.mark.parametrize("N,H,W,C,K,R,stride,pad", BCONVS)
def test_conv_bf16_fwd_dgrad_wgrad(E, N, H, W, C, K, R, stride, pad):
    import ctypes
    import sat_amd  # noqa
    from sat_amd import _lib as L
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
    g = torch.Generator().manual_seed(N * 100 + H + C + K + R)
    x = bf(torch.randn(N, C, H, W, generator=g)).requires_grad_()
    w = bf(torch.randn(K, C, R, R, generator=g)).requires_grad_()
    y = F.conv2d(x, w, None, stride, pad)
    dy = bf(torch.randn(y.shape, generator=g))
    y.backward(dy)
    xd = nhwc(x.detach()).cuda().to(torch.bfloat16)
    wd = w.detach().permute(0, 2, 3, 1).contiguous().cuda().to(torch.bfloat16)       # KRSC
    dyd = nhwc(dy).cuda().to(torch.bfloat16)
    P, Q = y.shape[2], y.shape[3]
    geom = L.ConvGeom(N=N, H=H, W=W, C=C, K=K, R=R, S=R, stride=stride, pad=pad)
    lib = L.lib()
    yd = torch.empty(N, P, Q, K, dtype=torch.bfloat16, device="cuda")
    L.check(lib.sat_conv2d_fwd_bf16(L.ptr(xd), L.ptr(wd), None, L.ptr(yd), ctypes.byref(geom), L.stream_ptr()), "conv fwd bf16")
    close(nchw(yd.float()), y, 1e-2, "bf16 conv fwd")                    # output rounding to bf16: 2^-8 relative
    dx = torch.empty(N, H, W, C, dtype=torch.bfloat16, device="cuda")
    L.check(lib.sat_conv2d_dgrad_bf16(L.ptr(dyd), L.ptr(wd), L.ptr(dx), ctypes.byref(geom), 0, L.stream_ptr()), "conv dgrad bf16")
    close(nchw(dx.float()), x.grad, 1e-2, "bf16 conv dgrad")
    dw = torch.empty(K, R, R, C, dtype=torch.float32, device="cuda")
    slab = torch.empty(1 << 22, device="cuda")
    L.check(lib.sat_conv2d_wgrad_bf16(L.ptr(dyd), L.ptr(xd), L.ptr(dw), ctypes.byref(geom), L.ptr(slab), slab.numel(), L.stream_ptr()), "conv wgrad bf16")
    close(dw.permute(0, 3, 1, 2), w.grad, 1e-5 * (N * P * Q) ** 0.5, "bf16 conv wgrad (fp32 out)")


@pytest.mark.parametrize("kind,cin,planes,stride", [("basic", 16, 16, 1), ("basic", 16, 32, 2), ("bottleneck", 64, 16, 1), ("bottleneck", 32, 16, 2)])
def test_residual_block_bf16_storage(E, kind, cin, planes, stride):
    """bf16 activations / filter copies through one residual block against the oracle block with bf16 rounding at the same storage
    points (oracle/bf16_emulation.py): ReLU decisions agree, so the output is within one bf16 ulp (1e-2 of its range) and the input
    gradient and every parameter gradient within 3e-2 relative L2 (what is left: fp32 accumulation order flipping a bf16 rounding)."""
    from oracle import bf16_emulation as B16, sat_oracle as O
    g = torch.Generator().manual_seed(cin + planes + stride)
    torch.manual_seed(cin * 7 + planes)
    ref = O._Residual(kind, cin, planes, stride, 64)
    blk = E.Block(kind, cin, planes, stride, 64)
    with torch.no_grad():
        for p in ref.parameters():
            if p.dim() == 1:
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
    blk.load_state_dict(ref.state_dict()); E._channels_last_(blk); blk = blk.cuda().train()
    x = B16.bf(torch.randn(6, cin, 12, 12, generator=g)).requires_grad_()
    y = B16._block(ref, x); dy = B16.bf(torch.randn(y.shape, generator=g)); y.backward(dy)
    cache = {}
    Wt = lambda p: cache.setdefault(p, E.cast_bf16(p))
    rec = E._block_fwd(blk, nhwc(x.detach()).cuda().to(torch.bfloat16), True, Wt)
    l2 = lambda a, b: float((a.detach().float().cpu() - b.detach()).norm() / b.detach().norm())
    assert l2(nchw(rec.out), y) <= 5e-3
    grads = {}
    dx, _ = E._block_bwd(rec, nhwc(dy).cuda().to(torch.bfloat16), grads, True, Wt)
    errs = {"dx": l2(nchw(dx), x.grad)}
    refp = dict(ref.named_parameters())
    for k, p in blk.named_parameters():
        errs[k] = l2(grads[p], refp[k].grad)
    print(errs)
    assert max(errs.values()) <= 3e-2, errs


@pytest.mark.parametrize("arch,es,px,D,nb,damp", [("resnet18", 3, 64, 32, 8, 0.25), ("resnet50", 7, 256, 256, 8, 0.25), ("resnext50_32x4d", 7, 256, 64, 8, 0.25)])
def test_whole_encoder_bf16_storage_against_the_rounding_oracle(E, arch, es, px, D, nb, damp):
    """The whole encoder in bf16 mode (bf16 activations and filter copies, fp32 statistics / accumulation / parameter gradients) against
    the CPU oracle that rounds to bf16 at the same storage points (oracle/bf16_emulation.py), forward and backward from IDENTICAL inputs
    (the image, and an output gradient whose values are bf16-exact).  ("resnet50", 7, 256) is BASELINE configs[1]'s encoder at its real
    resolution; the nets are the well-conditioned (residual-damped) variants.
    Forward: annotations within 3e-2 relative L2 of the emulation (measured 1e-2 / 2e-2: fp32 sums taken in another order flip bf16
    roundings, and a stack of BatchNorm layers re-amplifies every flip; one block alone agrees to 5e-3, test_residual_block_bf16_storage).
    Backward: two correct bf16 implementations differ by ~1e-1 on the gradient tensors of such a stack (the BatchNorm backward subtracts
    the mean of bf16-stored gradients: what is left is dominated by their rounding), so the criterion is the cost of the storage format
    as the emulation measures it: the HIP path's error against the fp32 oracle <= twice the emulation's error against the fp32 oracle
    + 2e-2, per tensor."""
    from oracle import bf16_emulation as B16, prng, sat_oracle as O
    import copy
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    hp = O.default_hparams(encoder_arch=arch, encoder_dim=D, input_size=px, encoder_size=es)
    torch.manual_seed(3)
    ref = O.build_encoder(hp)
    if damp is not None:
        _damp(ref, damp)
    enc = E.get_encoder(O.default_hparams(encoder_arch=arch, encoder_dim=D, input_size=px, encoder_size=es))
    enc.load_state_dict(ref.state_dict())
    enc = enc.cuda().train(); enc.precision = "bf16"
    img = torch.from_numpy(prng.uniform((nb, 3, px, px), 77, 0.0, 1.0))
    ref32 = copy.deepcopy(ref)
    y32 = ref32(img.clone())
    dy = B16.bf(torch.from_numpy(prng.uniform(tuple(y32.shape), 78)))
    y32.backward(dy)
    y_ref = B16.encoder_forward(ref, img)
    y_ref.backward(dy)
    y = enc(img.cuda())
    l2 = lambda a, b: float((a.detach().cpu().double() - b.detach().double()).norm() / max(1e-12, float(b.detach().double().norm())))   # noqa: E731
    ann_err = l2(y, y_ref)
    print("bf16 encoder vs the rounding oracle: annotations relative L2", ann_err, " (emulation vs fp32:", l2(y_ref, y32), ")")
    assert ann_err <= 3e-2
    y.backward(dy.cuda())
    gemu, g32 = dict(ref.named_parameters()), dict(ref32.named_parameters())
    rows = sorted(((l2(p.grad, g32[k].grad) - 2 * l2(gemu[k].grad, g32[k].grad), l2(p.grad, g32[k].grad), l2(gemu[k].grad, g32[k].grad), l2(p.grad, gemu[k].grad), k)
                   for k, p in enc.named_parameters()), reverse=True)
    print("bf16 encoder: (HIP vs fp32, emulation vs fp32, HIP vs emulation) worst margins", [(round(a, 4), round(b, 4), round(c, 4), k) for _, a, b, c, k in rows[:4]])
    assert rows[0][0] <= 2e-2, rows[:4]


@pytest.mark.parametrize("kind,cin,planes,nb,hw", [("bottleneck", 256, 64, 8, 32), ("bottleneck", 512, 128, 5, 17), ("basic", 64, 64, 6, 24)])
def test_bn_backward_statistics_from_the_dgrad_epilogue_equal_the_statistics_pass(E, monkeypatch, kind, cin, planes, nb, hw):
    """bf16 mode: the data-gradient launch that writes a BatchNorm's output gradient also leaves (sum g, sum g * xhat) per row tile
    (conv_dgrad(..., bn=...)), and bn_bwd(..., tiles=...) skips its statistics pass over dy and x.  Two residual blocks in a row (the second
    block's input-gradient launch - an accumulating one - produces the first block's last BatchNorm's statistics): every gradient of the fused
    path against the path with the separate statistics kernel.  Same stored values on both sides, so the only difference is the summation
    order of the statistics (per-tile fp32 partials of signed terms that largely cancel): dgamma / dbeta 1e-4 relative, dx and the filter
    gradients within bf16 rounding of the changed statistics."""
    from oracle import sat_oracle as O
    torch.manual_seed(cin + planes)
    g = torch.Generator().manual_seed(7 + cin)
    blocks = [E.Block(kind, cin, planes, 1, 64), E.Block(kind, cin, planes, 1, 64)]
    for b in blocks:
        with torch.no_grad():
            for p in b.parameters():
                if p.dim() == 1:
                    p.copy_(torch.rand(p.shape, generator=g) + 0.5)
        E._channels_last_(b); b.cuda().train()
    x = torch.randn(nb, hw, hw, cin, generator=g).cuda().to(torch.bfloat16)
    dy = torch.randn(nb, hw, hw, cin, generator=g).cuda().to(torch.bfloat16)
    cache = {}
    Wt = lambda p: cache.setdefault(p, E.cast_bf16(p))              # noqa: E731

    def run(fused):
        monkeypatch.setattr(E, "_BN_BWD_EPILOGUE", fused)
        r0 = E._block_fwd(blocks[0], x, True, Wt); r1 = E._block_fwd(blocks[1], r0.out, True, Wt)
        grads = {}
        d, tiles = E._block_bwd(r1, dy, grads, True, Wt, dout_tiles=None, prev=r0)
        assert (tiles is not None) == fused
        d, t0 = E._block_bwd(r0, d, grads, True, Wt, dout_tiles=tiles, prev=None)
        assert t0 is None
        return d.float(), {k: v.float().clone() for k, v in grads.items()}

    dx_a, ga = run(True)
    dx_b, gb = run(False)
    names = {p: "%d.%s" % (i, k) for i, b in enumerate(blocks) for k, p in b.named_parameters()}
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))           # noqa: E731
    for p in ga:
        tol = 5e-4 if p.dim() == 1 else 2e-3        # (the first BatchNorms of the chain also see dx values whose bf16 rounding flipped)
        assert rel(ga[p], gb[p]) <= tol, (names[p], rel(ga[p], gb[p]))
    assert rel(dx_a, dx_b) <= 2e-3, rel(dx_a, dx_b)


@pytest.mark.parametrize("kind,cin,planes,nb,hw", [("bottleneck", 256, 64, 8, 32), ("bottleneck", 512, 128, 5, 17), ("basic", 64, 64, 6, 24), ("basic", 128, 128, 3, 40)])
def test_identity_path_joined_inside_the_data_gradient_launch_is_bit_identical(E, monkeypatch, kind, cin, planes, nb, hw):
    """bf16 mode, identity blocks: dx = dgrad(conv1) + relu_mask(dout) formed by ``sat_conv2d_dgrad_bf16_fused`` reading dout and the sign bits
    (no masked copy of dout is written) against the two-step path (BatchNorm backward writes the masked gradient, the data gradient
    accumulates onto it).  The same values meet in the same fp32 add and one bf16 rounding, so every gradient must be equal bit for bit."""
    torch.manual_seed(cin * 3 + planes)
    g = torch.Generator().manual_seed(11 + cin)
    # a projection block (stride 2 shortcut: its BatchNorm backward masks dout on the fly) in front of an identity block
    blocks = [E.Block(kind, cin // 2, planes, 2, 64), E.Block(kind, cin, planes, 1, 64)]
    for b in blocks:
        with torch.no_grad():
            for p in b.parameters():
                if p.dim() == 1:
                    p.copy_(torch.rand(p.shape, generator=g) + 0.5)
        E._channels_last_(b); b.cuda().train()
    assert blocks[0].downsample is not None and blocks[1].downsample is None and blocks[0].cout == cin
    x = torch.randn(nb, 2 * hw, 2 * hw, cin // 2, generator=g).cuda().to(torch.bfloat16)
    dy = torch.randn(nb, hw, hw, cin, generator=g).cuda().to(torch.bfloat16)
    cache = {}
    Wt = lambda p: cache.setdefault(p, E.cast_bf16(p))              # noqa: E731

    def run(join):
        monkeypatch.setattr(E, "_DGRAD_JOIN", join)
        r0 = E._block_fwd(blocks[0], x, True, Wt); r1 = E._block_fwd(blocks[1], r0.out, True, Wt)
        grads = {}
        keep = dy.clone()
        d, tiles = E._block_bwd(r1, keep, grads, True, Wt, dout_tiles=None, prev=r0)
        assert torch.equal(keep, dy), "the incoming gradient must not be modified"
        d, _ = E._block_bwd(r0, d, grads, True, Wt, dout_tiles=tiles, prev=None)
        return d.clone(), {k: v.clone() for k, v in grads.items()}

    dx_a, ga = run(True)
    dx_b, gb = run(False)
    assert torch.equal(dx_a, dx_b)
    names = {p: "%d.%s" % (i, k) for i, b in enumerate(blocks) for k, p in b.named_parameters()}
    for p in ga:
        assert torch.equal(ga[p], gb[p]), names[p]


@pytest.mark.parametrize("kind,cin,planes,stride,nb,hw", [("bottleneck", 64, 64, 1, 6, 24), ("bottleneck", 256, 128, 2, 4, 34), ("basic", 64, 128, 2, 5, 26)])
def test_projection_shortcut_normalised_inside_the_last_batchnorm_kernel_is_bit_identical(E, monkeypatch, kind, cin, planes, stride, nb, hw):
    """bf16 mode, projection blocks: relu(bn_last(c) + bn_shortcut(cd)) with the shortcut's BatchNorm applied inside the last BatchNorm's kernel
    (``sat_bn_train_fwd_tiles_bf16_resbn``, the normalised shortcut rounded to bf16 as if stored) against the path that writes the shortcut out:
    outputs, saved statistics, running statistics and every gradient equal bit for bit."""
    import copy
    torch.manual_seed(cin + 7 * planes)
    g = torch.Generator().manual_seed(3 + cin)
    proto = E.Block(kind, cin, planes, stride, 64)
    with torch.no_grad():
        for p in proto.parameters():
            if p.dim() == 1:
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
    assert proto.downsample is not None
    x = torch.randn(nb, hw, hw, cin, generator=g).cuda().to(torch.bfloat16)

    def run(fused):
        monkeypatch.setattr(E, "_FWD_RES_BN", fused)
        blk = copy.deepcopy(proto); E._channels_last_(blk); blk.cuda().train()
        cache = {}
        Wt = lambda p: cache.setdefault(p, E.cast_bf16(p))              # noqa: E731
        r = E._block_fwd(blk, x, True, Wt)
        dy = torch.randn(r.out.shape, generator=torch.Generator().manual_seed(5)).cuda().to(torch.bfloat16)
        grads = {}
        d, _ = E._block_bwd(r, dy, grads, True, Wt)
        names = {p: k for k, p in blk.named_parameters()}
        bufs = {k: v.clone() for k, v in blk.named_buffers()}
        return r.out.clone(), d.clone(), {names[p]: v.clone() for p, v in grads.items()}, bufs

    out_a, d_a, g_a, b_a = run(True)
    out_b, d_b, g_b, b_b = run(False)
    assert torch.equal(out_a, out_b) and torch.equal(d_a, d_b)
    for k in g_b:
        assert torch.equal(g_a[k], g_b[k]), k
    for k in b_b:
        assert torch.equal(b_a[k], b_b[k]), k


@pytest.mark.parametrize("N,H,W,C,K,R,stride,pad", [(4, 16, 16, 64, 64, 3, 1, 1), (3, 9, 9, 64, 256, 1, 1, 0), (2, 20, 20, 8, 64, 7, 2, 3),
                                                    (5, 13, 11, 128, 72, 3, 2, 1), (8, 32, 32, 64, 128, 3, 1, 1),
                                                    (2, 64, 64, 64, 64, 3, 1, 1), (3, 61, 59, 256, 64, 1, 1, 0),       # >= 8192 rows x 64 filters: 128x64 tiles
                                                    (8, 64, 64, 128, 128, 3, 1, 1), (7, 63, 65, 64, 256, 1, 1, 0)])    # >= 192 tiles of 128x128
def test_conv_epilogue_statistics_feed_batchnorm(E, N, H, W, C, K, R, stride, pad):
    """bf16 conv whose epilogue leaves per-row-tile (sum, sum of squares) of its stored output; BatchNorm built on those tiles
    must equal BatchNorm with its own statistics pass over the same tensor (ragged last tile, 64- and 128-row tiles, both kernels)."""
    g = torch.Generator().manual_seed(N + H + C + K)
    x = (torch.randn(N, H, W, C, generator=g) + 0.3).cuda().to(torch.bfloat16)
    w = (torch.randn(K, C, R, R, generator=g) / (C * R * R) ** 0.5).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y, tiles = E.conv_fwd_stats(x, w, stride, pad)
    y_ref = E.conv_fwd(x, w, stride, pad)
    assert torch.equal(y, y_ref)
    assert tiles is not None and tiles[1] in (64, 128)
    rows = y.numel() // K
    nt = -(-rows // tiles[1])
    ts = tiles[0][:nt * K * 2].reshape(nt, K, 2).double().cpu()
    yf = y.reshape(rows, K).double().cpu()
    assert float((ts[:, :, 0].sum(0) - yf.sum(0)).abs().max()) <= 1e-4 * max(1.0, float(yf.sum(0).abs().max()))
    assert float((ts[:, :, 1].sum(0) - (yf * yf).sum(0)).abs().max()) <= 1e-4 * float((yf * yf).sum(0).abs().max())
    bn_a, bn_b = torch.nn.BatchNorm2d(K).cuda(), torch.nn.BatchNorm2d(K).cuda()
    with torch.no_grad():
        bn_a.weight.copy_(torch.rand(K, generator=g) + 0.5); bn_a.bias.copy_(torch.randn(K, generator=g) * 0.1)
        bn_b.weight.copy_(bn_a.weight); bn_b.bias.copy_(bn_a.bias)
    out_a, st_a = E.bn_fwd(y, bn_a, None, True, True, tiles=tiles)
    out_b, st_b = E.bn_fwd(y, bn_b, None, True, True)
    close(st_a[0], st_b[0], 1e-5, "mean from tiles"); close(st_a[1], st_b[1], 1e-4, "invstd from tiles")
    close(out_a.float(), out_b.float(), 1e-2, "bn output from tiles")
    close(bn_a.running_var, bn_b.running_var, 1e-5, "running_var from tiles")


def test_bf16_stem_as_tap_pairs_equals_conv2d(E):
    """bf16 stem in the tap-pair layout (zero-padded 4-channel image viewed as pixel pairs, 7x4 filter, strides (2, 1)): the helper
    kernels are exact re-layouts, and forward / filter gradient equal F.conv2d(7x7, stride 2, pad 3) on the same bf16-rounded
    operands up to fp32 accumulation order - and equal the 8-channel layout it replaces."""
    from sat_amd import _lib as L
    import ctypes as C
    lib = L.lib()
    g = torch.Generator().manual_seed(31)
    N, H, W, K = 3, 20, 26, 64
    img = torch.rand(N, 3, H, W, generator=g)
    w = torch.randn(K, 3, 7, 7, generator=g) * 0.1
    mean = (C.c_float * 3)(0.485, 0.456, 0.406); std = (C.c_float * 3)(0.229, 0.224, 0.225)
    st = L.stream_ptr()
    imgd = img.cuda()
    x0 = torch.empty(N, H + 6, (W + 6) // 2, 8, dtype=torch.bfloat16, device="cuda")
    L.check(lib.sat_image_normalize_nhwc4_padded_bf16(L.ptr(imgd), L.ptr(x0), N, H, W, mean, std, st), "normalize padded")
    xn = ((img - torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)) / torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))
    xpad = x0.view(N, H + 6, W + 6, 4).float().cpu()
    assert float(xpad[:, :3].abs().max()) == 0 and float(xpad[:, -3:].abs().max()) == 0 and float(xpad[:, :, :3].abs().max()) == 0
    assert float(xpad[:, :, -3:].abs().max()) == 0 and float(xpad[..., 3].abs().max()) == 0
    xb = xn.bfloat16().float()                                         # what the kernel stores
    assert float((xpad[:, 3:-3, 3:-3, :3].permute(0, 3, 1, 2) - xb).abs().max()) <= 2 ** -7 * float(xb.abs().max())       # division vs multiply: 1 bf16 ulp
    w3 = w.permute(0, 2, 3, 1).contiguous().cuda()                     # (K, 7, 7, 3) memory
    wp = torch.empty(K, 8, 7, 4, dtype=torch.bfloat16, device="cuda").contiguous(memory_format=torch.channels_last)
    L.check(lib.sat_stem_filter_pairs(L.ptr(w3), L.ptr(wp), K, st), "filter pairs")
    wpm = wp.permute(0, 2, 3, 1).float().cpu()                         # (K, 7, 4, 8)
    wb = w.bfloat16().float()
    for s_ in range(7):
        assert torch.equal(wpm[:, :, s_ // 2, (s_ % 2) * 4:(s_ % 2) * 4 + 3], wb[:, :, :, s_].permute(0, 2, 1))
    assert float(wpm[:, :, 3, 4:].abs().max()) == 0 and float(wpm[..., 3].abs().max()) == 0 and float(wpm[..., 7].abs().max()) == 0
    y = E.conv_fwd(x0, wp, 2, 0, stride_w=1)
    xin = xpad[:, 3:-3, 3:-3, :3].permute(0, 3, 1, 2).contiguous()      # the kernel's own bf16 input values
    ref = F.conv2d(xin, wb, None, 2, 3)
    assert y.shape == (N, H // 2, W // 2, K)
    close(nchw(y.float()), ref, 1e-2, "stem forward (pairs)")
    dy = torch.randn(N, H // 2, W // 2, K, generator=g).bfloat16().cuda()
    dwp = E.conv_wgrad(dy, x0, wp, 2, 0, stride_w=1)
    dw3 = torch.empty(K, 7, 7, 3, device="cuda")
    L.check(lib.sat_stem_filter_grad_unpairs(L.ptr(dwp), L.ptr(dw3), K, st), "grad unpairs")
    wr = wb.clone().requires_grad_()
    F.conv2d(xin, wr, None, 2, 3).backward(dy.float().cpu().permute(0, 3, 1, 2))
    close(dw3.permute(0, 3, 1, 2), wr.grad, 2e-3, "stem filter gradient (pairs)")
    # the 8-channel layout it replaces
    x8 = torch.empty(N, H, W, 8, dtype=torch.bfloat16, device="cuda")
    L.check(lib.sat_image_normalize_nhwc8_bf16(L.ptr(imgd), L.ptr(x8), N, H, W, mean, std, st), "normalize nhwc8")
    w8 = torch.empty(K, 8, 7, 7, dtype=torch.bfloat16, device="cuda").contiguous(memory_format=torch.channels_last)
    L.check(lib.sat_stem_filter_pad(L.ptr(w3), L.ptr(w8), K * 49, st), "filter pad")
    y8 = E.conv_fwd(x8, w8, 2, 3)
    close(y.float(), y8.float(), 1e-2, "pairs vs 8-channel layout")


def test_nan_is_not_swallowed_by_relu_or_max_pool(E):
    """torch's ReLU and max pool pass a NaN on; so do the kernels (fmaxf(NaN, 0) = 0 would hide a broken layer from the loss)."""
    from sat_amd import _lib as L
    x = torch.randn(2, 6, 6, 8)
    x[0, 2, 3, 1] = float("nan")
    bn = torch.nn.BatchNorm2d(8).cuda().eval()
    y, _ = E.bn_fwd(x.cuda(), bn, None, True, False)
    ref = F.relu(bn.cpu()(x.permute(0, 3, 1, 2))).permute(0, 2, 3, 1)
    assert torch.equal(torch.isnan(y.cpu()), torch.isnan(ref)) and int(torch.isnan(ref).sum()) == 1
    xp = x.cuda()
    P = 3
    yd = torch.empty(2, P, P, 8, device="cuda"); am = torch.empty(2, P, P, 8, dtype=torch.uint8, device="cuda")
    L.check(L.lib().sat_maxpool3x3s2_fwd(L.ptr(xp), L.ptr(yd), L.ptr(am), 2, 6, 6, 8, L.stream_ptr()), "maxpool")
    refp = F.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    assert torch.equal(torch.isnan(yd.cpu()), torch.isnan(refp)) and int(torch.isnan(refp).sum()) >= 1


@pytest.mark.parametrize("arch,px,nb,es", [("resnet18", 64, 6, 3), ("resnet50", 128, 4, 7), ("resnet34", 96, 3, None), ("wide_resnet50_2", 64, 5, 14)])
def test_block_loop_inside_the_library_is_bit_identical_to_the_per_layer_driver(E, arch, px, nb, es):
    """csrc/encoder_loop.hip runs the trunk's residual blocks in one call per direction over one arena; it calls the same per-layer entry points
    in the same order as encoder.py's `_block_fwd` / `_block_bwd`, so annotations, every gradient and every BatchNorm buffer must be BIT-equal
    (bf16 mode, training; also with the weight gradients on the side stream)."""
    from oracle import prng, sat_oracle as O
    outs = []
    for loop in (True, False):
        hp = O.default_hparams(encoder_arch=arch, encoder_dim=48, input_size=px, encoder_size=es)
        torch.manual_seed(9)
        enc = E.get_encoder(hp).cuda().train()
        enc.precision = "bf16"
        old, old_side = E._BLOCK_LOOP, E._WGRAD_SIDE_MIN_INPUT_PIXELS
        E._BLOCK_LOOP, E._WGRAD_SIDE_MIN_INPUT_PIXELS = loop, 0          # side stream on whatever the batch size
        try:
            res = []
            for it in range(2):                                       # second pass: BatchNorm buffers have moved, bf16 filter copies are reused
                img = torch.from_numpy(prng.uniform((nb, 3, px, px), 40 + it, 0.0, 1.0)).cuda()
                for p in enc.parameters():
                    p.grad = None
                y = enc(img)
                dy = torch.from_numpy(prng.uniform(tuple(y.shape), 50 + it)).cuda()
                y.backward(dy)
                torch.cuda.synchronize()
                res.append((y.detach().clone(), {k: p.grad.clone() for k, p in enc.named_parameters()}, {k: v.clone() for k, v in enc.state_dict().items()}))
        finally:
            E._BLOCK_LOOP, E._WGRAD_SIDE_MIN_INPUT_PIXELS = old, old_side
        outs.append(res)
    for (ya, ga, sa), (yb, gb, sb) in zip(*outs):
        assert torch.equal(ya, yb), "annotations differ"
        for k in ga:
            assert torch.equal(ga[k], gb[k]), "gradient of %s differs (max |d| %.3e)" % (k, float((ga[k] - gb[k]).abs().max()))
        for k in sa:
            assert torch.equal(sa[k], sb[k]), "buffer %s differs" % k
