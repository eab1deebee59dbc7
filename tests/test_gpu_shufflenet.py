"""GPU parity of the ShuffleNetV2 encoders (the reference's CLI default --encoder_arch shufflenet_v2_x0_5, train.py:43; model.py:30-31): the
depthwise 3x3 / channel-shuffle kernels (C ABI: sat_dwconv3x3_*, sat_shuffle_*) against torch on the CPU, and the whole get_encoder against
the oracle's build_encoder with the same weights.  Like the ResNets, torchvision's ShuffleNetV2 is third-party arithmetic that is absent from
the reference tree: parity is unpinned at the reference level and pinned structurally (tests/test_oracle_golden.py: parameter counts and
feature dims of dev/encoder_summaries.txt:28-35)."""
import copy
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
def S():
    import sat_amd  # noqa: F401
    from sat_amd import encoder_shuffle
    return encoder_shuffle


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("N,H,W,C,stride", [(2, 9, 9, 8, 1), (2, 10, 11, 24, 2), (3, 7, 5, 48, 1), (2, 8, 8, 96, 2), (1, 1, 1, 8, 1), (2, 2, 3, 16, 2),
                                             (4, 28, 28, 24, 2), (2, 14, 14, 352, 1), (16, 56, 56, 24, 2)])
def test_depthwise3x3_fwd_dgrad_wgrad(S, N, H, W, C, stride, dtype):
    from oracle import prng
    bf = dtype == "bf16"
    rnd = (lambda t: t.to(torch.bfloat16).float()) if bf else (lambda t: t)
    x = rnd(torch.from_numpy(prng.uniform((N, C, H, W), 5)))
    conv = torch.nn.Conv2d(C, C, 3, stride, 1, bias=False, groups=C)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(prng.uniform((C, 1, 3, 3), 6)))
    xr = x.clone().requires_grad_(True)
    y_ref = conv(xr)
    dy = rnd(torch.from_numpy(prng.uniform(tuple(y_ref.shape), 7)))
    y_ref.backward(dy)
    adt = torch.bfloat16 if bf else torch.float32
    cg = copy.deepcopy(conv).cuda()
    xg = nhwc(x).cuda().to(adt)
    y = S.dw_fwd(xg, cg)
    assert y.dtype == adt
    tol = 1e-2 if bf else 1e-5            # bf16: the output is stored rounded (2^-9 relative), inputs are bf16-exact
    close(nchw(y.float()), y_ref, tol, "depthwise forward")
    dyg = nhwc(dy).cuda().to(adt)
    dx = S.dw_dgrad(dyg, cg, tuple(xg.shape))
    close(nchw(dx.float()), xr.grad, tol, "depthwise data gradient")
    dw = S.dw_wgrad(dyg, xg, cg)
    assert dw.dtype == torch.float32 and tuple(dw.shape) == (C, 1, 3, 3)
    close(dw, conv.weight.grad, 2e-5, "depthwise filter gradient")          # fp32 sums of exact products in both modes
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("rows,Ch", [(7, 8), (49 * 3, 24), (1, 96), (1000, 88)])
def test_channel_shuffle_join_and_split(S, rows, Ch, dtype):
    from oracle import prng, sat_oracle as O
    adt = torch.bfloat16 if dtype == "bf16" else torch.float32
    a = torch.from_numpy(prng.uniform((1, rows, 1, Ch), 1)).to(adt); b = torch.from_numpy(prng.uniform((1, rows, 1, Ch), 2)).to(adt)
    ref = nhwc(O.channel_shuffle(torch.cat((nchw(a.float()), nchw(b.float())), 1), 2))       # (1, rows, 1, 2 Ch)
    full = S.shuffle_join(a.cuda(), b.cuda(), False)
    assert torch.equal(full.float().cpu(), ref)
    x1, x2 = S.shuffle_join(a.cuda(), b.cuda(), True)
    assert torch.equal(x1.float().cpu(), ref[..., :Ch]) and torch.equal(x2.float().cpu(), ref[..., Ch:])
    # the split is the inverse permutation, from the whole tensor or from its halves
    da, db = S.shuffle_split(full)
    assert torch.equal(da.cpu(), a) and torch.equal(db.cpu(), b)
    da, db = S.shuffle_split((x1, x2))
    assert torch.equal(da.cpu(), a) and torch.equal(db.cpu(), b)


def _unpad(g, like):
    """x1_0 / x2_0 hold their tensors zero-padded to multiples of 8 channels: the leading corner is the reference-shaped tensor, the rest must be 0"""
    if tuple(g.shape) == tuple(like.shape):
        return g
    idx = tuple(slice(0, n) for n in like.shape)
    rest = g.clone(); rest[idx] = 0
    assert float(rest.abs().max()) == 0.0, "a padded channel carries a non-zero value"
    return g[idx]


def _zero_gradient_bias(key):
    """The BatchNorm behind a depthwise convolution feeds a 1x1 convolution and another train-mode BatchNorm with no ReLU in between: a shift of
    one of its channels is a per-channel constant after the 1x1 and the next BatchNorm subtracts it - the gradient of its bias is exactly
    zero.  What any implementation computes there is rounding noise: it is measured against the scale of the same layer's weight gradient."""
    return key.endswith("branch1.1.bias") or key.endswith("branch2.4.bias")


@pytest.mark.parametrize("arch,es,px,D,nb", [("shufflenet_v2_x0_5", None, 224, None, 8),          # the reference's defaults: 224 px, no projection (train.py:43-50)
                                             ("shufflenet_v2_x0_5", 3, 64, 32, 8), ("shufflenet_v2_x1_5", None, 128, 64, 4),
                                             ("shufflenet_v2_x0_5", 14, 256, 512, 4),
                                             # 58- / 122-channel branches, held zero-padded at 64 / 128 (encoder_shuffle.py): same acceptance
                                             ("shufflenet_v2_x1_0", None, 224, None, 8), ("shufflenet_v2_x1_0", 5, 128, 48, 4), ("shufflenet_v2_x2_0", None, 128, 64, 4)])
def test_whole_shufflenet_encoder_against_oracle(S, arch, es, px, D, nb):
    """fp32 parity mode, forward + every gradient + running statistics + eval mode against the CPU oracle; acceptance as for the ResNets
    (tests/test_gpu_encoder.py::test_whole_encoder_against_oracle): as close to the fp64 run of the oracle as the fp32 CPU run is."""
    from oracle import prng, sat_oracle as O
    from sat_amd import encoder as E
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    hp = O.default_hparams(encoder_arch=arch, encoder_dim=D, input_size=px, encoder_size=es)
    torch.manual_seed(3)
    ref = O.build_encoder(hp)                                   # CPU, train mode
    hp2 = O.default_hparams(encoder_arch=arch, encoder_dim=D, input_size=px, encoder_size=es)
    torch.manual_seed(3)
    enc = E.get_encoder(hp2)
    assert hp2.encoder_dim == hp.encoder_dim
    assert list(enc.state_dict().keys()) == list(ref.state_dict().keys())
    for k, v in enc.state_dict().items():                       # same constructor order and initialisers: the same seed gives the same network
        assert torch.equal(v, ref.state_dict()[k]), k
    enc = enc.cuda().train()
    img = torch.from_numpy(prng.uniform((nb, 3, px, px), 77, 0.0, 1.0))
    ref64 = copy.deepcopy(ref).double()
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
        if _zero_gradient_bias(k):
            nrm = max(1e-12, float(g64[k[:-4] + "weight"].grad.norm()))
        err_gpu = float((_unpad(p.grad, exact).cpu().double() - exact).norm()) / nrm
        err_cpu = float((gref[k].grad.double() - exact).norm()) / nrm
        worst = max(worst, (err_gpu, err_cpu, k))
        slack = 5e-3 if px >= 128 else 2e-2          # one ReLU decision within fp32 rounding of zero taken the other way (see the ResNet test)
        assert err_gpu <= 2 * err_cpu + slack, "%s: HIP %.3e vs CPU-fp32 %.3e (relative L2 to fp64)" % (k, err_gpu, err_cpu)
    print("worst relative grad error vs fp64 (HIP, CPU fp32, tensor):", worst)
    sd, sr = enc.state_dict(), ref.state_dict()
    for k in sd:
        if "running" in k:
            close(sd[k], sr[k], 1e-4, k)
        if "num_batches" in k:
            assert int(sd[k]) == int(sr[k]), k
    enc.eval(); ref.eval()
    with torch.no_grad():
        close(enc(img.cuda()), ref(img.clone()), 2e-4, "eval annotations")


@pytest.mark.parametrize("arch,es,px,D,nb", [("shufflenet_v2_x0_5", None, 224, None, 8), ("shufflenet_v2_x0_5", 7, 256, 256, 8), ("shufflenet_v2_x1_5", None, 128, 64, 8),
                                             ("shufflenet_v2_x1_0", None, 224, None, 8)])
def test_whole_shufflenet_encoder_bf16_storage_against_the_rounding_oracle(S, arch, es, px, D, nb):
    """bf16 mode against the CPU oracle that rounds to bf16 at the same storage points (oracle/bf16_emulation.py).  A freshly initialised
    ShuffleNetV2 amplifies any perturbation by ~1.2x per unit (16 units: a 1e-6 relative change of the image moves the annotations by 2e-5;
    the emulation itself sits 0.34 relative L2 away from the fp32 oracle on these inputs), so two correct bf16 implementations whose fp32
    sums run in different orders cannot agree tightly on the whole net: single units do (test_shuffle_unit_bf16_storage, 1e-2).  Whole net:
    the HIP path is closer to the emulation than half the emulation's own distance from fp32 (+3e-2, the ResNet bound); gradients: the HIP
    path's error against the fp32 oracle <= twice the emulation's + 2e-2 per tensor, as for the ResNets
    (tests/test_gpu_encoder.py::test_whole_encoder_bf16_storage_against_the_rounding_oracle)."""
    from oracle import bf16_emulation as B16, prng, sat_oracle as O
    from sat_amd import encoder as E
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    hp = O.default_hparams(encoder_arch=arch, encoder_dim=D, input_size=px, encoder_size=es)
    torch.manual_seed(3)
    ref = O.build_encoder(hp)
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
    assert y.dtype == torch.float32
    l2 = lambda a, b: float((a.detach().cpu().double() - b.detach().double()).norm() / max(1e-12, float(b.detach().double().norm())))   # noqa: E731
    ann_err = l2(y, y_ref)
    emu_cost = l2(y_ref, y32)
    print("bf16 shufflenet vs the rounding oracle: annotations relative L2", ann_err, " (emulation vs fp32:", emu_cost, ")")
    assert ann_err <= 0.5 * emu_cost + 3e-2
    y.backward(dy.cuda())
    gemu, g32 = dict(ref.named_parameters()), dict(ref32.named_parameters())
    for k, p in enc.named_parameters():          # exactly-zero gradients: rounding noise on both sides, bounded by the emulation's
        if _zero_gradient_bias(k):
            assert float(p.grad.norm()) <= 4 * float(gemu[k].grad.norm()) + 1e-3 * float(gemu[k[:-4] + "weight"].grad.norm()), k
    hip = {k: _unpad(p.grad, g32[k].grad) for k, p in enc.named_parameters()}
    rows = sorted(((l2(hip[k], g32[k].grad) - 2 * l2(gemu[k].grad, g32[k].grad), l2(hip[k], g32[k].grad), l2(gemu[k].grad, g32[k].grad), l2(hip[k], gemu[k].grad), k)
                   for k in hip if not _zero_gradient_bias(k)), reverse=True)
    print("bf16 shufflenet: (HIP vs fp32, emulation vs fp32, HIP vs emulation) worst margins", [(round(a, 4), round(b, 4), round(c, 4), k) for _, a, b, c, k in rows[:4]])
    assert rows[0][0] <= 2e-2, rows[:4]
    sd, sr = enc.state_dict(), ref.state_dict()
    for k in sd:
        if "running" in k:
            close(sd[k], sr[k], 2e-2, k)
        if "num_batches" in k:
            assert int(sd[k]) == int(sr[k]), k


@pytest.mark.parametrize("inp,oup,stride,nb,hw", [(24, 48, 2, 8, 28), (48, 48, 1, 8, 14), (96, 192, 2, 6, 15), (192, 192, 1, 8, 7), (352, 352, 1, 4, 9)])
def test_shuffle_unit_bf16_storage(S, inp, oup, stride, nb, hw):
    """One unit in bf16 storage from identical bf16-exact inputs against the rounding emulation: output and input gradient within 1e-2
    relative L2, parameter gradients within 3e-2 (the bound of tests/test_gpu_encoder.py::test_residual_block_bf16_storage)."""
    from oracle import bf16_emulation as B16, prng, sat_oracle as O
    from sat_amd import encoder as E
    torch.manual_seed(inp + stride)
    ref = O._ShuffleUnit(inp, oup, stride).train()
    with torch.no_grad():
        for mod in ref.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5); mod.bias.uniform_(-0.3, 0.3)
    u = S.ShuffleUnit(inp, oup, stride)
    u.load_state_dict(ref.state_dict())
    u = u.cuda().train()
    x = B16.bf(torch.from_numpy(prng.uniform((nb, inp, hw, hw), 21))).requires_grad_(True)
    if stride == 1:
        x1, x2 = x.chunk(2, dim=1)
        out = torch.cat((x1, B16._shuffle_branch(x2, ref.branch2)), 1)
    else:
        out = torch.cat((B16._shuffle_branch(x, ref.branch1), B16._shuffle_branch(x, ref.branch2)), 1)
    y_ref = O.channel_shuffle(out, 2)
    dy = B16.bf(torch.from_numpy(prng.uniform(tuple(y_ref.shape), 22)))
    y_ref.backward(dy)
    Wt = E._weight_reader(True)
    xg = nhwc(x.detach()).cuda().to(torch.bfloat16)
    xin = xg if stride > 1 else tuple(t.contiguous() for t in xg.chunk(2, dim=3))
    r, y = S._unit_fwd(u, xin, True, Wt, False)
    l2 = lambda a, b: float((a.detach().cpu().double() - b.detach().double()).norm() / max(1e-12, float(b.detach().double().norm())))   # noqa: E731
    errs = {"out": l2(nchw(y.float()), y_ref)}
    grads = {}
    dx = S._unit_bwd(r, nhwc(dy).cuda().to(torch.bfloat16), grads, Wt)
    dx = dx if stride > 1 else torch.cat(dx, 3)
    errs["dx"] = l2(nchw(dx.float()), x.grad)
    assert errs["out"] <= 1e-2 and errs["dx"] <= 1e-2, errs
    gref = dict(ref.named_parameters())
    for k, p in u.named_parameters():
        g = grads[p]
        if _zero_gradient_bias("x." + k):
            assert float(g.norm()) <= 4 * float(gref[k].grad.norm()) + 1e-3 * float(gref[k[:-4] + "weight"].grad.norm()), k
            continue
        errs[k] = l2(g.reshape(gref[k].shape), gref[k].grad)
    print(errs)
    assert max(errs.values()) <= 3e-2, errs


def test_frozen_trunk_trains_only_the_projection(S):
    """encoder_finetune_after (train.py:48): the trunk's parameters do not require gradients until the unfreeze step; the 1x1 projection does"""
    from oracle import prng, sat_oracle as O
    from sat_amd import encoder as E
    hp = O.default_hparams(encoder_arch="shufflenet_v2_x0_5", encoder_dim=32, input_size=64)
    torch.manual_seed(1)
    ref = O.build_encoder(hp)
    enc = E.get_encoder(O.default_hparams(encoder_arch="shufflenet_v2_x0_5", encoder_dim=32, input_size=64))
    enc.load_state_dict(ref.state_dict())
    enc = enc.cuda().train()
    for k, p in list(enc.named_parameters()) + list(ref.named_parameters()):
        p.requires_grad = k.startswith("7.")
    img = torch.from_numpy(prng.uniform((4, 3, 64, 64), 9, 0.0, 1.0))
    y_ref = ref(img.clone()); y = enc(img.cuda())
    dy = torch.from_numpy(prng.uniform(tuple(y_ref.shape), 10))
    y_ref.backward(dy); y.backward(dy.cuda())
    gr = dict(ref.named_parameters())
    for k, p in enc.named_parameters():
        if k.startswith("7."):
            close(p.grad, gr[k].grad, 2e-4, k)
        else:
            assert p.grad is None, k


def test_padded_widths_stay_exactly_zero_through_training():
    """shufflenet_v2_x1_0 (58 / 116 / 232-channel branches held at 64 / 120 / 232): after optimizer steps with weight decay and gradient clipping
    every padded entry of every parameter, gradient and Adam moment is still exactly 0, the state dict has the reference's shapes and equals
    the leading corner of the held tensors, and a model rebuilt from that state dict computes the same loss bit for bit."""
    model, _, hp = _make_model(dict(encoder_arch="shufflenet_v2_x1_0", weight_decay=1e-2, opt="adamw", encoder_lr=1e-3, clip_value=1.0, grad_clip="norm",
                                    encoder_finetune_after=1))          # > 0: the encoder's parameters are in the optimizer (model.py:777, F10)
    model.set_precision("bf16")
    opt = model.configure_optimizers()
    opt = opt[0][0] if isinstance(opt, tuple) else opt
    img, caps, lengths = _batch(hp, B=4)
    b = (img.cuda(), caps.cuda(), lengths)
    for it in range(4):
        opt.zero_grad(set_to_none=True)
        model.training_step(b, it)["loss"].backward()
        opt.step()
    padded = 0
    for mod in model.encoder.modules():
        for name, shape in mod.__dict__.get("_sat_real_shapes", {}).items():
            t = getattr(mod, name)
            padded += 1
            for what, full in (("value", t), ("gradient", getattr(t, "grad", None)), ("exp_avg", opt.state.get(t, {}).get("exp_avg")), ("exp_avg_sq", opt.state.get(t, {}).get("exp_avg_sq"))):
                if full is None:
                    continue
                rest = full.clone(); rest[tuple(slice(0, n) for n in shape)] = 0
                assert float(rest.abs().max()) == 0.0, (name, what)
    assert padded > 100
    assert any(p is q for g in opt.param_groups for p in g["params"] for q in model.encoder.parameters())
    sd = model.state_dict()
    assert tuple(sd["encoder.3.0.branch1.2.weight"].shape) == (58, 24, 1, 1) and tuple(sd["encoder.4.1.branch2.3.weight"].shape) == (116, 1, 3, 3)
    twin, _, _ = _make_model(dict(encoder_arch="shufflenet_v2_x1_0"), seed=7)
    twin.load_state_dict(sd)
    twin.set_precision("bf16")
    model.eval(); twin.eval()
    with torch.no_grad():
        l1, l2 = model.training_step(b, 0)["loss"], twin.training_step(b, 0)["loss"]
    assert torch.equal(l1, l2)


# ----------------------------------------------------------------------------- the whole train step behind the reference's SAT surface
def _make_model(over=None, seed=42):
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    from oracle import sat_oracle as O
    kw = dict(encoder_arch="shufflenet_v2_x0_5", encoder_dim=None, input_size=64, encoder_size=None, vocab_size=120, embed_dim=24,
              attention_dim=16, decoder_dim=40, deep_output=True, weight_decay=0.0, decoder_lr=1e-3, embedding_lr=1e-2,
              encoder_lr=1e-4, opt="adam", adam_b1=0.9, adam_b2=0.999, momentum=0.9, nesterov=False, scheduler=None)
    kw.update(over or {})
    hp = O.default_hparams(**kw)
    torch.manual_seed(seed)
    model = M.SAT(**vars(hp))
    oracle = O.OracleSAT(O.default_hparams(**kw), {k: v.clone() for k, v in model.state_dict().items()})
    return model.cuda().train(), oracle, hp


def _batch(hp, B=6, R=3, T=9, seed=5):
    from oracle import prng
    img = torch.from_numpy(prng.uniform((B, 3, hp.input_size, hp.input_size), seed, 0.0, 1.0))
    caps, lengths = prng.captions(B, R, T, hp.vocab_size, seed + 1)
    return img, torch.from_numpy(caps), torch.from_numpy(lengths)


@pytest.mark.parametrize("eps,D,arch", [(1.0, None, "shufflenet_v2_x0_5"), (0.0, 32, "shufflenet_v2_x0_5"), (1.0, 32, "shufflenet_v2_x1_0")])
def test_training_step_with_the_cli_default_encoder_matches_oracle(eps, D, arch):
    """SAT(encoder_arch="shufflenet_v2_x0_5") - the reference's defaults (train.py:43, :50: no projection, encoder_dim = 1024) and the
    projected variant - one training_step against the CPU oracle: loss, accuracy, packed logits, attention maps, every gradient."""
    model, oracle, hp = _make_model(dict(encoder_arch=arch, encoder_dim=D, decoder_tf="always" if eps == 1.0 else None))
    assert model.hp.encoder_dim == (1024 if D is None else D)
    img, caps, lengths = _batch(hp)
    loss_o, out_o = oracle.step_loss(img, caps, lengths, eps)
    loss_o.backward()
    img_g = img.cuda()
    metrics = model.training_step((img_g, caps.cuda(), lengths), 0)
    assert torch.equal(img_g.cpu(), img)
    assert abs(metrics["loss"].item() - loss_o.item()) <= 1e-4 * max(1.0, abs(loss_o.item()))
    assert abs(float(metrics["accuracy"]) - float(out_o["acc"])) < 1e-6
    lp, tp, alphas = model.train_batch((img_g, caps.cuda(), lengths), eps)
    rel = lambda a, b: float((a.detach().cpu().double() - b.detach().double()).abs().max()) / max(1.0, float(b.detach().double().abs().max()))   # noqa: E731
    assert rel(lp.data, out_o["logits_packed"]) <= 2e-4 and rel(alphas, out_o["alphas"]) <= 1e-4
    metrics["loss"].backward()
    og = oracle.named_grads()
    worst = (0.0, "")
    for k, p in model.named_parameters():
        assert p.grad is not None, k
        ref = og[k].double()
        nrm = float(og[k[:-4] + "weight"].double().norm()) if _zero_gradient_bias(k) else float(ref.norm())
        e = float((_unpad(p.grad, og[k]).cpu().double() - ref).norm()) / max(1e-9, nrm)
        worst = max(worst, (e, k))
        assert e <= 2e-2, "%s: relative L2 gradient error %.3e" % (k, e)        # fp32 through a batch-6 net that ends in a 2 x 2 map
    print("worst gradient error", worst)


def test_replayed_step_with_the_shufflenet_encoder_is_bit_equal_to_the_eager_step():
    """sat_amd/graph.py with the shufflenet encoder in bf16 mode: the step replayed from a hipGraph leaves the same loss, parameters, BatchNorm
    buffers and optimizer moments as the eager loop, bit for bit (see tests/test_gpu_graph.py)."""
    from sat_amd.graph import GraphedTrainStep
    over = dict(decoder_tf="always", lr_warmup_steps=3)
    eager, _, hp = _make_model(over)
    graphed, _, _ = _make_model(over)
    eager.set_precision("bf16"); graphed.set_precision("bf16")
    eager.configure_optimizers(); graphed.configure_optimizers()
    opt_e, opt_g = eager._train_optimizer(), graphed._train_optimizer()
    step = GraphedTrainStep(graphed, opt_g)
    batches = []
    for seed in (11, 23):
        img, caps, lengths = _batch(hp, B=4, seed=seed)
        batches.append((img.cuda(), caps.cuda(), lengths))
    for it in range(7):
        b = batches[it % 2]
        opt_e.zero_grad(set_to_none=True)
        out_e = eager.training_step(b, it)
        out_e["loss"].backward()
        opt_e.step()
        out_g = step(b, it)
        assert torch.equal(out_e["loss"].detach(), out_g["loss"]), "step %d: loss %r vs %r" % (it, float(out_e["loss"]), float(out_g["loss"]))
        for (k, x), (_, y) in zip(eager.state_dict().items(), graphed.state_dict().items()):
            assert torch.equal(x, y), "step %d: %s differs" % (it, k)
    assert step.stats["captured"] >= 2 and step.stats["replayed"] >= 3, dict(step.stats)


def test_cli_default_configuration_trains():
    """The reference's own defaults end to end (train.py:43-63, :75: shufflenet_v2_x0_5 at 224 px, no projection, embed 256, attention 128,
    decoder 512, decoder_tf None = always sample, Adam) with a batch of 16 images x 5 captions, bf16 storage (--precision 16): the loss is
    finite and falls over 12 steps on one repeated batch."""
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    from oracle import sat_oracle as O
    hp = O.default_hparams(encoder_arch="shufflenet_v2_x0_5", encoder_dim=None, input_size=224, vocab_size=2000, embed_dim=256, attention_dim=128,
                           decoder_dim=512, decoder_tf=None, opt="adam", encoder_lr=1e-5, decoder_lr=1e-3, embedding_lr=1e-2, precision=16, weight_decay=0.0,
                           adam_b1=0.9, adam_b2=0.999, momentum=0.9, nesterov=False, scheduler=None)
    torch.manual_seed(0)
    model = M.SAT(**vars(hp)).cuda().train()
    assert model.sat_precision == "bf16" and model.hp.encoder_dim == 1024
    model.configure_optimizers()
    opt = model._train_optimizer()
    img, caps, lengths = _batch(hp, B=16, R=5, T=18, seed=3)
    b = (img.cuda(), caps.cuda(), lengths)
    losses = []
    for it in range(12):
        opt.zero_grad(set_to_none=True)
        out = model.training_step(b, it)
        out["loss"].backward()
        opt.step()
        losses.append(float(out["loss"].detach()))
    assert all(l == l and abs(l) < 1e4 for l in losses), losses
    assert losses[-1] < losses[0], losses
