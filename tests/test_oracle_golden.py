"""CPU: the oracle (oracle/sat_oracle.py) against the fixtures captured from the
reference itself (tests/golden/make_golden.py).  Tolerance: the oracle runs the
same ATen ops in the same order, so on the generating host it is bit-equal;
another host's BLAS may round differently, hence 2e-6 relative-to-scale."""
import glob
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import prng, sat_oracle as O

TOL = 2e-6


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def close(a, b, tol=TOL):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, float(np.abs(b).max()) if b.size else 1.0)
    err = float(np.abs(a - b).max()) if b.size else 0.0
    assert err <= tol * scale, "max|d|=%g scale=%g" % (err, scale)


def sd_from(g, grad=False):
    sd = {k[3:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith("sd.")}
    if "output.output.weight" not in sd:               # tied
        sd["output.output.weight"] = sd["embedding.weight"]
    if grad:
        for k, v in sd.items():
            v.requires_grad_()
    return sd


def test_g1_attention(golden_dir):
    g = load(golden_dir, "g1_attention")
    sd = {"attention.encoder_att.weight": torch.tensor(g["We"], requires_grad=True),
          "attention.decoder_att.weight": torch.tensor(g["Wd"], requires_grad=True),
          "attention.f_att.weight": torch.tensor(g["wf"], requires_grad=True)}
    ann = torch.tensor(g["ann"], requires_grad=True); hid = torch.tensor(g["hid"], requires_grad=True)
    z, alpha = O.soft_attention(sd, ann, hid)
    close(z.detach(), g["z"]); close(alpha.detach(), g["alpha"])
    ((z * torch.tensor(g["gz"])).sum() + (alpha * torch.tensor(g["ga"])).sum()).backward()
    close(ann.grad, g["d_ann"]); close(hid.grad, g["d_hid"])
    close(sd["attention.encoder_att.weight"].grad, g["d_We"])
    close(sd["attention.decoder_att.weight"].grad, g["d_Wd"])
    close(sd["attention.f_att.weight"].grad, g["d_wf"])
    np.testing.assert_allclose(alpha.detach().sum((1, 2)).numpy(), 1.0, atol=1e-6)


@pytest.mark.parametrize("layers", [1, 2])
@pytest.mark.parametrize("N", [4, 5])
def test_g2_initlstm_batch_mixing(golden_dir, layers, N):
    g = load(golden_dir, "g2_initlstm")
    sd = {k: torch.tensor(g["L%d_%s" % (layers, k)]) for k in
          ("init_lstm.factorize.weight", "init_lstm.factorize.bias", "init_lstm.init.weight", "init_lstm.init.bias")}
    tag = "L%d_N%d_" % (layers, N)
    h0, c0 = O.init_state(sd, torch.tensor(g[tag + "ann"]), layers, 9)
    close(h0, g[tag + "h0"]); close(c0, g[tag + "c0"])
    # F3: h0[0, i] is chunk i of the flat (N, 2*layers*n) buffer, i.e. comes from row i // (2*layers)
    flat = torch.cat([h0, c0]).reshape(N, 2 * layers * 9)
    assert torch.equal(flat[1, :9], h0.reshape(-1, 9)[2 * layers])


@pytest.mark.parametrize("tag", ["deep", "shallow", "tied", "layers2"])
def test_g3_decode_step(golden_dir, tag):
    g = load(golden_dir, "g3_step_" + tag)
    sd = sd_from(g)
    layers = g["h"].shape[0]
    deep = "output.context.weight" in sd
    ann, h, c, tok = (torch.tensor(g[k]) for k in ("ann", "h", "c", "tok"))
    y = O.embed(sd, tok)
    z, alpha = O.soft_attention(sd, ann, h[-1])
    beta = O.beta_gate(sd, h[-1])
    x = torch.cat([y, beta * z], 1).unsqueeze(0)
    hn, cn = O.lstm_step(sd, x, h, c, layers)
    hm, cm = O.lstm_step_math(sd, x, h, c, layers)
    logit = O.deep_output(sd, y, hn[-1], z, deep)
    for got, key in ((z, "z"), (alpha, "alpha"), (beta, "beta"), (hn, "hn"), (cn, "cn"), (logit, "logit")):
        close(got, g[key])
    close(hm, g["hn"], 1e-5); close(cm, g["cn"], 1e-5)      # written-out cell vs ATen's


G4 = ["tf1", "tf0", "tf05", "smooth", "shallow", "tied", "layers2", "embnorm", "gamma"]


def g4_hp(g, sd):
    V, m = sd["embedding.weight"].shape
    n = sd["lstm.weight_hh_l0"].shape[1]
    return SimpleNamespace(decoder_layers=int(g["hp_layers"]), decoder_dim=n, vocab_size=V, embed_dim=m,
                           deep_output=bool(g["hp_deep_output"]), weight_tying=bool(g["hp_weight_tying"]),
                           label_smoothing=float(g["hp_label_smoothing"]), att_gamma=float(g["hp_att_gamma"]),
                           embed_norm=(None if g["hp_embed_norm"] < 0 else float(g["hp_embed_norm"])))


@pytest.mark.parametrize("tag", G4)
def test_g4_g5_train_batch_and_grads(golden_dir, tag, record_property):
    g = load(golden_dir, "g4_train_" + tag)
    sd = sd_from(g, grad=True)
    hp = g4_hp(g, sd)
    ann = torch.tensor(g["ann"], requires_grad=True)
    draws = iter(g["draws"].tolist())
    loss, out = O.training_loss(sd, hp, ann, torch.tensor(g["caps"]), torch.tensor(g["lengths"]),
                                float(g["epsilon"]), draw=lambda: next(draws))
    assert next(draws, None) is None, "the oracle must consume exactly the reference's RNG draws (F7)"
    assert out["batch_sizes"] == g["batch_sizes"].tolist()
    assert np.array_equal(out["targets_packed"].numpy(), g["targets_packed"])
    close(out["logits_packed"].detach(), g["logits_packed"])
    close(out["alphas"].detach(), g["alphas"])
    close(out["ce"].detach(), g["ce"]); close(out["ds"].detach(), g["ds"]); close(loss.detach(), g["loss"])
    close(out["acc"], g["acc"])
    ds_bits = np.float32(out["ds"].item()).view(np.uint32)
    ulps = abs(int(ds_bits) - int(g["ds_bits"]))
    record_property("ds_ulps", ulps)
    assert ulps <= 4, "doubly-stochastic loss differs by %d ulps" % ulps
    loss.backward()
    close(ann.grad, g["d_ann"], 1e-5)
    for k in g.files:
        if k.startswith("g."):
            name = k[2:]
            assert sd[name].grad is not None, name
            got = sd[name].grad
            if name == "embedding.weight" and hp.weight_tying and hp.deep_output:
                pass                                     # one shared tensor: grad already summed
            close(got, g[k], 1e-5)


def test_g6_label_smoothing(golden_dir):
    g = load(golden_dir, "g6_label_smoothing")
    t = torch.tensor(g["t"])
    for s in (0.0, 0.15, 0.3):
        x = torch.tensor(g["x"], requires_grad=True)
        loss = O.label_smoothing_ce(x, t, s)
        loss.backward()
        close(loss.detach(), g["loss_%g" % s]); close(x.grad, g["grad_%g" % s])
    # dev/dev_label_smoothing.py:18-27: smoothing 0 is plain cross entropy
    close(g["loss_0"], g["ce_torch"], 1e-6)


def test_g8_c1_decoder_shapes(golden_dir):
    """C1 decoder shapes (N=40, L=49, D=256, V=6400, T=22); weights regenerated from oracle/prng."""
    g = load(golden_dir, "g8_c1_decoder")
    hp = O.default_hparams(vocab_size=6400, encoder_dim=256, embed_dim=256, attention_dim=128, decoder_dim=512)
    sd = {k: torch.from_numpy(v).requires_grad_() for k, v in prng.decoder_state(hp, 80).items()}
    ann = torch.from_numpy(prng.uniform((8, 256, 7, 7), 801, 0.0, 2.0)).requires_grad_()
    caps, lengths = prng.captions(8, 5, 22, 6400, 802, min_len=8)
    loss, out = O.training_loss(sd, hp, ann, torch.from_numpy(caps), torch.from_numpy(lengths), 1.0)
    lp = out["logits_packed"].detach()
    assert lp.shape[0] == int(g["n_tokens"])
    close(lp[:64, :32], g["logits_head"], 1e-5)
    assert abs(lp.double().sum().item() - float(g["logits_sum"])) <= 1e-5 * float(g["logits_abs"])
    close(out["alphas"][:4].detach(), g["alphas_head"], 1e-5)
    close(out["ce"].detach(), g["ce"], 1e-5); close(out["ds"].detach(), g["ds"], 1e-5)
    loss.backward()
    close(ann.grad[:2, :16], g["d_ann_head"], 1e-4)
    for k in g.files:
        if k.startswith("gsum."):
            name = k[5:]
            got = sd[name].grad.double().sum().item()
            assert abs(got - float(g[k])) <= 1e-4 * max(1e-3, float(g["gabs." + name])), name


def test_encoder_structure(golden_dir):
    """dev/encoder_summaries.txt:2-18 parameter counts / feature dims, 256px -> 8x8 (F2),
    state-dict key layout and the F2 resize semantics (readme.md:118-121)."""
    g = load(golden_dir, "g_encoder")
    expect = {"resnet18": (11.18, 512), "resnet50": (23.51, 2048), "resnet101": (42.50, 2048), "wide_resnet101_2": (124.84, 2048)}
    for arch, (mparams, feat) in expect.items():
        assert round(int(g["params." + arch]) / 1e6, 2) == mparams
        assert int(g["features." + arch]) == feat
    n, f = O.trunk_param_count("resnet18")
    assert (n, f) == (int(g["params.resnet18"]), 512)
    hp = O.default_hparams(encoder_arch="resnet18", encoder_dim=32, input_size=64)
    torch.manual_seed(int(g["seed"]))
    enc = O.build_encoder(hp)
    assert list(enc.state_dict().keys()) == g["keys"].tolist()
    assert "1.weight" in enc.state_dict() and "9.bias" in enc.state_dict() and "5.0.conv1.weight" in enc.state_dict()
    img = torch.from_numpy(prng.uniform((2, 3, 64, 64), 901, 0.0, 1.0))
    y = enc(img.clone())
    assert list(y.shape) == g["out_shape"].tolist()
    close(y[0, :8].detach(), g["out_head"], 1e-4)       # same torch build => same initialiser stream
    for es, L in ((7, 7), (14, 14), (None, 8)):
        hp = O.default_hparams(encoder_arch="resnet18", encoder_dim=16, input_size=256, encoder_size=es)
        e = O.build_encoder(hp).eval()
        with torch.no_grad():
            assert e(torch.rand(1, 3, 256, 256)).shape == (1, 16, L, L)


def test_resnext_structure_matches_the_reference_summary():
    """dev/encoder_summaries.txt:12: resnext50_32x4d has 22.98 M parameters and 2048 features (the trunk without pooling / fc, model.py:28-29):
    torchvision's grouped Bottleneck (32 groups, 4 channels per group at the first stage), restated in the oracle and in the product."""
    n, f = O.trunk_param_count("resnext50_32x4d")
    assert round(n / 1e6, 2) == 22.98 and f == 2048
    import sat_amd  # noqa: F401
    from sat_amd import encoder as E
    hp = O.default_hparams(encoder_arch="resnext50_32x4d", encoder_dim=32, input_size=64)
    enc = E.get_encoder(hp)
    ref = O.build_encoder(O.default_hparams(encoder_arch="resnext50_32x4d", encoder_dim=32, input_size=64))
    assert list(enc.state_dict().keys()) == list(ref.state_dict().keys())
    assert all(tuple(a.shape) == tuple(b.shape) for a, b in zip(enc.state_dict().values(), ref.state_dict().values()))
    assert tuple(enc.state_dict()["5.0.conv2.weight"].shape) == (128, 4, 3, 3)          # 32 groups of 4 channels
    trunk = sum(p.numel() for k, p in enc.named_parameters() if not k.startswith("9."))
    assert trunk == n


def test_shufflenet_structure_matches_the_reference_summary():
    """dev/encoder_summaries.txt:28-35: shufflenet_v2_x0_5 / x1_0 / x1_5 / x2_0 = 0.34 / 1.25 / 2.48 / 5.34 M parameters without the classifier and
    1024 / 1024 / 1024 / 2048 features; the x0_5 trunk (the reference CLI's default arch, train.py:43) maps 224 px to a 7 x 7 grid."""
    expect = {"shufflenet_v2_x0_5": (0.34, 1024), "shufflenet_v2_x1_0": (1.25, 1024), "shufflenet_v2_x1_5": (2.48, 1024), "shufflenet_v2_x2_0": (5.34, 2048)}
    for arch, (mparams, feat) in expect.items():
        n, f = O.trunk_param_count(arch)
        assert round(n / 1e6, 2) == mparams and f == feat, arch
    hp = O.default_hparams(encoder_arch="shufflenet_v2_x0_5", encoder_dim=None, input_size=224)
    torch.manual_seed(5)
    ref = O.build_encoder(hp)
    assert hp.encoder_dim == 1024                                              # model.py:56-57 stores the trunk width back
    assert ref(torch.rand(2, 3, 224, 224)).shape == (2, 1024, 7, 7)
    keys = list(ref.state_dict().keys())
    assert keys[0] == "1.0.weight" and "3.0.branch1.0.weight" in keys and "5.3.branch2.6.bias" in keys and "6.1.running_var" in keys
    # channel_shuffle(groups = 2) interleaves the two halves
    x = torch.arange(8.0).view(1, 8, 1, 1)
    assert O.channel_shuffle(x, 2).flatten().tolist() == [0, 4, 1, 5, 2, 6, 3, 7]
    import sat_amd  # noqa: F401
    from sat_amd import encoder as E
    torch.manual_seed(5)
    hp2 = O.default_hparams(encoder_arch="shufflenet_v2_x0_5", encoder_dim=None, input_size=224)
    enc = E.get_encoder(hp2)
    assert hp2.encoder_dim == 1024 and keys == list(enc.state_dict().keys())
    assert all(torch.equal(v, ref.state_dict()[k]) for k, v in enc.state_dict().items() if "running" not in k and "num_batches" not in k)
    assert all(bool(((v - 0.9).abs() < 1e-6).all()) for k, v in enc.state_dict().items() if "running_var" in k)      # the zero-image probe
    for arch in ("shufflenet_v2_x1_0", "shufflenet_v2_x2_0"):                   # 58- / 122-channel branches are HELD at 64 / 128; the state dict keeps the reference's shapes
        torch.manual_seed(5)
        ref = O.build_encoder(O.default_hparams(encoder_arch=arch, encoder_dim=None, input_size=224))
        torch.manual_seed(5)
        enc = E.get_encoder(O.default_hparams(encoder_arch=arch, encoder_dim=None, input_size=224))
        sd, so = enc.state_dict(), ref.state_dict()
        assert list(sd) == list(so) and all(sd[k].shape == so[k].shape and torch.equal(sd[k], so[k]) for k in sd), arch
        assert any(p.shape[0] % 8 == 0 and p.shape[0] != so[k].shape[0] for k, p in enc.named_parameters()), arch
        enc.load_state_dict({k: v + 1 if v.dtype.is_floating_point else v for k, v in so.items()})
        assert all(torch.equal(v, so[k] + 1 if v.dtype.is_floating_point else so[k]) for k, v in enc.state_dict().items()), arch


def test_mobilenet_v2_structure_matches_the_reference_summary():
    """dev/encoder_summaries.txt:36-37: mobilenet_v2 = 1280 features, 2.22 M parameters without the classifier (model.py:38-39 keeps ``features``)."""
    n, f = O.trunk_param_count("mobilenet_v2")
    assert round(n / 1e6, 2) == 2.22 and f == 1280
    hp = O.default_hparams(encoder_arch="mobilenet_v2", encoder_dim=None, input_size=224)
    torch.manual_seed(5)
    ref = O.build_encoder(hp)
    assert hp.encoder_dim == 1280 and ref(torch.rand(2, 3, 224, 224)).shape == (2, 1280, 7, 7)
    keys = list(ref.state_dict().keys())
    assert keys[0] == "1.0.0.weight" and "1.1.conv.0.0.weight" in keys and "1.1.conv.2.running_var" in keys and "1.17.conv.3.bias" in keys and "1.18.1.weight" in keys
    assert tuple(ref.state_dict()["1.2.conv.1.0.weight"].shape) == (96, 1, 3, 3)          # depthwise 3x3 of the first t = 6 block
    import sat_amd  # noqa: F401
    from sat_amd import encoder as E
    torch.manual_seed(5)
    hp2 = O.default_hparams(encoder_arch="mobilenet_v2", encoder_dim=None, input_size=224)
    enc = E.get_encoder(hp2)
    assert hp2.encoder_dim == 1280 and keys == list(enc.state_dict().keys())
    assert all(torch.equal(v, ref.state_dict()[k]) for k, v in enc.state_dict().items() if "running" not in k and "num_batches" not in k)
    assert all(bool(((v - 0.9).abs() < 1e-6).all()) for k, v in enc.state_dict().items() if "running_var" in k)      # the zero-image probe
    for arch in ("mobilenet_v3_large", "squeezenet1_1", "densenet121", "mnasnet1_0"):          # model.py:32-41: not built, refused like an unknown name
        with pytest.raises(ValueError, match="Encoder not supported"):
            E.get_encoder(O.default_hparams(encoder_arch=arch, encoder_dim=None, input_size=224))


def test_fixtures_are_small(golden_dir):
    total = sum(os.path.getsize(p) for p in glob.glob(os.path.join(golden_dir, "*.npz")))
    assert total < 4 << 20


def _beam_cases(g):
    names = {0: None, 1: "LN", 2: "WR", 3: "BAR"}
    return [(ci, int(c[0]), names[int(c[1])], bool(c[2]), int(c[3])) for ci, c in enumerate(g["cases"])]


def check_beam_against_golden(g, ci, return_all, caps, scores, alphas, ppl, tol):
    B = g["ann"].shape[0]
    for b in range(B):
        n = int(g["c%d_b%d_n" % (ci, b)])
        cl = caps[b] if return_all else [caps[b]]
        sl = scores[b] if return_all else [scores[b]]
        al = alphas[b] if return_all else [alphas[b]]
        pl = ppl[b] if return_all else [ppl[b]]
        assert len(cl) == n
        for j in range(n):
            key = "c%d_b%d_%d_" % (ci, b, j)
            assert list(cl[j]) == g[key + "tok"].tolist(), (ci, b, j)
            assert abs(float(sl[j]) - float(g[key + "score"])) <= tol * max(1.0, abs(float(g[key + "score"])))
            assert abs(float(pl[j]) - float(g[key + "ppl"])) <= tol * max(1.0, abs(float(g[key + "ppl"])))
            a = np.asarray(al[j].cpu() if torch.is_tensor(al[j]) else al[j], np.float64)
            assert a.shape == g[key + "alpha"].shape and np.abs(a - g[key + "alpha"]).max() <= tol


def test_g7_beam_search(golden_dir):
    """model.py:214-472 (beam sampling; rescoring None / LN / WR / BAR; return_all both ways)."""
    g = load(golden_dir, "g7_beam")
    sd = sd_from(g)
    hp = O.default_hparams(vocab_size=sd["embedding.weight"].shape[0], deep_output=True)
    hp.decoder_dim = sd["lstm.weight_hh_l0"].shape[1]
    ann = torch.tensor(g["ann"])
    for ci, beamk, rm, ra, mgl in _beam_cases(g):
        with torch.no_grad():
            out = O.beam_search(sd, hp, ann, beamk=beamk, max_gen_length=mgl, rescore_method=rm, rescore_reward=0.5, return_all=ra)
        check_beam_against_golden(g, ci, ra, *out, tol=2e-6)


G9_METHODS = ["beam", "multinomial", "topk"]


def g9_cases(g):
    out = []
    for ci, c in enumerate(g["cases"]):
        noise = float(g["noise"][ci])
        out.append(dict(ci=ci, layers=int(c[0]), method=G9_METHODS[int(c[1])], topk=int(c[2]), beamk=int(c[3]), mgl=int(c[4]),
                        return_all=bool(c[5]), seed=int(c[6]), noise=(noise if noise != 0.0 else None)))
    return out


def g9_state(g, layers):
    pre = "l%d.sd." % layers
    sd = {k[len(pre):]: torch.tensor(g[k]) for k in g.files if k.startswith(pre)}
    hp = O.default_hparams(vocab_size=sd["embedding.weight"].shape[0], deep_output=True, decoder_layers=layers)
    hp.decoder_dim = sd["lstm.weight_hh_l0"].shape[1]
    return sd, hp


def test_g9_sampled_decoding_noise_and_stacked_layers(golden_dir):
    """model.py:360-379 (multinomial / topk sampling), 322-324 (decoder_noise) and decoder_layers=2, against the reference
    run under torch.manual_seed: the restatement makes the same generator calls in the same order."""
    g = load(golden_dir, "g9_sampled")
    ann = torch.tensor(g["ann"])
    for c in g9_cases(g):
        sd, hp = g9_state(g, c["layers"])
        torch.manual_seed(c["seed"])
        with torch.no_grad():
            out = O.beam_search(sd, hp, ann, beamk=c["beamk"], max_gen_length=c["mgl"], rescore_method="LN", return_all=c["return_all"],
                                sample_method=c["method"], sample_topk=c["topk"], decoder_noise=c["noise"])
        check_beam_against_golden(g, c["ci"], c["return_all"], *out, tol=2e-6)


def _g10_corpus(g, name):
    refs, caps = [], []
    for i in range(int(g[name + "_nseg"])):
        caps.append(g["%s_cap%d" % (name, i)].tolist())
        refs.append([g["%s_ref%d_%d" % (name, i, j)].tolist() for j in range(int(g["%s_nref%d" % (name, i)]))])
    return refs, caps


def test_g10_bleu_against_the_references_own_token_bleu(golden_dir):
    """metrics.corpus_bleu (nltk's corpus BLEU restated; nltk is absent, parity with it unpinned) against the reference's own
    independent BLEU, dev/dev_corpus_metrics.py:token_bleu, wherever the definitions coincide: every weighted n-gram order has
    at least one clipped match (token_bleu adds 1e-9 inside the log, nltk's method0 substitutes the smallest float otherwise)."""
    import sat_amd  # noqa: F401
    from sat_amd import metrics
    g = load(golden_dir, "g10_metrics")
    checked = 0
    for name in g["names"].tolist():
        refs, caps = _g10_corpus(g, name)
        for w, expect in zip(g["weights"].tolist(), g[name + "_token_bleu"].tolist()):
            nums = [sum(metrics.modified_precision(r, c, n)[0] for r, c in zip(refs, caps)) for n in range(1, 5)]
            if any(wi != 0 and nums[i] == 0 for i, wi in enumerate(w)):
                continue
            got = metrics.corpus_bleu(refs, caps, weights=w)
            assert abs(got - expect) <= 1e-6 * max(1.0, expect), (name, w, got, expect)
            checked += 1
    assert checked >= 20


def test_bleu_gleu_hand_computed_cases():
    import sat_amd  # noqa: F401
    from sat_amd import metrics
    refs = [[[1, 2, 3], [4, 5, 6]], [[1], [4, 5]]]
    caps = [[2, 4, 5, 6, 7], [1, 3, 6, 7, 8, 9]]
    assert abs(metrics.corpus_bleu(refs, caps, (1, 0, 0, 0)) - 5 / 11) < 1e-12            # clipped unigrams 4/5 and 1/6, no brevity penalty
    assert metrics.corpus_bleu([[[1, 2]]], [[3, 4]], (0.5, 0.5)) == 0                      # no unigram match
    # brevity penalty: hypothesis of 2 tokens against a closest reference of 4 -> exp(1 - 4/2)
    import math
    assert abs(metrics.corpus_bleu([[[1, 2, 3, 4]]], [[1, 2]], (1,)) - math.exp(-1.0)) < 1e-12
    # GLEU: hypothesis [1,2,3] vs reference [1,2,4]: shared 1..4-grams {1,2,(1,2)} = 3 of max(6, 6)
    assert abs(metrics.corpus_gleu([[[1, 2, 4]]], [[1, 2, 3]]) - 0.5) < 1e-12
    assert metrics.corpus_gleu([[[5]]], [[]]) == 0.0


def test_gumbel_topk_draws_like_torch_multinomial():
    """The batched search draws hypotheses as top-k of log p + Gumbel noise.  Statistical identity with torch.multinomial
    (without replacement): frequencies of the ORDERED pairs over 60,000 draws agree within 5 standard errors."""
    g = torch.Generator().manual_seed(123)
    p = torch.tensor([0.40, 0.05, 0.25, 0.0, 0.20, 0.10])             # one impossible category
    n, k, C = 60000, 2, 6
    ref = torch.multinomial(p.expand(n, C), k, replacement=False, generator=g)
    u = torch.rand(n, C, generator=g).clamp_(1e-12, 1 - 1e-7)
    keys = torch.log(p)[None, :] - torch.log(-torch.log(u))            # vectorised form of the same rule
    mine = torch.topk(keys, k, dim=1).indices
    assert torch.equal(mine[:5], torch.stack([O.gumbel_topk(p, k, -torch.log(-torch.log(u[i]))) for i in range(5)]))
    code = lambda x: (x[:, 0] * C + x[:, 1])
    fr, fm = torch.bincount(code(ref), minlength=C * C).double() / n, torch.bincount(code(mine), minlength=C * C).double() / n
    se = torch.sqrt(fr.clamp_min(1e-9) * (1 - fr) / n)
    assert float(((fr - fm).abs() / (se + 1e-4)).max()) < 5.0, (fr, fm)
    assert float(fm.reshape(C, C)[3].sum()) == 0 and float(fm.reshape(C, C)[:, 3].sum()) == 0      # p = 0 is never drawn
