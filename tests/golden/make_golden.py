"""Generate tests/golden/*.npz by running the REFERENCE (imported unmodified
from /root/reference through ref_shim) on seeded inputs.  Build container only.

    python tests/golden/make_golden.py

Fixtures hold numbers only (inputs, weights, expected outputs).  SURVEY 8c list:
G1 SoftAttention fwd+grads, G2 InitLSTM (F3), G3 one decode step, G4 train_batch
(+losses, DS bit pattern), G5 gradients for G4, G6 LabelSmoothing, G8 config-size
spot checks (C1 decoder shapes; weights regenerated from oracle/prng.py).
"""
import os
import sys

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

from oracle import prng, sat_oracle as O  # noqa: E402

torch.set_num_threads(1)   # fixed reduction order while generating
ref = ref_shim.load_reference()


class _PassThrough(nn.Module):
    """Stands where ``self.encoder`` is so that a fixture can feed annotations."""

    def forward(self, x):
        return x


def build(hp, seed):
    model = ref_shim.make_sat(ref, hp)
    model.encoder = _PassThrough()
    sd_np = prng.decoder_state(hp, seed)
    missing = model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=False)
    assert not missing.unexpected_keys, missing
    assert all(k.startswith("encoder") for k in missing.missing_keys), missing
    model.train()
    return model, sd_np


def f32bits(x):
    return np.asarray(np.float32(x)).view(np.uint32)


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print("%-28s %7.1f KB" % (name, os.path.getsize(path) / 1024))


def small_hp(**over):
    base = dict(vocab_size=23, encoder_dim=12, embed_dim=10, attention_dim=7, decoder_dim=9, input_size=64)
    base.update(over)
    return O.default_hparams(**base)


# ---------------------------------------------------------------- G1
def g1():
    hp = small_hp()
    model, sd = build(hp, 11)
    N, H, W = 5, 2, 3
    ann = torch.from_numpy(prng.uniform((N, hp.encoder_dim, H, W), 101)).requires_grad_()
    hid = torch.from_numpy(prng.uniform((N, hp.decoder_dim), 102)).requires_grad_()
    gz = torch.from_numpy(prng.uniform((N, hp.encoder_dim), 103))
    ga = torch.from_numpy(prng.uniform((N, H, W), 104))
    z, alpha = model.attention(ann, hid)
    ((z * gz).sum() + (alpha * ga).sum()).backward()
    att = model.attention
    save("g1_attention", ann=ann.detach(), hid=hid.detach(), gz=gz, ga=ga,
         We=sd["attention.encoder_att.weight"], Wd=sd["attention.decoder_att.weight"], wf=sd["attention.f_att.weight"],
         z=z.detach(), alpha=alpha.detach(), d_ann=ann.grad, d_hid=hid.grad,
         d_We=att.encoder_att.weight.grad, d_Wd=att.decoder_att.weight.grad, d_wf=att.f_att.weight.grad)


# ---------------------------------------------------------------- G2
def g2():
    out = {}
    for layers in (1, 2):
        for N in (4, 5):
            hp = small_hp(decoder_layers=layers)
            model, sd = build(hp, 20 + layers)
            ann = torch.from_numpy(prng.uniform((N, hp.encoder_dim, 3, 2), 200 + N))
            h0, c0 = model.init_lstm(ann)
            tag = "L%d_N%d_" % (layers, N)
            out.update({tag + "ann": ann, tag + "h0": h0.detach(), tag + "c0": c0.detach()})
            for k in ("init_lstm.factorize.weight", "init_lstm.factorize.bias", "init_lstm.init.weight", "init_lstm.init.bias"):
                out["L%d_%s" % (layers, k)] = sd[k]
    save("g2_initlstm", **out)


# ---------------------------------------------------------------- G3
def g3():
    for tag, over in (("deep", {}), ("shallow", dict(deep_output=False)),
                      ("tied", dict(weight_tying=True)), ("layers2", dict(decoder_layers=2))):
        hp = small_hp(**over)
        model, sd = build(hp, 30)
        N, H, W = 4, 3, 2
        ann = torch.from_numpy(prng.uniform((N, hp.encoder_dim, H, W), 301))
        h = torch.from_numpy(prng.uniform((hp.decoder_layers, N, hp.decoder_dim), 302))
        c = torch.from_numpy(prng.uniform((hp.decoder_layers, N, hp.decoder_dim), 303))
        tok = torch.from_numpy(prng.integers((N,), 304, 0, hp.vocab_size))
        with torch.no_grad():                                   # model.py:526-547, one iteration
            y = model.embedding_dropout(model.embedding(tok))
            z, alpha = model.attention(ann, h[-1])
            beta = model.beta(h[-1])
            h_in = torch.cat([y, beta * z], dim=1).unsqueeze(0)
            _, (hn, cn) = model.lstm(h_in, (h, c))
            logit = model.output(y, hn[-1], z)
        arrs = {"sd." + k: v for k, v in sd.items()}
        save("g3_step_" + tag, ann=ann, h=h, c=c, tok=tok, z=z, alpha=alpha, beta=beta, hn=hn, cn=cn, logit=logit, **arrs)


# ---------------------------------------------------------------- G4 + G5
def g4(tag, eps, seed, B=3, R=2, T=8, H=2, W=3, **over):
    hp = small_hp(**over)
    model, sd = build(hp, seed)
    ann = torch.from_numpy(prng.uniform((B, hp.encoder_dim, H, W), seed + 1)).requires_grad_()
    caps_np, len_np = prng.captions(B, R, T, hp.vocab_size, seed + 2)
    caps, lengths = torch.from_numpy(caps_np), torch.from_numpy(len_np)
    # The reference draws torch.rand(1) once per step > 2 (model.py:518).  Record the draws.
    n_draws = max(0, min(T - 1, int(len_np.max())) - 3)
    torch.manual_seed(seed)
    draws = np.array([float(torch.rand(1)) for _ in range(n_draws)], np.float64)
    torch.manual_seed(seed)
    lp, tp, alphas = model.train_batch((ann, caps, lengths), epsilon=eps)
    ce = model.criterion(lp.data, tp.data)
    ds = hp.att_gamma * ((1 - alphas.sum(dim=1)) ** 2).mean()
    loss = ce + ds                                                     # model.py:592-594
    pred = torch.argmax(lp.data, dim=1)
    acc = torch.sum(pred == tp.data) / pred.shape[0]
    loss.backward()
    grads = {"g." + k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    arrs = {"sd." + k: v for k, v in sd.items() if not (k == "output.output.weight" and hp.weight_tying and hp.deep_output)}
    save("g4_train_" + tag, ann=ann.detach(), caps=caps, lengths=lengths, epsilon=np.float64(eps), draws=draws,
         logits_packed=lp.data.detach(), targets_packed=tp.data, batch_sizes=lp.batch_sizes,
         sorted_indices=lp.sorted_indices, alphas=alphas.detach(),
         ce=ce.detach(), ds=ds.detach(), loss=loss.detach(), acc=acc,
         ce_bits=f32bits(ce.item()), ds_bits=f32bits(ds.item()), loss_bits=f32bits(loss.item()),
         d_ann=ann.grad, hp_label_smoothing=hp.label_smoothing, hp_att_gamma=hp.att_gamma,
         hp_deep_output=int(hp.deep_output), hp_weight_tying=int(hp.weight_tying), hp_layers=hp.decoder_layers,
         hp_embed_norm=(-1.0 if hp.embed_norm is None else hp.embed_norm), **arrs, **grads)


# ---------------------------------------------------------------- G6
def g6():
    util = sys.modules["util"]
    x = torch.from_numpy(prng.uniform((20, 10), 601, -3, 3)).requires_grad_()
    t = torch.from_numpy(prng.integers((20,), 602, 0, 10))
    out = dict(x=x.detach(), t=t, ce_torch=torch.nn.functional.cross_entropy(x, t).detach())
    for s in (0.0, 0.15, 0.3):
        x.grad = None
        loss = util.LabelSmoothing(s)(x, t)
        loss.backward()
        out["loss_%g" % s] = loss.detach()
        out["grad_%g" % s] = x.grad.clone()
    save("g6_label_smoothing", **out)


# ---------------------------------------------------------------- G7 (beam search, model.py:214-472)
def g7():
    hp = small_hp(vocab_size=31)
    model, sd = build(hp, 70)
    model.eval()
    ann = torch.from_numpy(prng.uniform((3, hp.encoder_dim, 2, 3), 701, 0.0, 1.5))
    arrs = {"sd." + k: v for k, v in sd.items()}
    arrs["ann"] = ann
    cases = [(1, None, False, 6), (3, None, False, 6), (5, "LN", False, 7), (3, "WR", True, 5), (4, "BAR", True, 6), (3, "LN", True, 4)]
    arrs["cases"] = np.array([[c[0], {None: 0, "LN": 1, "WR": 2, "BAR": 3}[c[1]], int(c[2]), c[3]] for c in cases])
    for ci, (beamk, rm, ra, mgl) in enumerate(cases):
        caps, scores, alphas, ppl = model.caption(ann, beamk=beamk, max_gen_length=mgl, temperature=1.0, sample_method="beam",
                                                  rescore_method=rm, rescore_reward=0.5, return_all=ra)
        for b in range(ann.shape[0]):
            cl = caps[b] if ra else [caps[b]]
            sl = scores[b] if ra else [scores[b]]
            al = alphas[b] if ra else [alphas[b]]
            pl_ = ppl[b] if ra else [ppl[b]]
            arrs["c%d_b%d_n" % (ci, b)] = np.int64(len(cl))
            for j in range(len(cl)):
                arrs["c%d_b%d_%d_tok" % (ci, b, j)] = np.array(cl[j], np.int64)
                arrs["c%d_b%d_%d_score" % (ci, b, j)] = np.float64(sl[j])
                arrs["c%d_b%d_%d_ppl" % (ci, b, j)] = np.float64(pl_[j])
                arrs["c%d_b%d_%d_alpha" % (ci, b, j)] = al[j].numpy()
    save("g7_beam", **arrs)



# ---------------------------------------------------------------- G9 (sampled decoding, decoder noise, stacked layers)
G9_CASES = [  # (layers, sample_method, sample_topk, decoder_noise, beamk, max_gen_length, return_all, seed)
    (1, "multinomial", 3, None, 3, 6, True, 11), (1, "topk", 2, None, 4, 6, False, 12), (1, "beam", 3, 0.3, 3, 5, True, 13),
    (2, "beam", 3, None, 3, 6, True, 14), (2, "multinomial", 3, 0.2, 3, 5, True, 15), (2, "topk", 3, 0.1, 2, 6, False, 16)]


def g9():
    """SAT.forward with sample_method multinomial / topk (model.py:360-379), decoder_noise (model.py:322-324) and
    decoder_layers = 2, run under torch.manual_seed(seed): the draws come from the CPU generator, so a restatement that
    makes the same calls in the same order under the same seed reproduces them."""
    arrs = {"cases": np.array([[c[0], ["beam", "multinomial", "topk"].index(c[1]), c[2], c[4], c[5], int(c[6]), c[7]] for c in G9_CASES]),
            "noise": np.array([0.0 if c[3] is None else c[3] for c in G9_CASES])}
    ann = torch.from_numpy(prng.uniform((2, 12, 2, 3), 901, 0.0, 1.5))
    arrs["ann"] = ann
    models = {}
    for layers in (1, 2):
        hp = small_hp(vocab_size=29, decoder_layers=layers)
        model, sd = build(hp, 90 + layers)
        model.eval()
        models[layers] = model
        arrs.update({"l%d.sd.%s" % (layers, k): v for k, v in sd.items()})
    for ci, (layers, method, topk, noise, beamk, mgl, ra, seed) in enumerate(G9_CASES):
        torch.manual_seed(seed)
        caps, scores, alphas, ppl = models[layers].caption(ann, beamk=beamk, max_gen_length=mgl, temperature=1.0, sample_method=method,
                                                           sample_topk=topk, decoder_noise=noise, rescore_method="LN", return_all=ra)
        for b in range(ann.shape[0]):
            cl = caps[b] if ra else [caps[b]]
            sl = scores[b] if ra else [scores[b]]
            al = alphas[b] if ra else [alphas[b]]
            pl_ = ppl[b] if ra else [ppl[b]]
            arrs["c%d_b%d_n" % (ci, b)] = np.int64(len(cl))
            for j in range(len(cl)):
                arrs["c%d_b%d_%d_tok" % (ci, b, j)] = np.array(cl[j], np.int64)
                arrs["c%d_b%d_%d_score" % (ci, b, j)] = np.float64(sl[j])
                arrs["c%d_b%d_%d_ppl" % (ci, b, j)] = np.float64(pl_[j])
                arrs["c%d_b%d_%d_alpha" % (ci, b, j)] = al[j].numpy()
    save("g9_sampled", **arrs)


# ---------------------------------------------------------------- G10 (validation metrics: the reference's own BLEU)
def g10():
    """token_bleu of dev/dev_corpus_metrics.py:19-55 (the author's BLEU written "straight from the paper", kept in the reference
    to be compared with nltk) on its own toy corpus and on seeded token corpora.  The module is imported as it stands (its
    top-level nltk import resolves to ref_shim's placeholders; its prints go to stdout)."""
    import contextlib, importlib.util, io
    spec = importlib.util.spec_from_file_location("ref_dev_corpus_metrics", os.path.join(ref_shim.REFERENCE_DIR, "dev", "dev_corpus_metrics.py"))
    mod = importlib.util.module_from_spec(spec)
    with contextlib.redirect_stdout(io.StringIO()):
        spec.loader.exec_module(mod)
    weights = [[1, 0, 0, 0], [0.5, 0.5, 0, 0], [0.33, 0.33, 0.33, 0], [0.25, 0.25, 0.25, 0.25], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]
    corpora = {"toy": (mod.references, mod.captions)}
    for ci, (nseg, vocab, seed) in enumerate([(12, 6, 1), (30, 9, 2), (8, 4, 3)]):
        rs = np.random.RandomState(seed)
        refs = [[rs.randint(1, vocab, size=rs.randint(5, 12)).tolist() for _ in range(rs.randint(2, 6))] for _ in range(nseg)]
        caps = [rs.randint(1, vocab, size=rs.randint(4, 13)).tolist() for _ in range(nseg)]
        corpora["rand%d" % ci] = (refs, caps)
    arrs = {"weights": np.array(weights), "names": np.array(sorted(corpora))}
    for name, (refs, caps) in corpora.items():
        arrs[name + "_nseg"] = np.int64(len(caps))
        for i, (rl, c) in enumerate(zip(refs, caps)):
            arrs["%s_cap%d" % (name, i)] = np.array(c, np.int64)
            arrs["%s_nref%d" % (name, i)] = np.int64(len(rl))
            for j, r in enumerate(rl):
                arrs["%s_ref%d_%d" % (name, i, j)] = np.array(r, np.int64)
        arrs[name + "_token_bleu"] = np.array([float(mod.token_bleu(refs, caps, list(w))) for w in weights])
    save("g10_metrics", **arrs)

# ---------------------------------------------------------------- G11 (input pipeline)
def g11():
    """Pillow's BILINEAR ``Image.resize`` (what torchvision's Resize / RandomResizedCrop call on PIL images, train.py:208-218)
    on seeded byte images, and the reference's own util.py helpers: BucketSampler order under np.random.seed (util.py:48-84),
    AddGaussianNoise under torch.manual_seed (util.py:121-130), crop_max_square (util.py:154-164)."""
    from PIL import Image
    import util as ref_util       # the reference's util.py (already imported by its model.py through ref_shim)
    rng = np.random.default_rng(1100)
    arrs = {}
    cases = [(37, 53, 16, 16), (48, 64, 24, 24), (50, 40, 64, 64), (61, 97, 32, 48), (32, 32, 32, 32), (100, 7, 9, 30), (5, 5, 17, 3), (120, 160, 56, 56)]
    arrs["resize_cases"] = np.array(cases, np.int64)
    for i, (h, w, oh, ow) in enumerate(cases):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if i == 7:      # a smooth picture: neighbouring pixels correlated like a photograph
            yy, xx = np.mgrid[0:h, 0:w]
            img = np.stack([yy * 255 // (h - 1), xx * 255 // (w - 1), (yy * 3 + xx * 2) % 256], -1).astype(np.uint8)
        arrs["resize_in%d" % i] = img
        arrs["resize_out%d" % i] = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    # crop then resize, as torchvision's resized_crop does on a PIL image
    img = rng.integers(0, 256, (90, 120, 3), dtype=np.uint8)
    boxes = [(3, 5, 80, 100), (0, 0, 90, 120), (10, 60, 33, 47)]
    arrs["crop_in"], arrs["crop_boxes"] = img, np.array(boxes, np.int64)
    for i, (t, l, h, w) in enumerate(boxes):
        arrs["crop_out%d" % i] = np.asarray(Image.fromarray(img).crop((l, t, l + w, t + h)).resize((28, 28), Image.BILINEAR))
    # crop_max_square (util.py:154-164; its resize uses Pillow's default filter for Image.resize = BICUBIC, not restated: size=None only)
    for i, (h, w) in enumerate([(30, 41), (41, 30), (17, 17)]):
        im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        arrs["square_in%d" % i] = im
        arrs["square_out%d" % i] = np.asarray(ref_util.crop_max_square(Image.fromarray(im), None))
    # BucketSampler
    rs = np.random.RandomState(5)
    lengths = [[int(v) for v in rs.randint(3, 9, size=3)] for _ in range(57)]
    arrs["bucket_lengths"] = np.array(lengths, np.int64)
    sampler = ref_util.BucketSampler(lengths, 8)
    np.random.seed(77)
    arrs["bucket_epoch0"] = np.array(list(iter(sampler)), np.int64)
    arrs["bucket_epoch1"] = np.array(list(iter(sampler)), np.int64)      # groups stay shuffled in place between epochs
    arrs["bucket_len"] = np.int64(len(sampler))
    # AddGaussianNoise
    x = torch.from_numpy(rng.random((3, 6, 5), dtype=np.float32))
    torch.manual_seed(9)
    arrs["noise_in"], arrs["noise_out"] = x.numpy(), ref_util.AddGaussianNoise(std=0.01)(x).numpy()
    save("g11_input_pipeline", **arrs)


# ---------------------------------------------------------------- G8 (C1 decoder shapes)
def g8():
    hp = O.default_hparams(vocab_size=6400, encoder_dim=256, embed_dim=256, attention_dim=128, decoder_dim=512, input_size=64)
    model, _ = build(hp, 80)
    B, R, T = 8, 5, 22
    ann = torch.from_numpy(prng.uniform((B, 256, 7, 7), 801, 0.0, 2.0)).requires_grad_()
    caps_np, len_np = prng.captions(B, R, T, 6400, 802, min_len=8)
    torch.set_num_threads(8)
    lp, tp, alphas = model.train_batch((ann, torch.from_numpy(caps_np), torch.from_numpy(len_np)), epsilon=1)
    ce = model.criterion(lp.data, tp.data)
    ds = ((1 - alphas.sum(dim=1)) ** 2).mean()
    (ce + ds).backward()
    g = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    sums = {"gsum." + k: v.double().sum() for k, v in g.items()}
    asums = {"gabs." + k: v.double().abs().sum() for k, v in g.items()}
    save("g8_c1_decoder", logits_head=lp.data[:64, :32].detach(), logits_sum=lp.data.double().sum().detach(),
         logits_abs=lp.data.double().abs().sum().detach(), alphas_head=alphas[:4].detach(), alphas_sum=alphas.double().sum().detach(),
         ce=ce.detach(), ds=ds.detach(), ds_bits=f32bits(ds.item()), d_ann_head=ann.grad[:2, :16], d_ann_sum=ann.grad.double().sum(),
         d_ann_abs=ann.grad.double().abs().sum(), n_tokens=np.int64(lp.data.shape[0]), **sums, **asums)
    torch.set_num_threads(1)


# ---------------------------------------------------------------- encoder (structural + reference get_encoder logic)
def g_encoder():
    hp = O.default_hparams(encoder_arch="resnet18", encoder_dim=32, input_size=64)
    torch.manual_seed(7)
    enc_ref = ref.get_encoder(hp)                 # reference slicing / 1x1 conv / Normalize over the repo's ResNet
    hp2 = O.default_hparams(encoder_arch="resnet18", encoder_dim=32, input_size=64)
    torch.manual_seed(7)
    enc_own = O.build_encoder(hp2)
    k1, k2 = list(enc_ref.state_dict().keys()), list(enc_own.state_dict().keys())
    assert k1 == k2, "state-dict keys differ"
    for k in k1:
        assert torch.equal(enc_ref.state_dict()[k], enc_own.state_dict()[k]), k
    img = torch.from_numpy(prng.uniform((2, 3, 64, 64), 901, 0.0, 1.0))
    y1 = enc_ref(img.clone())
    y2 = enc_own(img.clone())
    assert torch.equal(y1, y2), "oracle build_encoder != reference get_encoder on the same trunk"
    counts = {}
    for arch in ("resnet18", "resnet50", "resnet101", "wide_resnet101_2"):
        n, f = O.trunk_param_count(arch)
        counts["params." + arch] = np.int64(n)
        counts["features." + arch] = np.int64(f)
    save("g_encoder", keys=np.array(k1), out_shape=np.array(y1.shape), out_head=y1[0, :8].detach(),
         out_sum=y1.double().sum().detach(), seed=np.int64(7), **counts)


# ---------------------------------------------------------------- G12: learning-rate trace of the reference's own training_step
def g12():
    """model.py:614-626 (+ configure_optimizers, model.py:720-817) executed by the reference itself: per-batch learning rates of
    every parameter group under linear warm-up, CosineAnnealingWarmRestarts (per-batch stepping, t0 re-fit), OneCycleLR, and
    warm-up under gradient accumulation (trainer.global_step advances once per `accumulate` batches)."""
    from types import SimpleNamespace
    cases = {
        "warm_cosine": dict(scheduler="cosine", lr_warmup_steps=5, cosine_iterations=6, cosine_multi=1, accumulate=1),
        "warm_cosine_tm2": dict(scheduler="cosine", lr_warmup_steps=3, cosine_iterations=4, cosine_multi=2, accumulate=1),
        "one_cycle": dict(scheduler="one_cycle", lr_warmup_steps=4, accumulate=1),
        "warm_accumulate2": dict(scheduler=None, lr_warmup_steps=6, accumulate=2),
        "warm_plateau": dict(scheduler="plateau", lr_warmup_steps=4, accumulate=1),
    }
    out = {}
    for name, over in cases.items():
        hp = small_hp(opt="adam", decoder_lr=2e-3, embedding_lr=5e-3, encoder_lr=1e-4, weight_decay=1e-4, adam_b1=0.9, adam_b2=0.999,
                      momentum=0.9, nesterov=False, epochs=2, train_loader_len=10, min_lr=1e-6, lr_gamma=0.5, plateau_patience=1,
                      milestones=[1], one_cycle_pct=0.3, one_cycle_div=10.0, one_cycle_fdiv=100.0, decoder_tf=None, **over)
        model, _ = build(hp, 60)
        opt = model.configure_optimizers()
        model.optimizers = lambda opt=opt: opt
        model.logger = SimpleNamespace(experiment=SimpleNamespace(add_scalar=lambda *a, **k: None))
        model.current_epoch = 0
        B, R, T, H, W = 2, 2, 6, 2, 2
        ann = torch.from_numpy(prng.uniform((B, hp.encoder_dim, H, W), 61))
        caps_np, len_np = prng.captions(B, R, T, hp.vocab_size, 62)
        batch = (ann, torch.from_numpy(caps_np), torch.from_numpy(len_np))
        trace = []
        n_batches = hp.epochs * hp.train_loader_len
        for b in range(n_batches):
            g = b // hp.accumulate
            model.trainer = SimpleNamespace(global_step=g)
            model.global_step = g
            torch.manual_seed(1000 + b)
            metrics = model.training_step(batch, b)
            trace.append([pg["lr"] for pg in opt.param_groups])
            metrics["loss"].backward()
            if (b + 1) % hp.accumulate == 0:
                opt.step(); opt.zero_grad()
        out[name + ".lr"] = np.array(trace, np.float64)
        out[name + ".init_lr"] = np.array(model.opt_init_lr, np.float64)
        for k, v in over.items():
            out[name + ".hp." + k] = np.array(-1 if v is None else v) if not isinstance(v, str) else np.array(v)
    save("g12_lr_trace", **out)


if __name__ == "__main__":
    g1(); g2(); g3()
    g4("tf1", 1.0, 40)
    g4("tf0", 0.0, 41)
    g4("tf05", 0.5, 42, T=10)
    g4("smooth", 1.0, 43, label_smoothing=0.15)
    g4("shallow", 1.0, 44, deep_output=False)
    g4("tied", 1.0, 45, weight_tying=True)
    g4("layers2", 1.0, 46, decoder_layers=2)
    g4("embnorm", 1.0, 47, embed_norm=0.3)
    g4("gamma", 0.0, 48, att_gamma=0.5, B=4, R=3, T=9, H=3, W=3)
    g6(); g7(); g8(); g9(); g10(); g11(); g12(); g_encoder()
