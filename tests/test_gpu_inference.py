"""GPU parity of the inference row (BASELINE config C5): SAT.caption / forward beam search on the HIP
step kernels against the reference fixtures (G7: token ids exact; scores, perplexities, alphas within 1e-4) and,
for a ResNet-backed model, against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_oracle_golden import _beam_cases, check_beam_against_golden, g9_cases, g9_state, sd_from  # noqa: E402


def test_g7_beam_search_on_hip(golden_dir):
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    from oracle import sat_oracle as O
    g = np.load(os.path.join(golden_dir, "g7_beam.npz"))
    sd = sd_from(g)
    V, m = sd["embedding.weight"].shape
    hp = O.default_hparams(vocab_size=V, embed_dim=m, decoder_dim=sd["lstm.weight_hh_l0"].shape[1],
                           encoder_dim=sd["attention.encoder_att.weight"].shape[1], attention_dim=sd["attention.encoder_att.weight"].shape[0])
    dec = M.SATDecoder(hp).cuda().eval()
    dec.load_decoder_state(sd)
    ann = torch.tensor(g["ann"])
    B, D, Hh, Ww = ann.shape
    ann_bld = ann.permute(0, 2, 3, 1).reshape(B, Hh * Ww, D).contiguous().cuda()
    for ci, beamk, rm, ra, mgl in _beam_cases(g):
        out = dec.beam_decode(ann_bld, (Hh, Ww), beamk=beamk, max_gen_length=mgl, rescore_method=rm, rescore_reward=0.5, return_all=ra)
        check_beam_against_golden(g, ci, ra, *out, tol=1e-4)


@pytest.mark.parametrize("arch,encoder_dim,es", [("resnet18", 32, 3), ("shufflenet_v2_x0_5", None, None), ("shufflenet_v2_x1_0", 32, None), ("mobilenet_v2", 48, 3)])
def test_caption_end_to_end_matches_oracle(arch, encoder_dim, es):
    """caption(img): eval-mode encoder (running statistics) + beam search, C5-style (beamk 5) on small models of every encoder family (the
    reference CLI's default encoder without projection: 1024-dimensional annotations)."""
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    from oracle import prng, sat_oracle as O
    over = dict(encoder_arch=arch, encoder_dim=encoder_dim, input_size=64, encoder_size=es, vocab_size=60, embed_dim=24, attention_dim=16,
                decoder_dim=40, deep_output=True)
    torch.manual_seed(11)
    model = M.SAT(**vars(O.default_hparams(**over))).cuda()
    oracle = O.OracleSAT(O.default_hparams(**over), {k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    img = torch.from_numpy(prng.uniform((3, 3, 64, 64), 31, 0.0, 1.0))
    caps, scores, alphas, ppl = model.caption(img.cuda(), beamk=5, max_gen_length=8, rescore_method="LN")
    assert not model.training
    oracle.encoder.eval()
    with torch.no_grad():
        ann = oracle.encoder(img.clone())
        ocaps, oscores, oalphas, oppl = O.beam_search(oracle.sd, oracle.hp, ann, beamk=5, max_gen_length=8, rescore_method="LN")
    assert caps == ocaps
    for a, b in zip(scores, oscores):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b))
    for a, b in zip(alphas, oalphas):
        assert a.shape == b.shape and float((a - b).abs().max()) <= 1e-4
    with pytest.raises(AssertionError):
        model.caption(img.cuda(), sample_method="greedy")


def test_g9_sampled_decoding_noise_and_stacked_layers_on_hip(golden_dir):
    """sample_method multinomial / topk (model.py:360-379), decoder_noise (model.py:322-324), decoder_layers=2 against the
    reference fixture.  The reference drew from the CPU generator under torch.manual_seed; the HIP path gets the same
    draws by sampling its device-computed probabilities with that generator (the `multinomial` / `randn` hooks)."""
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    g = np.load(os.path.join(golden_dir, "g9_sampled.npz"))
    ann = torch.tensor(g["ann"])
    B, D, Hh, Ww = ann.shape
    ann_bld = ann.permute(0, 2, 3, 1).reshape(B, Hh * Ww, D).contiguous().cuda()
    for c in g9_cases(g):
        sd, hp = g9_state(g, c["layers"])
        hp.embed_dim = sd["embedding.weight"].shape[1]
        hp.encoder_dim, hp.attention_dim = sd["attention.encoder_att.weight"].shape[1], sd["attention.encoder_att.weight"].shape[0]
        dec = M.SATDecoder(hp).cuda().eval()
        dec.load_decoder_state(sd)
        torch.manual_seed(c["seed"])
        out = dec.beam_decode(ann_bld, (Hh, Ww), beamk=c["beamk"], max_gen_length=c["mgl"], rescore_method="LN", return_all=c["return_all"],
                              sample_method=c["method"], sample_topk=c["topk"], decoder_noise=c["noise"],
                              multinomial=lambda p, k: torch.multinomial(p.cpu(), k), randn=lambda shape: torch.randn(shape))
        check_beam_against_golden(g, c["ci"], c["return_all"], *out, tol=1e-4)


def test_sampled_decoding_runs_on_device_generators():
    """default samplers (torch.multinomial / torch.randn on the GPU): structural checks only, the draws are not reproducible."""
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    from oracle import prng, sat_oracle as O
    hp = O.default_hparams(vocab_size=50, encoder_dim=16, embed_dim=12, attention_dim=8, decoder_dim=20, decoder_layers=2)
    dec = M.SATDecoder(hp).cuda().eval()
    ann_bld = torch.from_numpy(prng.uniform((2, 6, 16), 77, 0.0, 1.0)).cuda()
    for method in ("multinomial", "topk"):
        caps, scores, alphas, ppl = dec.beam_decode(ann_bld, (2, 3), beamk=3, max_gen_length=5, sample_method=method, decoder_noise=0.1, return_all=True)
        assert len(caps) == 2 and all(len(c) >= 1 for c in caps)
        for c, a, sc in zip(caps, alphas, scores):
            assert all(al.shape == (len(tok), 2, 3) for tok, al in zip(c, a)) and all(np.isfinite(x) for x in sc)


def test_validation_step_scores_generated_captions():
    """val_batch / validation_step / score_captions (model.py:646-697): captions from the HIP beam search, BLEU/GLEU from
    metrics.py, best mean-embedding cosine similarity -- against the same quantities from the oracle's captions on the CPU."""
    import sat_amd  # noqa: F401
    from sat_amd import metrics, model as M
    from oracle import prng, sat_oracle as O
    import torch.nn.functional as F
    over = dict(encoder_arch="resnet18", encoder_dim=32, input_size=64, encoder_size=3, vocab_size=60, embed_dim=24, attention_dim=16,
                decoder_dim=40, deep_output=True, val_beamk=3, val_max_len=7)
    torch.manual_seed(5)
    model = M.SAT(**vars(O.default_hparams(**over))).cuda()
    oracle = O.OracleSAT(O.default_hparams(**over), {k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    B, R, T = 4, 3, 9
    img = torch.from_numpy(prng.uniform((B, 3, 64, 64), 41, 0.0, 1.0))
    caps, lengths = prng.captions(B, R, T, 60, 42, min_len=3)
    caps, lengths = torch.from_numpy(caps), torch.from_numpy(lengths)
    got = model.validation_step((img.cuda(), caps.cuda(), lengths), 0)
    oracle.encoder.eval()
    with torch.no_grad():
        ann = oracle.encoder(img.clone())
        ocaps, _, _, oppl = O.beam_search(oracle.sd, oracle.hp, ann, beamk=3, max_gen_length=7, temperature=1.0, rescore_method="LN")
        refs = [[c[1:l] for c, l in zip(r, lengths[i].tolist())] for i, r in enumerate(caps.tolist())]
        E = oracle.sd["embedding.weight"]
        best = []
        for i in range(B):
            cv = E[torch.tensor(ocaps[i])].mean(0, keepdim=True)
            best.append(max(float(F.cosine_similarity(E[caps[i][j][1:int(lengths[i][j])]].mean(0, keepdim=True), cv)) for j in range(R)))
    expect = {"bleu1": metrics.corpus_bleu(refs, ocaps, (1, 0, 0, 0)), "bleu4": metrics.corpus_bleu(refs, ocaps), "gleu": metrics.corpus_gleu(refs, ocaps),
              "cosine_similarity": sum(best) / B, "perplexity": sum(oppl) / B}
    assert set(got) == {"bleu1", "bleu2", "bleu3", "bleu4", "cosine_similarity", "gleu", "perplexity"}
    for k, v in expect.items():
        assert abs(got[k] - v) <= 1e-4 * max(1.0, abs(v)), (k, got[k], v)
    means = model.validation_epoch_end([got, got])
    assert abs(means["gleu"] - got["gleu"]) < 1e-12
    assert model.training_epoch_end([{"loss": torch.tensor(2.0), "accuracy": 0.5}, {"loss": torch.tensor(4.0), "accuracy": 0.0}]) == {"loss": 3.0, "accuracy": 0.25}


def test_g7_batched_beam_search_on_hip(golden_dir):
    """The batched search (every image at once, no host round trip per step) against the same reference fixture as the
    per-image loop: token ids exact, scores / perplexities / attention maps within 1e-4, list order included."""
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    from oracle import sat_oracle as O
    g = np.load(os.path.join(golden_dir, "g7_beam.npz"))
    sd = sd_from(g)
    V, m = sd["embedding.weight"].shape
    hp = O.default_hparams(vocab_size=V, embed_dim=m, decoder_dim=sd["lstm.weight_hh_l0"].shape[1],
                           encoder_dim=sd["attention.encoder_att.weight"].shape[1], attention_dim=sd["attention.encoder_att.weight"].shape[0])
    dec = M.SATDecoder(hp).cuda().eval()
    dec.load_decoder_state(sd)
    ann = torch.tensor(g["ann"])
    B, D, Hh, Ww = ann.shape
    ann_bld = ann.permute(0, 2, 3, 1).reshape(B, Hh * Ww, D).contiguous().cuda()
    for ci, beamk, rm, ra, mgl in _beam_cases(g):
        out = dec.beam_decode_batched(ann_bld, (Hh, Ww), beamk=beamk, max_gen_length=mgl, rescore_method=rm, rescore_reward=0.5, return_all=ra)
        check_beam_against_golden(g, ci, ra, *out, tol=1e-4)


@pytest.mark.parametrize("layers", [1, 2])
def test_batched_beam_search_equals_the_per_image_loop(layers):
    """larger random model, 9 images, beams 1 / 4, temperature schedule, two LSTM layers: identical captions and list order,
    scores and maps within 1e-5 of the per-image path."""
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    from oracle import prng, sat_oracle as O
    hp = O.default_hparams(vocab_size=83, encoder_dim=32, embed_dim=24, attention_dim=16, decoder_dim=40, decoder_layers=layers)
    torch.manual_seed(3 + layers)
    dec = M.SATDecoder(hp).cuda().eval()
    ann_bld = torch.from_numpy(prng.uniform((9, 12, 32), 55, 0.0, 1.0)).cuda()
    for beamk, rm, ra in [(1, None, False), (4, "BAR", True), (4, "LN", False)]:
        a = dec.beam_decode(ann_bld, (3, 4), beamk=beamk, max_gen_length=9, temperature=[1.0, 0.7], rescore_method=rm, return_all=ra)
        b = dec.beam_decode_batched(ann_bld, (3, 4), beamk=beamk, max_gen_length=9, temperature=[1.0, 0.7], rescore_method=rm, return_all=ra)
        assert a[0] == b[0], (beamk, rm)
        flat = lambda x: [v for e in x for v in (e if isinstance(e, list) else [e])]
        for u, v in zip(flat(a[1]), flat(b[1])):
            assert abs(u - v) <= 1e-5 * max(1.0, abs(u))
        for u, v in zip(flat(a[2]), flat(b[2])):
            assert u.shape == v.shape and float((u - v).abs().max()) <= 1e-5
        for u, v in zip(flat(a[3]), flat(b[3])):
            assert abs(u - v) <= 1e-5 * max(1.0, abs(u))


@pytest.mark.parametrize("method", ["multinomial", "topk"])
def test_batched_sampled_search_equals_the_per_image_loop_on_the_same_draws(method):
    """sample_method multinomial / topk (+ decoder noise) for all images at once: with the Gumbel / normal variates supplied as
    tables, the batched search returns what the per-image HIP loop returns when its ``multinomial`` hook draws by the same rule
    (top k of log p + the same variates; oracle.gumbel_topk, statistically identical to torch.multinomial) and its ``randn`` hook
    reads the same normals.  The per-image loop with torch's own samplers is pinned to the reference by G9."""
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    from oracle import prng, sat_oracle as O
    hp = O.default_hparams(vocab_size=83, encoder_dim=32, embed_dim=24, attention_dim=16, decoder_dim=40, decoder_layers=2)
    torch.manual_seed(21)
    dec = M.SATDecoder(hp).cuda().eval()
    B, K, S, V, st, n, NL = 5, 4, 8, 83, 3, 40, 2
    ann = torch.from_numpy(prng.uniform((B, 12, 32), 91, 0.0, 1.0)).cuda()
    g = torch.Generator().manual_seed(5)
    width = V if method == "multinomial" else st
    u = torch.rand(S + 1, B * K, width, generator=g).clamp_(1e-9, 1 - 1e-7)
    gum = (-torch.log(-torch.log(u))).cuda().contiguous()
    normals = torch.randn(S + 1, NL, B * K, n, generator=g).cuda().contiguous()
    noise = 0.3
    kw = dict(beamk=K, max_gen_length=S, temperature=[1.0, 0.8], rescore_method="LN", return_all=True, sample_method=method, sample_topk=st, decoder_noise=noise)
    got = dec.beam_decode_batched(ann, (3, 4), gumbel=gum, normals=normals, **kw)
    for b in range(B):
        state = {"step": 0, "nstep": 0}

        def draw(probs, k, b=b, state=state):
            state["step"] += 1                                            # called once per step >= 1
            s_ = state["step"]
            if method == "multinomial":
                rows = probs.numel() // V
                gv = gum[s_, b * K:b * K + rows, :].reshape(-1)
            else:
                rows = probs.numel() // st
                gv = gum[s_, b * K:b * K + rows, :].reshape(-1)
            return O.gumbel_topk(probs, k, gv)

        def randn(shape, b=b, state=state):
            s_ = state["nstep"]; state["nstep"] += 1                      # called once per step >= 0
            return normals[s_, :, b * K:b * K + shape[1], :]

        one = dec.beam_decode(ann[b:b + 1], (3, 4), multinomial=draw, randn=randn, **kw)
        assert one[0][0] == got[0][b], (method, b, one[0][0], got[0][b])
        for x, y in zip(one[1][0], got[1][b]):
            assert abs(x - y) <= 2e-5 * max(1.0, abs(x))
        for x, y in zip(one[2][0], got[2][b]):
            assert x.shape == y.shape and float((x - y).abs().max()) <= 2e-5


def test_batched_sampled_search_with_the_device_generator():
    """no tables: the counter-based generator.  Same seed -> same captions, another seed -> other draws; every hypothesis is
    well formed; forward() routes sampled decoding through the batched search."""
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    from oracle import prng, sat_oracle as O
    hp = O.default_hparams(vocab_size=83, encoder_dim=32, embed_dim=24, attention_dim=16, decoder_dim=40)
    torch.manual_seed(22)
    dec = M.SATDecoder(hp).cuda().eval()
    ann = torch.from_numpy(prng.uniform((7, 12, 32), 92, 0.0, 1.0)).cuda()
    END = hp.vocab_stoi["<END>"]
    for method in ("multinomial", "topk"):
        kw = dict(beamk=4, max_gen_length=10, temperature=1.0, return_all=True, sample_method=method, sample_topk=3, decoder_noise=0.2)
        a = dec.beam_decode_batched(ann, (3, 4), seed=7, **kw)
        b = dec.beam_decode_batched(ann, (3, 4), seed=7, **kw)
        c = dec.beam_decode_batched(ann, (3, 4), seed=8, **kw)
        assert a[0] == b[0] and a[1] == b[1]
        assert a[0] != c[0]
        for caps, scores, alphas in zip(a[0], a[1], a[2]):
            assert len(caps) == 4 and len(scores) == 4
            for cap, sc, al in zip(caps, scores, alphas):
                assert 0 < len(cap) <= 11 and all(0 <= t < 83 for t in cap) and END not in cap[:-1] and sc == sc
                assert al.shape[0] == len(cap) and float((al.sum((1, 2)) - 1).abs().max()) < 1e-4


def test_batched_beam_search_replayed_from_a_hipgraph():
    """graph=True captures the search once and replays it: identical output to the eager call, also for new annotations in the
    static buffer, after an in-place weight update (the graph reads the live parameters), and for a second shape."""
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    from oracle import prng, sat_oracle as O
    hp = O.default_hparams(vocab_size=83, encoder_dim=32, embed_dim=24, attention_dim=16, decoder_dim=40)
    torch.manual_seed(11)
    dec = M.SATDecoder(hp).cuda().eval()
    kw = dict(beamk=4, max_gen_length=9, temperature=[1.0, 0.7], rescore_method="BAR", return_all=True)

    def same(a, b):
        assert a[0] == b[0] and a[1] == b[1] and a[3] == b[3]
        for u, v in zip([x for e in a[2] for x in e], [x for e in b[2] for x in e]):
            assert torch.equal(u, v)

    for seed in (55, 56, 57):
        ann = torch.from_numpy(prng.uniform((9, 12, 32), seed, 0.0, 1.0)).cuda()
        same(dec.beam_decode_batched(ann, (3, 4), **kw), dec.beam_decode_batched(ann, (3, 4), graph=True, **kw))
    assert len(dec._beam_graphs) == 1
    with torch.no_grad():
        dec.output.output.weight.mul_(1.3); dec.lstm.weight_hh_l0.add_(0.01)
    same(dec.beam_decode_batched(ann, (3, 4), **kw), dec.beam_decode_batched(ann, (3, 4), graph=True, **kw))
    assert len(dec._beam_graphs) == 1
    ann5 = torch.from_numpy(prng.uniform((5, 12, 32), 58, 0.0, 1.0)).cuda()
    same(dec.beam_decode_batched(ann5, (3, 4), **kw), dec.beam_decode_batched(ann5, (3, 4), graph=True, **kw))
    same(dec.beam_decode_batched(ann, (3, 4), **kw), dec.beam_decode_batched(ann, (3, 4), graph=True, **kw))
    assert len(dec._beam_graphs) == 2


def test_batched_beam_search_at_c5_decoder_shapes():
    """BASELINE configs[4] decoder shapes (D = 512, n = 512, V = 6400, L = 49; beam 5): the batched search and its hipGraph
    replay against the per-image loop, fp32 mode (in bf16 mode the two paths round differently shaped GEMMs)."""
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    from oracle import prng, sat_oracle as O
    hp = O.default_hparams(vocab_size=6400, encoder_dim=512, embed_dim=256, attention_dim=128, decoder_dim=512)
    torch.manual_seed(5)
    dec = M.SATDecoder(hp).cuda().eval()
    ann = torch.from_numpy(prng.uniform((12, 49, 512), 95, 0.0, 2.0)).cuda()
    kw = dict(beamk=5, max_gen_length=14, temperature=1.0, rescore_method="LN", return_all=True)
    a = dec.beam_decode(ann, (7, 7), **kw)
    b = dec.beam_decode_batched(ann, (7, 7), **kw)
    c = dec.beam_decode_batched(ann, (7, 7), graph=True, **kw)
    assert a[0] == b[0]
    flat = lambda x: [v for e in x for v in e]
    for u, v in zip(flat(a[1]), flat(b[1])):
        assert abs(u - v) <= 2e-5 * max(1.0, abs(u))
    for u, v in zip(flat(a[2]), flat(b[2])):
        assert u.shape == v.shape and float((u - v).abs().max()) <= 2e-5
    assert b[0] == c[0] and b[1] == c[1] and all(torch.equal(u, v) for u, v in zip(flat(b[2]), flat(c[2])))


@pytest.mark.parametrize("images,beamk", [(12, 5), (64, 5), (64, 1)])
def test_c5_beam_search_against_the_cpu_oracle(images, beamk):
    """BASELINE configs[4] ("greedy-vs-beam") at its real decoder dimensions (D = 512, n = 512, A = 128, m = 256, V = 6400, L = 49; 64 images), beam 5
    AND greedy (beam 1): the batched on-device search against the CPU oracle's per-image beam search (oracle.sat_oracle.beam_search =
    model.py:237-472 restated; pinned by fixture G7): captions token for token, list order included; scores, perplexities and attention maps 1e-4."""
    import os
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    from oracle import prng, sat_oracle as O
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    hp = O.default_hparams(vocab_size=6400, encoder_dim=512, embed_dim=256, attention_dim=128, decoder_dim=512)
    torch.manual_seed(5)
    dec = M.SATDecoder(hp).cuda().eval()
    sd = {k: v.detach().cpu().clone() for k, v in dec.state_dict().items()}
    ann_bld = torch.from_numpy(prng.uniform((images, 49, 512), 95, 0.0, 2.0))
    ann_img = ann_bld.reshape(images, 7, 7, 512).permute(0, 3, 1, 2).contiguous()
    kw = dict(beamk=beamk, max_gen_length=14, temperature=1.0, rescore_method="LN", return_all=True)
    with torch.no_grad():
        want = O.beam_search(sd, hp, ann_img, **kw)
    got = dec.beam_decode_batched(ann_bld.cuda(), (7, 7), **kw)
    assert got[0] == want[0], "captions differ from the CPU oracle"
    flat = lambda x: [v for e in x for v in e]                            # noqa: E731
    for u, v in zip(flat(got[1]), flat(want[1])):
        assert abs(u - v) <= 1e-4 * max(1.0, abs(v)), (u, v)
    for u, v in zip(flat(got[3]), flat(want[3])):
        assert abs(u - v) <= 1e-4 * max(1.0, abs(v)), (u, v)
    for u, v in zip(flat(got[2]), flat(want[2])):
        assert u.shape == v.shape and float((u - v).abs().max()) <= 1e-4


@pytest.mark.parametrize("beamk", [5, 1])
def test_caption_end_to_end_resnet50_at_256px_against_the_cpu_oracle(beamk):
    """``caption()`` of BASELINE configs[4] end to end - resnet50 trunk at 256 px in eval mode, 1x1 projection, encoder_size 7, then the batched
    search - on 4 images in fp32 parity mode against the CPU oracle (its encoder + ``beam_search``): token ids exact, scores / perplexities /
    attention maps 1e-4.  (Untrained BatchNorm running statistics let activations grow block by block: the residual branches are damped, as in
    the train-step tests, so that the comparison is about the arithmetic and not about 1e4-sized annotations.)"""
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    from oracle import prng, sat_oracle as O
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    over = dict(encoder_arch="resnet50", encoder_dim=512, input_size=256, encoder_size=7, vocab_size=6400, embed_dim=256, attention_dim=128,
                decoder_dim=512, deep_output=True)
    torch.manual_seed(21)
    model = M.SAT(**vars(O.default_hparams(**over)))
    with torch.no_grad():
        for blk in [b for li in (5, 6, 7, 8) for b in model.encoder[li]]:
            blk.bn3.weight.fill_(0.25)
    model = model.cuda()
    model.set_precision("fp32")
    oracle = O.OracleSAT(O.default_hparams(**over), {k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    img = torch.from_numpy(prng.uniform((4, 3, 256, 256), 77, 0.0, 1.0))
    kw = dict(beamk=beamk, max_gen_length=12, rescore_method="LN")
    caps, scores, alphas, ppl = model.caption(img.cuda(), **kw)
    oracle.encoder.eval()
    with torch.no_grad():
        ann = oracle.encoder(img.clone())
        ocaps, oscores, oalphas, oppl = O.beam_search(oracle.sd, oracle.hp, ann, **kw)
    assert ann.shape == (4, 512, 7, 7)
    assert caps == ocaps
    for a, b in zip(scores, oscores):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (a, b)
    for a, b in zip(ppl, oppl):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (a, b)
    for a, b in zip(alphas, oalphas):
        assert a.shape == b.shape and float((a - b).abs().max()) <= 1e-4
