"""The train step replayed from a hipGraph (sat_amd/graph.py) is the eager step: same launches, captured once per (shapes, packing plan,
teacher-forcing flags) and replayed.  Two models from one seed, one driven by the eager loop the reference's trainer runs
(zero_grad / training_step / backward / optimizer.step, model.py:559-628 + train.py:273-287), one by ``GraphedTrainStep``: after every
step the loss, every parameter, every BatchNorm buffer and the optimizer moments must be BIT-equal - over two alternating batches with
different caption lengths (two graphs), a learning rate that moves every step (warm-up: the pointer / learning-rate table is re-uploaded
beside the replays), Adam's step-dependent bias corrections (device-side hyper-parameters), and an in-place parameter change from outside
(the bf16 filter copies are remade: every graph is dropped and captured again)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(precision, over=None, seed=42):
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    from oracle import sat_oracle as O
    kw = dict(encoder_arch="resnet18", encoder_dim=64, input_size=64, encoder_size=3, vocab_size=120, embed_dim=24, attention_dim=16,
              decoder_dim=40, deep_output=True, weight_decay=0.0, decoder_lr=1e-3, embedding_lr=1e-2, encoder_lr=1e-4, opt="adam",
              adam_b1=0.9, adam_b2=0.999, momentum=0.9, nesterov=False, scheduler=None, decoder_tf="always", encoder_finetune_after=1,
              lr_warmup_steps=4)
    kw.update(over or {})
    hp = O.default_hparams(**kw)
    torch.manual_seed(seed)
    model = M.SAT(**vars(hp)).cuda().train()
    model.set_precision(precision)
    return model, hp


def _batch(hp, seed, B=4, R=3, T=9):
    from oracle import prng
    img = torch.from_numpy(prng.uniform((B, 3, hp.input_size, hp.input_size), seed, 0.0, 1.0))
    caps, lengths = prng.captions(B, R, T, hp.vocab_size, seed + 1)
    return img.cuda(), torch.from_numpy(caps).cuda(), torch.from_numpy(lengths)


def _same(a, b, what):
    for (k, x), (_, y) in zip(a, b):
        assert torch.equal(x, y), "%s: %s differs (max |d| = %.3e)" % (what, k, float((x.double() - y.double()).abs().max()))


def _state(model, opt):
    out = [(k, v.detach().clone()) for k, v in model.state_dict().items() if torch.is_tensor(v)]
    for i, g in enumerate(opt.param_groups):
        for j, p in enumerate(g["params"]):
            st = opt.state.get(p, {})
            for name in ("exp_avg", "exp_avg_sq", "momentum_buffer"):
                if st.get(name) is not None:
                    out.append(("opt.%d.%d.%s" % (i, j, name), st[name].detach().clone()))
    return out


@pytest.mark.parametrize("precision,tf", [("bf16", "always"), ("fp32", None)])
def test_replayed_step_is_bit_equal_to_the_eager_step(precision, tf):
    from sat_amd.graph import GraphedTrainStep
    over = dict(decoder_tf=tf)
    eager, hp = _make(precision, over)
    graphed, _ = _make(precision, over)
    _same(_state(eager, eager.configure_optimizers()), _state(graphed, graphed.configure_optimizers()), "construction")
    opt_e, opt_g = eager._train_optimizer(), graphed._train_optimizer()
    step = GraphedTrainStep(graphed, opt_g)
    batches = [_batch(hp, 11), _batch(hp, 23)]
    assert batches[0][2].tolist() != batches[1][2].tolist()
    for it in range(9):
        b = batches[it % 2]
        if it == 6:            # somebody rescales a filter in place (weight clamping, a loaded checkpoint): the bf16 copies must be remade
            with torch.no_grad():
                eager.encoder[1].weight.mul_(0.5); graphed.encoder[1].weight.mul_(0.5)
        opt_e.zero_grad(set_to_none=True)
        out_e = eager.training_step(b, it)
        out_e["loss"].backward()
        opt_e.step()
        out_g = step(b, it)
        assert torch.equal(out_e["loss"].detach(), out_g["loss"]), "step %d: loss %r vs %r" % (it, float(out_e["loss"]), float(out_g["loss"]))
        assert torch.equal(out_e["accuracy"], out_g["accuracy"])
        assert [g["lr"] for g in opt_e.param_groups] == [g["lr"] for g in opt_g.param_groups]
        _same(_state(eager, opt_e), _state(graphed, opt_g), "after step %d" % it)
    # steps 0, 1: eager (first sight of each plan); 2, 3: capture + replay; 4, 5: replay; 6: eager (copies remade, graphs dropped) ...
    assert step.stats["captured"] >= 2 and step.stats["replayed"] >= 5, dict(step.stats)
    assert eager.sat_global_step() == graphed.sat_global_step() == 9


def test_replay_with_the_persistent_gradient_buckets_and_norm_clipping():
    """``GradSync`` in one process keeps every gradient in its bucket slice; the captured backward writes there and the optimizer table never moves"""
    from sat_amd.dist import GradSync
    from sat_amd.graph import GraphedTrainStep
    eager, hp = _make("bf16")
    graphed, _ = _make("bf16")
    opt_e, opt_g = eager.configure_optimizers(), graphed.configure_optimizers()
    for o in (opt_e, opt_g):
        o.set_clipping("norm", 0.5)
    sync_e, sync_g = GradSync(eager), GradSync(graphed)
    step = GraphedTrainStep(graphed, opt_g, sync=sync_g)
    b = _batch(hp, 31)
    for it in range(5):
        opt_e.zero_grad(set_to_none=True)
        out_e = eager.training_step(b, it)
        out_e["loss"].backward()
        sync_e.finish()
        opt_e.step()
        out_g = step(b, it)
        assert torch.equal(out_e["loss"].detach(), out_g["loss"])
        _same(_state(eager, opt_e), _state(graphed, opt_g), "after step %d" % it)
    assert step.stats["replayed"] == 4 and float(opt_g.last_grad_norm) == float(opt_e.last_grad_norm)
    sync_e.remove(); sync_g.remove()


def test_what_a_graph_cannot_express_runs_eagerly():
    from sat_amd.graph import GraphedTrainStep
    model, hp = _make("bf16", dict(dropout=0.1))
    opt = model.configure_optimizers()
    step = GraphedTrainStep(model, opt)
    b = _batch(hp, 7)
    for it in range(3):
        out = step(b, it)
        assert torch.isfinite(out["loss"])
    assert step.stats["eager: dropout"] == 3 and step.stats["replayed"] == 0
