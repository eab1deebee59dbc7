"""GPU parity of the whole train step (encoder -> decoder -> losses -> backward) behind the reference's
``SAT`` surface (train_batch / training_step / configure_optimizers) against the CPU oracle, plus the
drop-in checks that do not need the reference at run time."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).abs().max()) / max(1.0, float(b.abs().max()))


def make(hp_over=None, seed=42, damp_residual=None):
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    from oracle import sat_oracle as O
    over = dict(encoder_arch="resnet18", encoder_dim=32, input_size=64, encoder_size=3, vocab_size=120, embed_dim=24,
                attention_dim=16, decoder_dim=40, deep_output=True, weight_decay=0.0, decoder_lr=1e-3, embedding_lr=1e-2,
                encoder_lr=1e-5, opt="adam", adam_b1=0.9, adam_b2=0.999, momentum=0.9, nesterov=False, scheduler=None)
    over.update(hp_over or {})
    hp = O.default_hparams(**over)
    torch.manual_seed(seed)
    model = M.SAT(**vars(hp))
    if damp_residual is not None:              # scale the last BatchNorm of every residual branch (a well-conditioned net)
        with torch.no_grad():
            for blk in [b for li in (5, 6, 7, 8) for b in model.encoder[li]]:
                (blk.bn3 if blk.kind == "bottleneck" else blk.bn2).weight.fill_(damp_residual)
    oracle = O.OracleSAT(O.default_hparams(**over), {k: v.clone() for k, v in model.state_dict().items()})
    return model.cuda().train(), oracle, hp


def batch(hp, B=6, R=3, T=9, seed=5):
    from oracle import prng
    img = torch.from_numpy(prng.uniform((B, 3, hp.input_size, hp.input_size), seed, 0.0, 1.0))
    caps, lengths = prng.captions(B, R, T, hp.vocab_size, seed + 1)
    return img, torch.from_numpy(caps), torch.from_numpy(lengths)


@pytest.mark.parametrize("eps", [1.0, 0.0])
def test_training_step_matches_oracle(eps):
    model, oracle, hp = make(dict(decoder_tf="always" if eps == 1.0 else None))
    img, caps, lengths = batch(hp)
    loss_o, out_o = oracle.step_loss(img, caps, lengths, eps)
    loss_o.backward()
    img_g = img.cuda()
    metrics = model.training_step((img_g, caps.cuda(), lengths), 0)
    assert torch.equal(img_g.cpu(), img), "the HIP encoder must not mutate its input (SURVEY F9)"
    assert abs(metrics["loss"].item() - loss_o.item()) <= 1e-4 * max(1.0, abs(loss_o.item()))
    assert abs(float(metrics["accuracy"]) - float(out_o["acc"])) < 1e-6
    assert metrics["epsilon_tf"] == eps
    lp, tp, alphas = model.train_batch((img_g, caps.cuda(), lengths), eps)
    assert rel(lp.data, out_o["logits_packed"]) <= 2e-4 and rel(alphas, out_o["alphas"]) <= 1e-4
    assert torch.equal(tp.data.cpu(), out_o["targets_packed"]) and lp.batch_sizes.tolist() == out_o["batch_sizes"]
    padded, lens = torch.nn.utils.rnn.pad_packed_sequence(lp, batch_first=True)       # PackedSequence is well formed
    assert lens.tolist() == lengths.reshape(-1).tolist() and padded.shape[0] == lengths.numel()
    metrics["loss"].backward()
    og = oracle.named_grads()
    for k, p in model.named_parameters():
        assert p.grad is not None, k
        e = float((p.grad.cpu().double() - og[k].double()).norm()) / max(1e-9, float(og[k].double().norm()))
        assert e <= 2e-2, "%s: relative L2 gradient error %.3e" % (k, e)        # fp32 through a batch-6 ResNet, see test_gpu_encoder
    for k in ("attention.f_att.weight", "output.output.weight", "lstm.weight_hh_l0", "embedding.weight", "encoder.9.weight"):
        p = dict(model.named_parameters())[k]
        assert rel(p.grad, og[k]) <= 5e-4, k                                     # decoder-side and last-layer grads are tight


def test_optimizer_groups_and_one_adam_step_follow_the_oracle():
    model, oracle, hp = make(dict(encoder_finetune_after=1, decoder_tf="always"))
    opt = model.configure_optimizers()
    # model.py:723-746: decoder (no-decay, decay), embedding, encoder (no-decay, decay)
    assert [len(g["params"]) > 0 for g in opt.param_groups] == [True] * 5
    assert [g["lr"] for g in opt.param_groups] == [1e-3, 1e-3, 1e-2, 1e-5, 1e-5] and model.opt_init_lr == [1e-3, 1e-3, 1e-2, 1e-5, 1e-5]
    img, caps, lengths = batch(hp)
    params_o = oracle.parameters()
    opt_o = torch.optim.Adam([{"params": [p for p in oracle.sd.values() if p is not oracle.sd["embedding.weight"]], "lr": 1e-3},
                              {"params": [oracle.sd["embedding.weight"]], "lr": 1e-2},
                              {"params": list(oracle.encoder.parameters()), "lr": 1e-5}])
    for _ in range(2):
        opt.zero_grad(); opt_o.zero_grad()
        model.training_step((img.cuda(), caps.cuda(), lengths), 0)["loss"].backward()
        oracle.step_loss(img, caps, lengths, 1.0)[0].backward()
        opt.step(); opt_o.step()
    sd = model.state_dict()
    for k, v in oracle.sd.items():
        assert rel(sd[k], v) <= 1e-3, k
    l1 = model.training_step((img.cuda(), caps.cuda(), lengths), 0)["loss"].item()
    l0 = oracle.step_loss(img, caps, lengths, 1.0)[0].item()
    assert abs(l1 - l0) <= 2e-3 * max(1.0, abs(l0)), (l1, l0)
    assert len(params_o) == len(list(model.parameters()))


def test_frozen_encoder_and_state_dict_roundtrip():
    model, oracle, hp = make()
    for p in model.encoder.parameters():
        p.requires_grad = False
    img, caps, lengths = batch(hp)
    model.training_step((img.cuda(), caps.cuda(), lengths), 0)["loss"].backward()
    assert all(p.grad is None for p in model.encoder.parameters())
    assert all(p.grad is not None for k, p in model.named_parameters() if not k.startswith("encoder."))
    import sat_amd  # noqa
    from sat_amd import model as M
    clone = M.SAT(**vars(hp)).cuda()
    clone.load_state_dict(model.state_dict())
    clone.eval(); model.eval()
    with torch.no_grad():
        a, _ = model.encode(img.cuda()); b, _ = clone.encode(img.cuda())
    assert torch.equal(a, b)


def test_pretrained_trunk_from_a_local_checkpoint_trains_only_what_the_reference_trains(tmp_path):
    """model.py:18-24: pretrained=<torchvision-format file> loads the trunk and freezes it; one step against the oracle holding the
    same state (annotations-side parameters: projection and decoder gradients, fp32 mode, 1e-3 of the largest entry)."""
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    from oracle import sat_oracle as O
    import os
    torch.manual_seed(11)
    net = O.ResNetOracle("resnet18")
    with torch.no_grad():
        for mod in net.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5); mod.bias.uniform_(-0.3, 0.3); mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.5, 2.0)
    path = os.path.join(str(tmp_path), "resnet18-feedbeef.pth")
    torch.save(net.state_dict(), path)
    over = dict(encoder_arch="resnet18", encoder_dim=32, input_size=64, encoder_size=3, vocab_size=120, embed_dim=24, attention_dim=16, decoder_dim=40,
                deep_output=True, decoder_tf="always")
    hp = O.default_hparams(**over); hp.pretrained = path
    torch.manual_seed(3)
    model = M.SAT(**vars(hp))
    oracle = O.OracleSAT(O.default_hparams(**over), {k: v.clone() for k, v in model.state_dict().items()})
    model = model.cuda().train()
    frozen = {k for k, p in model.named_parameters() if not p.requires_grad}
    assert frozen == {k for k, _ in model.named_parameters() if k.startswith("encoder.") and not k.startswith("encoder.9.")}
    img, caps, lengths = batch(hp)
    out = model.training_step((img.cuda(), caps.cuda(), lengths), 0); out["loss"].backward()
    loss, _ = oracle.step_loss(img.clone(), caps, lengths, epsilon=1.0); loss.backward()
    assert rel(out["loss"], loss) < 1e-4
    og = oracle.named_grads()
    for k, p in model.named_parameters():
        if k in frozen:
            assert p.grad is None, k
        else:
            assert rel(p.grad, og[k]) < 1e-3, (k, rel(p.grad, og[k]))


def test_one_process_gradsync_keeps_gradients_in_place_and_the_optimizer_fast_path_changes_nothing():
    """One process: GradSync keeps every gradient in its persistent bucket slice (no collective), so gradient addresses repeat and
    FusedOptimizer reuses the pointer table already on the device.  Four steps with it against four steps without: every parameter,
    Adam state and BatchNorm buffer bit-identical; the fast path ran on steps 2..4."""
    from sat_amd.dist import GradSync
    runs = []
    for use_sync in (False, True):
        model, _, hp = make(dict(decoder_tf="always"))
        model.set_precision("bf16")
        opt = model.configure_optimizers()
        sync = GradSync(model) if use_sync else None
        img, caps, lengths = batch(hp)
        img, caps = img.cuda(), caps.cuda()
        for _ in range(4):
            opt.zero_grad(set_to_none=True)
            model.training_step((img, caps, lengths), 0)["loss"].backward()
            if sync is not None:
                sync.finish()
            opt.step()
        torch.cuda.synchronize()
        if use_sync:
            assert getattr(opt, "fast_path_steps", 0) == 3, getattr(opt, "fast_path_steps", 0)
            assert all(p.grad is None or sync._of[id(p)].owns(p) for p in model.parameters())
            sync.remove()
        runs.append(({k: v.detach().clone() for k, v in model.state_dict().items()},
                     [st["exp_avg"].clone() for st in opt.state.values() if "exp_avg" in st]))
    (sd_a, m_a), (sd_b, m_b) = runs
    for k in sd_a:
        assert torch.equal(sd_a[k], sd_b[k]), k
    assert len(m_a) == len(m_b) and all(torch.equal(x, y) for x, y in zip(m_a, m_b))


BF16_CASES = {
    # name: (hparams, images, residual damping)
    "small": (dict(decoder_tf="always", encoder_dim=32, embed_dim=32, attention_dim=16, decoder_dim=64, vocab_size=128, input_size=128, encoder_size=None), 16, 0.25),
    # resnet50 at 256 px (stage-1 maps of 32768 rows: the real step's tile sizes), C1-width decoder
    "c2-encoder": (dict(decoder_tf="always", encoder_arch="resnet50", encoder_dim=256, embed_dim=256, attention_dim=128, decoder_dim=512, vocab_size=640,
                        input_size=256, encoder_size=7), 8, 0.25),
    # the same net as torch initialises it (no damping): the case whose ReLU flips made the fp32-oracle comparison useless
    "c2-encoder-undamped": (dict(decoder_tf="always", encoder_arch="resnet50", encoder_dim=256, embed_dim=256, attention_dim=128, decoder_dim=512, vocab_size=640,
                                 input_size=256, encoder_size=7), 8, None),
}


@pytest.mark.parametrize("size", sorted(BF16_CASES))
def test_bf16_mode_against_the_bf16_rounding_oracle(size):
    """hip_precision="bf16" (BASELINE configs[1]) has no reference counterpart (the reference only has fp16 AMP).  Its yardstick
    is the CPU oracle with bf16 rounding applied at the SAME storage points (oracle/bf16_emulation.py: activations, filter copies, GEMM
    operands).  Forward: ReLU / max-pool decisions agree, so loss, logits and attention weights are compared DIRECTLY with that oracle
    (loss 2e-3 relative, logits 2e-2 of their range, alphas 2e-3 absolute).  Backward: the gradient that enters the encoder carries a
    large per-channel common mode (spatial mean of InitLSTM, attention context) that the first BatchNorm backward subtracts again; stored
    in bf16, what is left of it is rounding noise of the common mode, so two correct bf16 implementations - this one and the CPU
    emulation - differ by tens of percent on encoder gradients once their inputs differ in the last fp32 bits (the layer-, block- and
    whole-encoder tests in test_gpu_encoder.py compare backward passes from IDENTICAL bf16 inputs and are the tight ones: 3e-2).  Here
    the criterion is the cost of the storage format itself, measured by the emulation: for every gradient tensor the HIP path's error
    against the fp32 oracle must not exceed twice the emulation's error against the fp32 oracle plus 2e-2."""
    import os
    from oracle import bf16_emulation as B16
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    over, nb, damp = BF16_CASES[size]
    model, oracle, hp = make(over, damp_residual=damp)
    model.set_precision("bf16")
    img, caps, lengths = batch(hp, B=nb)
    loss_f32, _ = oracle.step_loss(img, caps, lengths, 1.0)
    loss_f32.backward()
    g32 = {k: v.detach().clone().double() for k, v in oracle.named_grads().items()}
    for p in oracle.parameters():
        p.grad = None
    loss_o, out_o = B16.step_loss(oracle, img, caps, lengths, 1.0)
    loss_o.backward()
    lp, tp, alphas = model.train_batch((img.cuda(), caps.cuda(), lengths), 1.0)
    assert lp.data.dtype == torch.float32 and alphas.dtype == torch.float32
    assert rel(lp.data, out_o["logits_packed"]) <= 2e-2, rel(lp.data, out_o["logits_packed"])
    assert float((alphas.cpu() - out_o["alphas"].detach()).abs().max()) <= 2e-3
    m = model.training_step((img.cuda(), caps.cuda(), lengths), 0)
    assert abs(m["loss"].item() - loss_o.item()) <= 2e-3 * abs(loss_o.item()), (m["loss"].item(), loss_o.item())
    assert abs(m["loss"].item() - loss_f32.item()) <= 3e-2 * abs(loss_f32.item())
    m["loss"].backward()
    og = oracle.named_grads()
    rows = []
    for k, p in model.named_parameters():
        assert p.grad is not None and p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all(), k
        nrm = max(1e-12, float(g32[k].norm()))
        e_hip = float((p.grad.cpu().double() - g32[k]).norm()) / nrm            # what bf16 storage costs on the HIP path ...
        e_emu = float((og[k].double() - g32[k]).norm()) / nrm                   # ... and in the CPU emulation of the same storage points
        rows.append((e_hip - 2 * e_emu, e_hip, e_emu, k))
    rows.sort(reverse=True)
    print("bf16 mode: (HIP error, emulation error) against the fp32 oracle, worst margins:", [(round(a, 4), round(b, 4), k) for _, a, b, k in rows[:5]])
    print("           largest emulation errors:", sorted([(round(b, 4), k) for _, a, b, k in rows], reverse=True)[:3])
    # every tensor, in the test log (pytest -s / -rP) and, on the GPU box, in gpurun_out/: a regression that stays inside the 2 x e_emu
    # envelope still shows as a ratio e_hip / e_emu that moved
    table = ["%-44s e_hip %.5f  e_emu %.5f  ratio %5.2f  margin %+.5f" % (k, a, b, a / max(b, 1e-12), m) for m, a, b, k in sorted(rows, key=lambda r: r[3])]
    print("\n".join(["bf16 mode, relative L2 gradient error per tensor (HIP vs fp32 oracle | bf16-emulation vs fp32 oracle):"] + table))
    import os
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "bf16_margins.txt"), "w") as f:
            f.write("\n".join(table) + "\n")
    assert rows[0][0] <= 2e-2, rows[:4]


def test_bf16_filter_copies_stay_current_through_optimizer_steps():
    """bf16 mode reads bf16 copies of the conv filters.  The copy is made once (encoder.py) and then rewritten by the fused
    optimizer's update kernel; after every step it must equal a fresh cast of the fp32 master weight, bit for bit, and two
    trainings -- one keeping the copies, one re-casting every step -- must produce identical losses."""
    losses = []
    for keep in (True, False):
        model, _, hp = make(dict(encoder_finetune_after=1, decoder_tf="always", encoder_lr=1e-3))
        model.set_precision("bf16")
        model.__dict__["_sat_global_step"] = 2
        opt = model.configure_optimizers()
        img, caps, lengths = batch(hp)
        img, caps = img.cuda(), caps.cuda()
        run = []
        for step in range(4):
            opt.zero_grad(set_to_none=True)
            out = model.training_step((img, caps, lengths), 0)
            out["loss"].backward()
            opt.step()
            run.append(float(out["loss"]))
            convs = [p for n, p in model.encoder.named_parameters() if p.dim() == 4 and hasattr(p, "_sat_bf16_shadow")]
            assert len(convs) >= 20
            for p in convs:
                if keep:
                    assert p._sat_shadow_version == p._version
                    assert torch.equal(p._sat_bf16_shadow.float(), p.detach().to(torch.bfloat16).float())
                else:
                    del p._sat_bf16_shadow                       # force a fresh cast at the next forward
        losses.append(run)
    assert losses[0] == losses[1], losses
    assert losses[0][-1] < losses[0][0]


def test_c2_full_size_step_is_reproducible_bit_for_bit():
    """BASELINE configs[1] at full size (resnet50, 128 images x 5 captions, T = 22, bf16 mode): properties that need no CPU
    run - two identical models stepped on the same batch agree bit for bit in loss, every gradient and every updated
    parameter (all reductions have a fixed order: no floating-point atomics anywhere on the path), recycled device memory
    poisoned in between; the loss is finite, starts near ln(V), and BatchNorm running statistics move."""
    import math
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    hp, T, B, R = bench.hparams("c2")
    img, caps, lengths = bench.synthetic_batch(B, R, T, hp["vocab_size"], 1234, False)
    img, caps = img.cuda(), caps.cuda()

    def run():
        torch.manual_seed(42)
        model = M.SAT(**hp).cuda().train(); model.set_precision("bf16")
        model.__dict__["_sat_global_step"] = 2
        opt = model.configure_optimizers()
        losses = []
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            out = model.training_step((img.clone(), caps, lengths), 0)
            out["loss"].backward()
            grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
            opt.step()
            losses.append(out["loss"].detach().clone())
        return losses, grads, {k: v.detach().clone() for k, v in model.state_dict().items()}

    l1, g1, s1 = run()
    junk = torch.full((1 << 26,), float("nan"), device="cuda"); del junk           # recycled blocks now hold NaN
    l2, g2, s2 = run()
    assert all(torch.equal(a, b) for a, b in zip(l1, l2)), (l1, l2)
    assert [k for k in g1 if not torch.equal(g1[k], g2[k])] == []
    assert [k for k in s1 if not torch.equal(s1[k], s2[k])] == []
    assert all(math.isfinite(float(l)) for l in l1) and abs(float(l1[0]) - math.log(hp["vocab_size"])) < 1.0
    assert all(bool(torch.isfinite(v).all()) for v in g1.values()) and len(g1) > 150
    rm = s1["encoder.2.running_mean"]
    assert float(rm.abs().max()) > 0 and int(s1["encoder.2.num_batches_tracked"]) == 3      # model.py:46-48's probe forward at construction + 2 steps


@pytest.mark.parametrize("cfg,ragged", [("c2", True), ("c1", False), ("c3", True), ("c4", True)])
def test_full_size_steps_stay_finite(cfg, ragged):
    """BASELINE-size steps on ragged captions (C3 / C4: the per-GPU shards of configs[2] / [3], resnet101 L=196 with 32 images and
    wide_resnet101_2 D=1024 T=32 with 64 images): every gradient and parameter finite after three optimizer steps, loss falling.
    (ReLU written as fmaxf maps NaN to 0, so a kernel that produces NaN in the trunk does not show in the loss: check the tensors.)"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    hp, T, B, R = bench.hparams(cfg)
    torch.manual_seed(42)
    model = M.SAT(**hp).cuda().train(); model.set_precision("bf16")
    model.__dict__["_sat_global_step"] = 2
    opt = model.configure_optimizers()
    img, caps, lengths = bench.synthetic_batch(B, R, T, hp["vocab_size"], 77, ragged)
    img, caps = img.cuda(), caps.cuda()
    losses = []
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        out = model.training_step((img, caps, lengths), 0)
        out["loss"].backward()
        bad = [k for k, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
        assert bad == [], bad[:8]
        opt.step()
        losses.append(float(out["loss"].detach()))
    assert [k for k, p in model.named_parameters() if not bool(torch.isfinite(p).all())] == []
    assert [k for k, b in model.named_buffers() if b.is_floating_point() and not bool(torch.isfinite(b).all())] == []
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("cfg", ["c3", "c4"])
def test_c3_c4_shard_steps_are_reproducible_bit_for_bit(cfg):
    """BASELINE configs[2] / [3] at their per-GPU shard sizes (bench.py --config c3 / c4), bf16 mode: two identical models stepped on the
    same batch agree bit for bit in loss and every gradient (fixed-order reductions on every tile shape these nets launch)."""
    import math
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    hp, T, B, R = bench.hparams(cfg)
    img, caps, lengths = bench.synthetic_batch(B, R, T, hp["vocab_size"], 4321, True)
    img, caps = img.cuda(), caps.cuda()

    def run():
        torch.manual_seed(42)
        model = M.SAT(**hp).cuda().train(); model.set_precision("bf16")
        model.__dict__["_sat_global_step"] = 2
        out = model.training_step((img.clone(), caps, lengths), 0)
        out["loss"].backward()
        return out["loss"].detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    l1, g1 = run()
    junk = torch.full((1 << 26,), float("nan"), device="cuda"); del junk
    l2, g2 = run()
    assert torch.equal(l1, l2) and math.isfinite(float(l1)) and abs(float(l1) - math.log(hp["vocab_size"])) < 1.5
    assert [k for k in g1 if not torch.equal(g1[k], g2[k])] == []
    assert all(bool(torch.isfinite(v).all()) for v in g1.values()) and len(g1) > 300
