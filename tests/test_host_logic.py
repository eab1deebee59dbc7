"""CPU tests of the host-side logic around the hot path: the per-batch learning-rate recipe of ``SAT.training_step``
(reference model.py:614-626) against the trace the REFERENCE itself produced (fixture G12, tests/golden/make_golden.py:g12),
and the algorithmic-work formulas bench.py reports against SURVEY 8d's table."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

from oracle import sat_oracle as O  # noqa: E402


def _hp(case, d):
    over = {}
    for k in d.files:
        if k.startswith(case + ".hp."):
            v = d[k]
            name = k[len(case) + 4:]
            over[name] = (str(v) if v.dtype.kind in "US" else (None if float(v) == -1 else (int(v) if float(v).is_integer() else float(v))))
    return over


@pytest.mark.parametrize("case", ["warm_cosine", "warm_cosine_tm2", "one_cycle", "warm_accumulate2", "warm_plateau"])
@pytest.mark.parametrize("fused", [True, False])
def test_learning_rate_trace_equals_the_reference(case, fused):
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    d = np.load(os.path.join(GOLD, "g12_lr_trace.npz"))
    over = _hp(case, d)
    hp = O.default_hparams(vocab_size=23, encoder_dim=12, embed_dim=10, attention_dim=7, decoder_dim=9, input_size=64,
                           opt="adam", decoder_lr=2e-3, embedding_lr=5e-3, encoder_lr=1e-4, weight_decay=1e-4, adam_b1=0.9, adam_b2=0.999,
                           momentum=0.9, nesterov=False, epochs=2, train_loader_len=10, min_lr=1e-6, lr_gamma=0.5, plateau_patience=1,
                           milestones=[1], one_cycle_pct=0.3, one_cycle_div=10.0, one_cycle_fdiv=100.0, decoder_tf=None,
                           fused_optimizer=fused, **over)
    model = M.SAT(**vars(hp))
    opt = model.configure_optimizers()
    assert np.allclose(model.opt_init_lr, d[case + ".init_lr"], rtol=0, atol=0)
    want = d[case + ".lr"]
    acc = int(over.get("accumulate", 1))
    got = []
    for b in range(want.shape[0]):
        # what training_step does after the forward pass (the forward itself needs the GPU): the LR recipe, then the step count
        g = model.sat_global_step()
        assert g == b // acc
        model.step_learning_rate(g)
        got.append([pg["lr"] for pg in opt.param_groups])
        micro = model.__dict__["_sat_micro_batches"] + 1
        if micro >= acc:
            model.__dict__["_sat_global_step"] += 1; micro = 0
        model.__dict__["_sat_micro_batches"] = micro
    got = np.array(got)
    assert got.shape == want.shape
    assert np.allclose(got, want, rtol=1e-12, atol=1e-15), (got[:6], want[:6])


def test_algorithmic_work_equals_the_survey_table():
    import bench
    table = {"c1": (4.75, 3.21, 8.30, 3.38), "c2": (10.81, 6.42, 9.76, 7.12), "c3": (20.51, 25.69, 9.95, 13.01), "c4": (59.70, 51.38, 14.88, 37.36)}
    for cfg, (enc, pre, step, cap) in table.items():
        w = bench.algorithmic_work(cfg)
        assert abs(w["f_enc"] / 1e9 - enc) < 0.006 and abs(w["f_pre"] / 1e6 - pre) < 0.006
        assert abs(w["f_step"] / 1e6 - step) < 0.006 and abs(w["f_cap"] / 1e9 - cap) < 0.006


def test_bench_starts_its_own_ranks_without_touching_the_gpu(monkeypatch, capsys):
    """`python bench.py --gpus 2` (no launcher): the parent only builds a torch.distributed.run command line and relays the JSON line."""
    import bench
    seen = {}

    class FakeProc:
        def __init__(self, cmd, **kw):
            seen["cmd"], seen["env"] = cmd, kw.get("env")
            self.stdout = iter(["noise\n", '{"metric": "m", "value": 1}\n'])

        def wait(self):
            return 0

    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    monkeypatch.delenv("RANK", raising=False)
    bench.main()
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "2", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert capsys.readouterr().out.strip() == '{"metric": "m", "value": 1}'


def _torchvision_style_checkpoint(tmp_path, arch, name):
    """a checkpoint with torchvision's key names (conv1.*, bn1.*, layerN.*, fc.*) and non-trivial BatchNorm state"""
    torch.manual_seed(5)
    net = O.ResNetOracle(arch)
    with torch.no_grad():
        for mod in net.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5); mod.bias.uniform_(-0.3, 0.3); mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.5, 2.0)
    path = os.path.join(str(tmp_path), name)
    torch.save(net.state_dict(), path)
    return net, path


@pytest.mark.parametrize("how", ["path", "directory"])
def test_pretrained_encoder_loads_a_local_torchvision_checkpoint_and_freezes_the_trunk(tmp_path, monkeypatch, how):
    """model.py:18-24 + :46-48: weights from the checkpoint, every trunk parameter frozen, the 1x1 projection trainable, and the BatchNorm
    buffers after the zero-image probe equal to what the same probe leaves in a plain torch trunk holding the same weights."""
    from types import SimpleNamespace
    from sat_amd import encoder as E
    net, path = _torchvision_style_checkpoint(tmp_path, "resnet18", "resnet18-0123abcd.pth")
    args = SimpleNamespace(encoder_arch="resnet18", input_size=64, encoder_dim=32, encoder_size=None, mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])
    if how == "path":
        args.pretrained = path
    else:
        args.pretrained = True; monkeypatch.setenv("SAT_PRETRAINED_DIR", str(tmp_path))
    enc = E.get_encoder(args)
    trunk = torch.nn.Sequential(net.conv1, net.bn1, net.relu, net.maxpool, net.layer1, net.layer2, net.layer3, net.layer4).train()
    want = {k: v.clone() for k, v in net.state_dict().items() if not k.startswith("fc.")}
    trunk(torch.zeros(1, 3, 64, 64))                                           # the reference's probe (model.py:46-48)
    after = {k: v for k, v in net.state_dict().items() if not k.startswith("fc.")}
    index = {"conv1": "1", "bn1": "2", "layer1": "5", "layer2": "6", "layer3": "7", "layer4": "8"}
    got = enc.state_dict()
    moved = 0
    for k, v in after.items():
        head, _, rest = k.partition(".")
        g = got[index[head] + "." + rest]
        assert torch.allclose(g.float(), v.float(), rtol=1e-5, atol=1e-6), k
        moved += int(("running" in k) and not torch.equal(v, want[k]))
    assert moved > 30                                                           # the probe really changed the buffers
    frozen = [n for n, p in enc.named_parameters() if not p.requires_grad]
    free = [n for n, p in enc.named_parameters() if p.requires_grad]
    assert free == ["9.weight", "9.bias"] and len(frozen) == len(list(net.parameters())) - 2
    assert not enc.trunk_trainable


def test_pretrained_shufflenet_loads_a_local_checkpoint_and_probes_like_the_reference(tmp_path):
    """the same for the CLI's default arch (train.py:43, model.py:30-31): torchvision keys conv1 / stage2-4 / conv5, classifier dropped"""
    from types import SimpleNamespace
    from oracle import sat_oracle as O
    from sat_amd import encoder as E
    torch.manual_seed(11)
    net = O.ShuffleNetOracle("shufflenet_v2_x0_5")
    with torch.no_grad():
        for mod in net.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5); mod.bias.uniform_(-0.3, 0.3); mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.5, 2.0)
    path = os.path.join(str(tmp_path), "shufflenetv2_x0.5-f707e7126e.pth")
    torch.save(net.state_dict(), path)
    args = SimpleNamespace(encoder_arch="shufflenet_v2_x0_5", input_size=64, encoder_dim=48, encoder_size=None, mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225],
                           pretrained=path)
    enc = E.get_encoder(args)
    trunk = torch.nn.Sequential(net.conv1, net.maxpool, net.stage2, net.stage3, net.stage4, net.conv5).train()
    want = {k: v.clone() for k, v in net.state_dict().items() if not k.startswith("fc.")}
    trunk(torch.zeros(1, 3, 64, 64))
    index = {"conv1": "1", "stage2": "3", "stage3": "4", "stage4": "5", "conv5": "6"}
    got = enc.state_dict()
    moved = 0
    for k, v in net.state_dict().items():
        if k.startswith("fc."):
            continue
        head, _, rest = k.partition(".")
        assert torch.allclose(got[index[head] + "." + rest].float(), v.float(), rtol=1e-5, atol=1e-6), k
        moved += int(("running" in k) and not torch.equal(v, want[k]))
    assert moved > 60
    assert [n for n, p in enc.named_parameters() if p.requires_grad] == ["7.weight", "7.bias"] and not enc.trunk_trainable


def test_pretrained_mobilenet_v2_loads_a_local_checkpoint_and_probes_like_the_reference(tmp_path):
    """model.py:38-39: torchvision keys ``features.*``, classifier dropped; the probe moves the BatchNorm buffers as the reference's does"""
    from types import SimpleNamespace
    from oracle import sat_oracle as O
    from sat_amd import encoder as E
    torch.manual_seed(12)
    net = O.MobileNetV2Oracle()
    with torch.no_grad():
        for mod in net.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5); mod.bias.uniform_(-0.3, 0.3); mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.5, 2.0)
    path = os.path.join(str(tmp_path), "mobilenet_v2-b0353104.pth")
    torch.save(net.state_dict(), path)
    args = SimpleNamespace(encoder_arch="mobilenet_v2", input_size=64, encoder_dim=48, encoder_size=None, mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225],
                           pretrained=path)
    enc = E.get_encoder(args)
    want = {k: v.clone() for k, v in net.state_dict().items()}
    net.features.train()(torch.zeros(1, 3, 64, 64))
    got = enc.state_dict()
    moved = 0
    for k, v in net.state_dict().items():
        if k.startswith("classifier."):
            continue
        assert torch.allclose(got["1." + k[len("features."):]].float(), v.float(), rtol=1e-5, atol=1e-6), k
        moved += int(("running" in k) and not torch.equal(v, want[k]))
    assert moved > 60
    assert [n for n, p in enc.named_parameters() if p.requires_grad] == ["2.weight", "2.bias"] and not enc.trunk_trainable


def test_pretrained_without_a_local_file_says_so(tmp_path, monkeypatch):
    from types import SimpleNamespace
    from sat_amd import encoder as E
    monkeypatch.setenv("SAT_PRETRAINED_DIR", str(tmp_path)); monkeypatch.setenv("TORCH_HOME", str(tmp_path))
    args = SimpleNamespace(encoder_arch="resnet18", input_size=64, encoder_dim=None, encoder_size=None, mean=[0.5] * 3, std=[0.2] * 3, pretrained=True)
    with pytest.raises(RuntimeError, match="no local torchvision checkpoint"):
        E.get_encoder(args)
    args.pretrained = os.path.join(str(tmp_path), "nope.pth")
    with pytest.raises(FileNotFoundError):
        E.get_encoder(args)


def test_packing_plans_are_cached_by_caption_lengths_and_match_pack_padded_sequence():
    """model.py:553-554: the plan's index pair / batch sizes are what ``pack_padded_sequence(enforce_sorted=False)`` builds; plans are shared
    between batches with the same lengths (read-only) and rebuilt for different ones."""
    from sat_amd import decoder as Dk
    lens = torch.tensor([3, 5, 1, 5, 0, 2])
    a = Dk.PackPlan.cached(lens, 7, "cpu"); b = Dk.PackPlan.cached(lens.clone(), 7, "cpu"); c = Dk.PackPlan.cached(lens + 1, 7, "cpu")
    assert a is b and a is not c
    x = torch.arange(6 * 6, dtype=torch.float32).reshape(6, 6)
    keep = lens > 0                                        # pack_padded_sequence refuses zero lengths; the plan gives such rows no slot
    ref = torch.nn.utils.rnn.pack_padded_sequence(x[keep], lens[keep], batch_first=True, enforce_sorted=False)
    assert a.batch_sizes.tolist() == ref.batch_sizes.tolist()
    assert torch.equal(a.pack(x.unsqueeze(-1)).squeeze(-1), ref.data)
    assert torch.equal(a.sorted_indices_dev, a.sorted_indices) and torch.equal(a.unsorted_indices_dev[a.sorted_indices], torch.arange(6))


def test_precision_16_of_the_reference_cli_selects_the_bf16_mode():
    """train.py:31-32: ``--precision 16`` (torch AMP under Lightning) maps to the library's reduced-precision mode; 32 stays exact fp32"""
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    from oracle import sat_oracle as O
    kw = dict(vocab_size=23, encoder_dim=12, embed_dim=10, attention_dim=7, decoder_dim=9, input_size=64, encoder_arch="resnet18")
    assert M.SAT(**vars(O.default_hparams(**kw))).sat_precision == "fp32"
    assert M.SAT(**vars(O.default_hparams(precision=32, **kw))).sat_precision == "fp32"
    m16 = M.SAT(**vars(O.default_hparams(precision=16, **kw)))
    assert m16.sat_precision == "bf16" and m16.encoder.precision == "bf16"
    assert M.SAT(**vars(O.default_hparams(precision=16, hip_precision="fp32", **kw))).sat_precision == "fp32"
