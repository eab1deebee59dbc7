"""GPU parity of the multi-tensor optimizer step (csrc/optimizer.hip) against torch.optim run on the CPU
with the same parameter groups, gradients and clipping (train.py:93-96, 273-274; model.py:723-757)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _groups(ps):
    return [{"params": ps[:2], "lr": 1e-2, "weight_decay": 0.0}, {"params": ps[2:4], "lr": 3e-3, "weight_decay": 0.05},
            {"params": ps[4:], "lr": 1e-3, "weight_decay": 0.0}]


@pytest.mark.parametrize("kind,clip", [("adam", None), ("adam", ("norm", 0.5)), ("adamw", ("value", 0.02)), ("sgd", None), ("sgd", ("norm", 1.0))])
def test_fused_optimizer_matches_torch(kind, clip):
    import sat_amd  # noqa: F401
    from sat_amd.optim import FusedOptimizer
    g = torch.Generator().manual_seed(7)
    shapes = [(5,), (70001,), (33, 17), (200, 300), (64, 3, 3, 8), (1,), (131073,)]          # chunk boundaries at 65536 elements
    ref = [torch.randn(s, generator=g).requires_grad_() for s in shapes]
    dev = [p.detach().clone().cuda().requires_grad_() for p in ref]
    if kind == "sgd":
        topt = torch.optim.SGD(_groups(ref), lr=1e-2, momentum=0.9, nesterov=True)
    elif kind == "adam":
        topt = torch.optim.Adam(_groups(ref), lr=1e-2, betas=(0.8, 0.95))
    else:
        topt = torch.optim.AdamW(_groups(ref), lr=1e-2, betas=(0.8, 0.95))
    fopt = FusedOptimizer(_groups(dev), kind=kind, lr=1e-2, betas=(0.8, 0.95), momentum=0.9 if kind == "sgd" else 0.0, nesterov=kind == "sgd",
                          grad_clip=clip[0] if clip else None, clip_value=clip[1] if clip else 0.0)
    sched_t = torch.optim.lr_scheduler.ExponentialLR(topt, gamma=0.7)            # the reference's schedulers drive param_groups
    sched_f = torch.optim.lr_scheduler.ExponentialLR(fopt, gamma=0.7)
    for step in range(4):
        for p, q in zip(ref, dev):
            gr = torch.randn(p.shape, generator=g) * (0.1 if step % 2 else 1.0)
            p.grad = gr.clone(); q.grad = gr.cuda()
        if clip and clip[0] == "norm":
            norm = torch.nn.utils.clip_grad_norm_(ref, clip[1])
        elif clip:
            torch.nn.utils.clip_grad_value_(ref, clip[1])
        topt.step(); fopt.step()
        if clip and clip[0] == "norm":
            assert abs(float(fopt.last_grad_norm) - float(norm)) <= 1e-5 * float(norm)
        sched_t.step(); sched_f.step()
        for i, (p, q) in enumerate(zip(ref, dev)):
            err = float((q.detach().cpu() - p.detach()).abs().max())
            assert err <= 2e-6 * max(1.0, float(p.detach().abs().max())), (kind, clip, step, i, err)
    # state keys are torch's: checkpoints stay interchangeable
    st = fopt.state[dev[1]]
    assert set(st) == ({"step", "momentum_buffer"} if kind == "sgd" else {"step", "exp_avg", "exp_avg_sq"})
    tst = topt.state[ref[1]]
    key = "momentum_buffer" if kind == "sgd" else "exp_avg"
    assert float((st[key].cpu() - tst[key]).abs().max()) <= 1e-6 and float(st["step"]) == 4


def test_fused_optimizer_is_bitwise_reproducible_and_rejects_cpu():
    import sat_amd  # noqa: F401
    from sat_amd import _lib
    from sat_amd.optim import FusedOptimizer
    outs = []
    for _ in range(2):
        g = torch.Generator().manual_seed(3)
        ps = [torch.randn(100000, generator=g).cuda().requires_grad_(), torch.randn(257, 129, generator=g).cuda().requires_grad_()]
        opt = FusedOptimizer(ps, kind="adam", lr=1e-3, grad_clip="norm", clip_value=0.3)
        for _ in range(3):
            for p in ps:
                p.grad = torch.randn(p.shape, generator=g).cuda()
            opt.step()
        outs.append([p.detach().clone() for p in ps])
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    cpu = [torch.zeros(4, requires_grad=True)]
    cpu[0].grad = torch.ones(4)
    with pytest.raises(_lib.SatHipError):
        FusedOptimizer(cpu, kind="adam").step()


def test_pointer_table_survives_a_host_that_runs_ahead():
    """The pointer / learning-rate table is rebuilt on the host every step and uploaded asynchronously.  With no host
    synchronisation between steps the host runs steps ahead of the GPU: alternate two sets of gradient buffers and the learning
    rate every step, keep the GPU busy so that the uploads lag, and compare with torch.optim on the CPU."""
    import sat_amd  # noqa: F401
    from sat_amd.optim import FusedOptimizer
    g = torch.Generator().manual_seed(11)
    shapes = [(4099,), (300, 70), (65537,)]
    ref = [torch.randn(s, generator=g).requires_grad_() for s in shapes]
    dev = [p.detach().clone().cuda().requires_grad_() for p in ref]
    topt = torch.optim.Adam(ref, lr=1e-2)
    fopt = FusedOptimizer(dev, kind="adam", lr=1e-2)
    steps = 12
    grads = [[torch.randn(s, generator=g) for s in shapes] for _ in range(steps)]
    sets = [[torch.empty(s, device="cuda") for s in shapes] for _ in range(2)]
    staged = [[gr.cuda() for gr in gs] for gs in grads]
    ballast = torch.randn(4096, 4096, device="cuda")
    torch.cuda.synchronize()
    for step in range(steps):
        lr = 1e-2 if step % 2 == 0 else 3e-3
        for pg in topt.param_groups:
            pg["lr"] = lr
        for pg in fopt.param_groups:
            pg["lr"] = lr
        for _ in range(6):                       # ~ms of queued GPU work: the optimizer's upload executes long after the host moved on
            ballast = (ballast @ ballast).clamp_(-1, 1)
        bufs = sets[step % 2]
        for q, b, src in zip(dev, bufs, staged[step]):
            b.copy_(src); q.grad = b
        for p, gr in zip(ref, grads[step]):
            p.grad = gr.clone()
        topt.step(); fopt.step()                 # no synchronisation anywhere in the loop
    torch.cuda.synchronize()
    for i, (p, q) in enumerate(zip(ref, dev)):
        err = float((q.detach().cpu() - p.detach()).abs().max())
        assert err <= 5e-6 * max(1.0, float(p.detach().abs().max())), (i, err)


def test_fast_path_follows_replaced_bf16_copies_and_state_tensors():
    """Constant learning rate and persistent gradient buffers put FusedOptimizer on its fast path (device table reused).  The table also holds
    the addresses of the bf16 filter copies the encoder reads and of the moment tensors: a copy remade behind the optimizer's back (the
    encoder does that when ``p._version`` moved), a copy that appears later (set_precision fp32 -> bf16 mid-run) and a replaced moment tensor
    must all take the slow path once, not be written through the stale address."""
    import sat_amd  # noqa: F401
    from sat_amd.optim import FusedOptimizer
    g = torch.Generator().manual_seed(5)
    ref = [torch.randn(64, 8, 3, 3, generator=g).requires_grad_(), torch.randn(300, generator=g).requires_grad_()]
    dev = [p.detach().clone().cuda().requires_grad_() for p in ref]
    topt = torch.optim.Adam(ref, lr=1e-2)
    fopt = FusedOptimizer(dev, kind="adam", lr=1e-2)
    bufs = [torch.empty_like(q) for q in dev]
    keep = []

    def step(i):
        for p, q, b in zip(ref, dev, bufs):
            gr = torch.randn(p.shape, generator=g)
            p.grad = gr.clone(); b.copy_(gr.cuda()); q.grad = b
        topt.step(); fopt.step()

    def shadow_ok():
        sh = dev[0]._sat_bf16_shadow
        assert torch.equal(sh, dev[0].detach().to(torch.bfloat16)), "the bf16 copy the encoder reads is not the rounded master weight"

    step(0); step(1)
    n_fast = getattr(fopt, "fast_path_steps", 0)
    assert n_fast >= 1                                   # the second step already reused the table
    # (b) a copy that did not exist when the table was built
    dev[0]._sat_bf16_shadow = dev[0].detach().to(torch.bfloat16); dev[0]._sat_shadow_version = dev[0]._version
    step(2); shadow_ok()
    step(3); shadow_ok()
    # (a) the copy remade after an in-place change of the weight (what encoder.py does when p._version moved); the old copy stays allocated
    # here so that the new one has a different address, and is poisoned: a stale table would update the poisoned one
    with torch.no_grad():
        dev[0].mul_(0.5); ref[0].mul_(0.5)
    keep.append(dev[0]._sat_bf16_shadow); keep[-1].fill_(float("nan"))
    dev[0]._sat_bf16_shadow = dev[0].detach().to(torch.bfloat16); dev[0]._sat_shadow_version = dev[0]._version
    step(4); shadow_ok()
    assert torch.isnan(keep[-1].float()).all()           # nobody wrote through the old address
    # a moment tensor replaced from outside
    st = fopt.state[dev[1]]
    old = st["exp_avg"]; st["exp_avg"] = old.clone(); keep.append(old); old.fill_(float("nan"))
    step(5); step(6)
    torch.cuda.synchronize()
    assert torch.isnan(old).all()
    for i, (p, q) in enumerate(zip(ref, dev)):
        err = float((q.detach().cpu() - p.detach()).abs().max())
        assert err <= 5e-6 * max(1.0, float(p.detach().abs().max())), (i, err)
    assert fopt.fast_path_steps > n_fast                 # and the fast path is taken again once things repeat
