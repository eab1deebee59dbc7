"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (sat_amd/dist.py) averages gradients
exactly like one process on the concatenated batch, bucket by bucket, and parameters start identical.
Covers the persistent flat buckets (gradients written straight into their slices through ``_lib.grad_buffer`` and
adopted by autograd, or copied in by the hooks), the early launch from inside the encoder backward, gradient
accumulation (``no_sync``), parameters unfrozen after construction (encoder_finetune_after, reference
model.py:584-586) and the bf16 wire format."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _SinkLinear(torch.autograd.Function):
    """y = x @ w.T whose backward writes dw where the product's backward wrappers do: ``_lib.grad_buffer(w)``."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, g):
        from sat_amd import _lib as L
        x, w = ctx.saved_tensors
        dw = L.grad_buffer(w)
        torch.mm(g.t(), x, out=dw)
        return g @ w, dw


class Toy(torch.nn.Module):
    """Parameter names shaped like the product's (encoder.<idx>.* / decoder keys) so default_buckets applies."""

    def __init__(self):
        super().__init__()
        self.encoder = torch.nn.Sequential(*[torch.nn.Linear(6, 6) for _ in range(10)])
        self.embedding = torch.nn.Embedding(11, 6)
        self.sink = torch.nn.Parameter(torch.randn(6, 6) * 0.3)      # decoder-side parameter with a sink-writing backward
        self.output = torch.nn.Linear(6, 3)

    def forward(self, idx, x):
        h = self.encoder(x + self.embedding(idx))
        return self.output(_SinkLinear.apply(h, self.sink)).pow(2).mean()


def _data():
    g = torch.Generator().manual_seed(7)
    return torch.randint(0, 11, (8,), generator=g), torch.randn(8, 6, generator=g)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import sat_amd  # noqa: F401
    from sat_amd.dist import GradSync, broadcast_parameters, default_buckets
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                       # different init per rank on purpose
    model = Toy()
    broadcast_parameters(model)
    sync = GradSync(model)
    assert len(default_buckets(model)) == 5 and len(sync.buckets) == 5
    idx, x = _data()
    sl = slice(rank * 4, rank * 4 + 4)
    for step in range(2):                               # two steps: hooks re-arm, slices are re-used
        model.zero_grad(set_to_none=True)
        model(idx[sl], x[sl]).backward()
        sync.finish()
        for b in sync.buckets:                          # every gradient lives in its bucket slice: nothing to copy back
            assert all(b.owns(p) for p in b.params)
    ref_grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    res = {"plain": ref_grads}

    # early path: the encoder backward announces a finished stage whose gradients already sit in the bucket slices
    model.zero_grad(set_to_none=True)
    loss = model(idx[sl], x[sl])
    stage = sync.buckets[1]                              # the "projection + layer4" bucket
    early = {}
    for p, g in zip(stage.params, torch.autograd.grad(loss, stage.params, retain_graph=True)):
        v = stage.view(p); v.copy_(g); early[p] = v
    sync.encoder_stage_ready(early)
    assert stage.launched
    loss.backward()
    sync.finish()
    for k, p in model.named_parameters():
        assert torch.allclose(p.grad, ref_grads[k], atol=1e-7, rtol=1e-6), k
    assert not any(b.launched for b in sync.buckets)

    # accumulation: two micro-batches of two samples; only the second one exchanges
    model.zero_grad(set_to_none=True)
    a, b_ = slice(rank * 4, rank * 4 + 2), slice(rank * 4 + 2, rank * 4 + 4)
    with sync.no_sync():
        (0.5 * model(idx[a], x[a])).backward()
        sync.encoder_stage_ready({p: p.grad for p in stage.params})       # must not start anything
        assert not any(bk.launched for bk in sync.buckets)
    (0.5 * model(idx[b_], x[b_])).backward()
    sync.finish()
    res["accumulate"] = {k: p.grad.clone() for k, p in model.named_parameters()}

    # bf16 on the wire
    sync16 = None
    sync.remove()
    sync16 = GradSync(model, bucket_dtype=torch.bfloat16)
    model.zero_grad(set_to_none=True)
    model(idx[sl], x[sl]).backward()
    sync16.finish()
    res["bf16"] = {k: p.grad.clone() for k, p in model.named_parameters()}
    sync16.remove()

    # frozen encoder, unfrozen later (model.py:584-586): the buckets are laid out again
    for p in model.encoder.parameters():
        p.requires_grad = False
    sync = GradSync(model)
    assert len(sync.buckets) == 1
    model.zero_grad(set_to_none=True)
    model(idx[sl], x[sl]).backward()
    sync.finish()
    assert all(p.grad is None for p in model.encoder.parameters())
    for p in model.encoder.parameters():
        p.requires_grad = True
    model.zero_grad(set_to_none=True)
    dec_bucket = sync.buckets[0]
    model(idx[sl], x[sl]).backward()
    # the decoder bucket's all-reduce was started by this step's hooks on the OLD layout.  Let it complete before finish() re-lays the
    # buckets out (the case that used to reduce the decoder gradients twice: world x mean) ...
    assert dec_bucket.launched and dec_bucket.work is not None
    dec_bucket.work.wait()
    sync.finish()
    assert len(sync.buckets) == 5
    assert sync.buckets[0] is dec_bucket                 # ... the unchanged bucket is kept, with its buffer and its collective
    res["unfrozen"] = {k: p.grad.clone() for k, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)                    # and the step after runs through the hooks again
    model(idx[sl], x[sl]).backward()
    sync.finish()
    for k, p in model.named_parameters():
        assert torch.allclose(p.grad, res["unfrozen"][k], atol=1e-7, rtol=1e-6), k

    torch.save(res, out % rank)
    torch.save({k: p.detach().clone() for k, p in model.named_parameters()}, (out % rank) + ".w")
    dist.destroy_process_group()


def test_two_rank_gradient_mean_matches_single_process(tmp_path):
    out = str(tmp_path / "grads_%d.pt")
    mp.spawn(_worker, args=(2, 29533, out), nprocs=2, join=True)
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    w0, w1 = torch.load((out % 0) + ".w"), torch.load((out % 1) + ".w")
    sys.path.insert(0, ROOT)
    model = Toy()
    model.load_state_dict(w0)
    for k in w0:
        assert torch.equal(w0[k], w1[k]), k                 # broadcast made the replicas identical
    idx, x = _data()
    (0.5 * (model(idx[:4], x[:4]) + model(idx[4:], x[4:]))).backward()
    for case in ("plain", "accumulate", "bf16", "unfrozen"):
        g0, g1 = r0[case], r1[case]
        for k, p in model.named_parameters():
            assert torch.equal(g0[k], g1[k]), (case, k)         # every rank ends with the same gradient
            if case == "bf16":
                assert torch.allclose(g0[k], p.grad, atol=2e-3 * p.grad.abs().max().item() + 1e-6, rtol=1e-2), (case, k)
            else:
                assert torch.allclose(g0[k], p.grad, atol=1e-7, rtol=1e-5), (case, k)


def test_one_process_gradsync_leaves_unused_parameters_without_a_gradient():
    """world size 1: finish() must not invent zero gradients (Adam would decay its moments and apply weight decay on them); torch and the
    reference skip parameters whose grad is None."""
    sys.path.insert(0, ROOT)
    import sat_amd  # noqa: F401
    from sat_amd.dist import GradSync
    torch.manual_seed(3)
    model = Toy()
    idx, x = _data()
    unused = torch.nn.Parameter(torch.zeros(3))
    model.unused = unused
    sync2 = GradSync(model)
    model.zero_grad(set_to_none=True)
    model(idx, x).backward()
    sync2.finish()
    assert unused.grad is None
    for k, p in model.named_parameters():
        if p is not unused:
            assert p.grad is not None, k
            assert sync2._of[id(p)].owns(p), k          # produced gradients live in their slices (addresses repeat from step to step)
    sync2.remove()


def test_a_gradient_slice_is_handed_out_once_per_backward():
    """two backward nodes of one parameter inside ONE backward pass: the second must not be given the same bucket slice (it would overwrite
    the first node's gradient and autograd would add the slice to itself: 2 x g(b) instead of g(a) + g(b))"""
    sys.path.insert(0, ROOT)
    import sat_amd  # noqa: F401
    from sat_amd.dist import GradSync
    torch.manual_seed(4)
    model = Toy()
    idx, x = _data()
    ref = Toy(); ref.load_state_dict(model.state_dict())
    (ref(idx[:4], x[:4]) + ref(idx[4:], x[4:])).backward()
    sync = GradSync(model)
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        (model(idx[:4], x[:4]) + model(idx[4:], x[4:])).backward()
        sync.finish()
        assert torch.allclose(model.sink.grad, ref.sink.grad, atol=1e-7, rtol=1e-6)
        for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            assert torch.allclose(p.grad, q.grad, atol=1e-7, rtol=1e-6), k
    sync.remove()


def test_shufflenet_encoder_is_one_bucket():
    """the CLI's default encoder (0.34 M parameters) is exchanged as one bucket behind the decoder's; the ResNets keep one bucket per stage"""
    import sat_amd  # noqa: F401
    from oracle import sat_oracle as O
    from sat_amd import model as M
    from sat_amd.dist import default_buckets
    kw = dict(input_size=64, vocab_size=50, embed_dim=16, attention_dim=8, decoder_dim=24, weight_decay=0.0, decoder_lr=1e-3, embedding_lr=1e-2,
              encoder_lr=1e-4, opt="adam", adam_b1=0.9, adam_b2=0.999, momentum=0.9, nesterov=False, scheduler=None)
    shuffle = M.SAT(**vars(O.default_hparams(encoder_arch="shufflenet_v2_x0_5", encoder_dim=None, **kw)))
    buckets = default_buckets(shuffle)
    assert len(buckets) == 2
    assert sum(len(b) for b in buckets) == len(list(shuffle.parameters()))
    assert {id(p) for p in buckets[1]} == {id(p) for p in shuffle.encoder.parameters()}
    resnet = M.SAT(**vars(O.default_hparams(encoder_arch="resnet18", encoder_dim=32, **kw)))
    assert len(default_buckets(resnet)) == 5
