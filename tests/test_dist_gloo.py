"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (sat_amd/dist.py) averages gradients
exactly like one process on the concatenated batch, bucket by bucket, and parameters start identical."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Toy(torch.nn.Module):
    """Parameter names shaped like the product's (encoder.<idx>.* / decoder keys) so default_buckets applies."""

    def __init__(self):
        super().__init__()
        self.encoder = torch.nn.Sequential(*[torch.nn.Linear(6, 6) for _ in range(10)])
        self.embedding = torch.nn.Embedding(11, 6)
        self.output = torch.nn.Linear(6, 3)

    def forward(self, idx, x):
        return self.output(self.encoder(x + self.embedding(idx))).pow(2).mean()


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
    g = torch.Generator().manual_seed(7)
    idx = torch.randint(0, 11, (8,), generator=g); x = torch.randn(8, 6, generator=g)
    sl = slice(rank * 4, rank * 4 + 4)
    for step in range(2):                               # two steps: hooks re-arm
        model.zero_grad(set_to_none=True)
        model(idx[sl], x[sl]).backward()
        sync.finish()
    # early path: the encoder backward announces a finished stage before autograd has accumulated anything
    ref_grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    loss = model(idx[sl], x[sl])
    stage = sync.buckets[1]                              # the "projection + layer4" bucket
    sync.encoder_stage_ready(dict(zip(stage, torch.autograd.grad(loss, stage, retain_graph=True))))
    assert 1 in sync._early
    loss.backward()
    sync.finish()
    for k, p in model.named_parameters():
        assert torch.allclose(p.grad, ref_grads[k], atol=1e-7, rtol=1e-6), k
    assert not sync._early and not sync._inflight
    torch.save({k: p.grad.clone() for k, p in model.named_parameters()}, out % rank)
    torch.save({k: p.detach().clone() for k, p in model.named_parameters()}, (out % rank) + ".w")
    dist.destroy_process_group()


def test_two_rank_gradient_mean_matches_single_process(tmp_path):
    out = str(tmp_path / "grads_%d.pt")
    mp.spawn(_worker, args=(2, 29533, out), nprocs=2, join=True)
    g0, g1 = torch.load(out % 0), torch.load(out % 1)
    w0, w1 = torch.load((out % 0) + ".w"), torch.load((out % 1) + ".w")
    model = Toy()
    model.load_state_dict(w0)
    for k in w0:
        assert torch.equal(w0[k], w1[k]), k                 # broadcast made the replicas identical
    g = torch.Generator().manual_seed(7)
    idx = torch.randint(0, 11, (8,), generator=g); x = torch.randn(8, 6, generator=g)
    (0.5 * (model(idx[:4], x[:4]) + model(idx[4:], x[4:]))).backward()
    for k, p in model.named_parameters():
        assert torch.equal(g0[k], g1[k]), k                 # every rank ends with the same gradient
        assert torch.allclose(g0[k], p.grad, atol=1e-7, rtol=1e-5), k
