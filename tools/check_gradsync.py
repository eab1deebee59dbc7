"""Dev: the real SAT model under GradSync with several ranks on ONE GPU (gloo): the synchronised gradients must equal the mean
of the ranks' local gradients, bucket by bucket, including the stages announced from inside the encoder backward.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29561 tools/check_gradsync.py      (ARCH=shufflenet_v2_x0_5 | mobilenet_v2 for the other encoder families)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import sat_amd  # noqa
from sat_amd import model as M
from sat_amd.dist import GradSync, broadcast_parameters
from oracle import prng, sat_oracle as O

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
over = dict(encoder_arch=os.environ.get("ARCH", "resnet18"), encoder_dim=32, input_size=64, encoder_size=3, vocab_size=120, embed_dim=24, attention_dim=16, decoder_dim=40,
            deep_output=True, decoder_tf="always", weight_decay=0.0, decoder_lr=1e-3, embedding_lr=1e-2, encoder_lr=1e-5, opt="adam", adam_b1=0.9,
            adam_b2=0.999, momentum=0.9, nesterov=False, scheduler=None)
for prec in ("fp32", "bf16"):
    hp = O.default_hparams(**over)
    torch.manual_seed(42 + rank)                     # different init per rank: broadcast must fix it
    model = M.SAT(**vars(hp)).cuda().train(); model.set_precision(prec)
    model.__dict__["_sat_global_step"] = 2
    broadcast_parameters(model)
    img = torch.from_numpy(prng.uniform((4, 3, 64, 64), 10 + rank, 0.0, 1.0)).cuda()
    caps, lengths = prng.captions(4, 3, 9, 120, 20 + rank)
    caps, lengths = torch.from_numpy(caps).cuda(), torch.from_numpy(lengths)
    # local gradients (no exchange)
    model.zero_grad(set_to_none=True)
    model.training_step((img, caps, lengths), 0)["loss"].backward()
    local = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    want = {}
    for k, g in local.items():
        buf = [torch.empty_like(g) for _ in range(world)]
        dist.all_gather(buf, g)
        want[k] = sum(buf) / world
    # the same step under GradSync
    sync = GradSync(model)
    model.zero_grad(set_to_none=True)
    model.training_step((img, caps, lengths), 0)["loss"].backward()
    early = sorted(sync._early)
    sync.finish()
    worst = max(float((p.grad - want[k]).abs().max()) / max(1e-12, float(want[k].abs().max())) for k, p in model.named_parameters() if k in want)
    print("rank %d %s: %d tensors, buckets launched from inside the encoder backward: %s, worst relative difference to the mean of local gradients %.2e"
          % (rank, prec, len(want), early, worst), flush=True)
    assert worst <= 1e-6, worst
    sync.remove()
dist.barrier()
dist.destroy_process_group()
