"""Dev tool: is the fp32 HIP encoder forward/backward reproducible run to run (same process, different allocator state)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sat_amd  # noqa
from sat_amd import encoder as E
from oracle import prng, sat_oracle as O
arch, px, es = "resnet18", 64, 3
torch.manual_seed(3)
enc = E.get_encoder(O.default_hparams(encoder_arch=arch, encoder_dim=32, input_size=px, encoder_size=es)).cuda().train()
img = torch.from_numpy(prng.uniform((8, 3, px, px), 77, 0.0, 1.0)).cuda()
outs = []
for rep in range(3):
    junk = torch.full((1 << 22,), float(rep + 1) * 1e9, device="cuda"); del junk     # poison freed memory differently each time
    enc.zero_grad(set_to_none=True)
    y = enc(img)
    dy = torch.from_numpy(prng.uniform(tuple(y.shape), 78)).cuda()
    y.backward(dy)
    outs.append((y.detach().clone(), {k: p.grad.clone() for k, p in enc.named_parameters()}))
for rep in (1, 2):
    print("rep", rep, "fwd equal:", torch.equal(outs[0][0], outs[rep][0]))
    bad = [k for k in outs[0][1] if not torch.equal(outs[0][1][k], outs[rep][1][k])]
    print("   grads differing:", len(bad), bad[:6])
    for k in bad[:4]:
        a, b = outs[0][1][k], outs[rep][1][k]
        print("     ", k, float((a - b).abs().max()), float(a.abs().max()))
