import os, sys, torch
sys.path.insert(0, "/root/repo")
import sat_amd
from sat_amd import model as M
from oracle import sat_oracle as O, prng
torch.manual_seed(11)
net = O.ResNetOracle("resnet18")
path = "/tmp/resnet18-feedbeef.pth"; torch.save(net.state_dict(), path)
def make(fused):
    over = dict(encoder_arch="resnet18", encoder_dim=32, input_size=64, encoder_size=3, vocab_size=120, embed_dim=24, attention_dim=16, decoder_dim=40,
                deep_output=True, decoder_tf="always", encoder_finetune_after=2, weight_decay=0.0, decoder_lr=1e-3, embedding_lr=1e-2, encoder_lr=1e-4, opt="adam",
                adam_b1=0.9, adam_b2=0.999, momentum=0.9, nesterov=False, scheduler=None, fused_optimizer=fused)
    hp = O.default_hparams(**over); hp.pretrained = path
    torch.manual_seed(3)
    m = M.SAT(**vars(hp)).cuda().train()
    return m, hp
img = torch.from_numpy(prng.uniform((4, 3, 64, 64), 5, 0.0, 1.0)).cuda()
caps, lengths = prng.captions(4, 3, 9, 120, 6); caps = torch.from_numpy(caps).cuda(); lengths = torch.from_numpy(lengths)
res = {}
for fused in (False, True):
    m, hp = make(fused)
    opt = m.configure_optimizers()
    for it in range(5):
        opt.zero_grad(set_to_none=True)
        out = m.training_step((img, caps, lengths), it); out["loss"].backward(); opt.step()
        m.__dict__["_sat_global_step"] = it + 1
    res[fused] = {k: v.detach().clone() for k, v in m.state_dict().items()}
    print("fused", fused, "ok; trainable encoder params:", sum(p.requires_grad for p in m.encoder.parameters()))
worst = max(float((res[True][k].double() - res[False][k].double()).abs().max()) / max(1e-6, float(res[False][k].double().abs().max())) for k in res[True] if res[True][k].dtype.is_floating_point)
print("worst relative difference fused vs torch.optim.Adam after 5 steps:", worst)
