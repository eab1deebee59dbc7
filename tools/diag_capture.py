"""Dev tool: which part of the train step survives hipGraph capture.  usage: python tools/diag_capture.py <stage> [side=0|1]
stages: enc_fwd, fwd (encoder + decoder + losses), fwd_bwd, step (fwd + bwd + optimizer)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sat_amd  # noqa
from sat_amd import model as M, encoder as E
from oracle import sat_oracle as O, prng

stage = sys.argv[1]
E._WGRAD_STREAM = (len(sys.argv) > 2 and sys.argv[2] == "side=1")
kw = dict(encoder_arch="resnet18", encoder_dim=64, input_size=64, encoder_size=3, vocab_size=120, embed_dim=24, attention_dim=16,
          decoder_dim=40, deep_output=True, weight_decay=0.0, decoder_lr=1e-3, embedding_lr=1e-2, encoder_lr=1e-4, opt="adam",
          adam_b1=0.9, adam_b2=0.999, momentum=0.9, nesterov=False, scheduler=None, decoder_tf="always", encoder_finetune_after=1)
hp = O.default_hparams(**kw)
torch.manual_seed(1)
model = M.SAT(**vars(hp)).cuda().train(); model.set_precision("bf16")
opt = model.configure_optimizers()
img = torch.from_numpy(prng.uniform((4, 3, 64, 64), 3, 0.0, 1.0)).cuda()
caps, lengths = prng.captions(4, 3, 9, hp.vocab_size, 4)
caps, lengths = torch.from_numpy(caps).cuda(), torch.from_numpy(lengths)


def body():
    if stage == "enc_fwd":
        with torch.no_grad():
            return model.encoder(img)
    if stage == "fwd":
        with torch.no_grad():
            return model._step_losses((img, caps, lengths), 1.0)
    if stage == "dec":          # decoder + losses forward and backward from a leaf annotation tensor (no encoder backward)
        res = model.train_decode(ann_leaf, caps, lengths, 1.0)
        loss = res["ce"] + res["ds"]
        loss.backward()
        return loss.detach()
    if stage == "dec_nolos":    # decoder forward / backward only, gradient fed from outside
        res = model.train_decode(ann_leaf, caps, lengths, 1.0, with_loss=False)
        torch.autograd.backward([res["logits_packed"], res["alphas"]], [torch.ones_like(res["logits_packed"]), torch.ones_like(res["alphas"])])
        return ann_leaf.grad.sum()
    if stage == "enc":          # encoder forward and backward
        ann = model.encoder(img)
        ann.backward(gann)
        return ann.detach().sum()
    loss = model._step_losses((img, caps, lengths), 1.0)
    loss.backward()
    if stage == "step":
        opt.step_device()
    return loss.detach()


for _ in range(2):          # eager warm-up (every cache, kernel attribute, optimizer table)
    opt.zero_grad(set_to_none=True)
    l = model._step_losses((img, caps, lengths), 1.0); l.backward(); opt.step()
del l
opt.prepare_device_step(img.device); opt.step_host()
torch.cuda.synchronize()
opt.zero_grad(set_to_none=True)
with torch.no_grad():
    a0 = model.encoder(img)
    Bq, Dq, hq, wq = a0.shape
    ann_leaf = a0.permute(0, 2, 3, 1).reshape(Bq, hq * wq, Dq).clone().requires_grad_()
    gann = torch.ones_like(a0)
if stage in ("dec", "dec_nolos", "enc"):
    _o = body(); del _o; torch.cuda.synchronize(); opt.zero_grad(set_to_none=True); ann_leaf.grad = None
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
if os.environ.get("ST") == "1":
    torch.autograd.set_multithreading_enabled(False)
print("capturing", stage, "side stream", E._WGRAD_STREAM, flush=True)
with torch.cuda.graph(g, stream=s):
    out = body()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize()
print("replayed:", float(out.float().sum()), flush=True)
