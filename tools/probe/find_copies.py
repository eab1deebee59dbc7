import os, sys
sys.path.insert(0, "/root/repo")
import torch, bench, sat_amd
from sat_amd import model as M
from collections import Counter
import traceback
cfg = os.environ.get("CFG", "cli")
hp, T, B, R = bench.hparams(cfg)
if cfg == "cli":
    hp["decoder_tf"] = None; hp["deep_output"] = False
torch.manual_seed(42)
model = M.SAT(**hp).cuda().train(); model.set_precision("bf16"); model.__dict__["_sat_global_step"] = 2
opt = model.configure_optimizers()
img, caps, lengths = bench.synthetic_batch(B, R, T, hp["vocab_size"], 1234, False, px=hp["input_size"])
img, caps = img.cuda(), caps.cuda()
def step():
    opt.zero_grad(set_to_none=True)
    out = model.training_step((img, caps, lengths), 0); out["loss"].backward(); opt.step()
for _ in range(3): step()
torch.autograd.set_multithreading_enabled(False)
sites = Counter()
orig_copy, orig_clone, orig_contig = torch.Tensor.copy_, torch.Tensor.clone, torch.Tensor.contiguous
def where():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "sat_amd" in fr.filename or "show-attend" in fr.filename or "bench" in fr.filename:
            return "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
    return "?"
def copy_(self, src, *a, **k):
    if self.is_cuda: sites["copy_ " + where()] += 1
    return orig_copy(self, src, *a, **k)
def clone(self, *a, **k):
    if self.is_cuda: sites["clone " + where()] += 1
    return orig_clone(self, *a, **k)
def contiguous(self, *a, **k):
    if self.is_cuda and not self.is_contiguous(*a, **k): sites["contiguous " + where()] += 1
    return orig_contig(self, *a, **k)
torch.Tensor.copy_, torch.Tensor.clone, torch.Tensor.contiguous = copy_, clone, contiguous
step()
torch.Tensor.copy_, torch.Tensor.clone, torch.Tensor.contiguous = orig_copy, orig_clone, orig_contig
for k, v in sites.most_common(30): print(v, k)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
c = Counter()
for e in prof.events():
    if "emcpy" in e.name or "copy" in e.name.lower():
        c[e.name] += 1
for k, v in c.most_common(12): print(v, k[:100])
