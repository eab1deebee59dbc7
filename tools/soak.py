"""Dev: a few hundred C2 train steps on fixed synthetic data - loss must fall, memory must stay flat, nothing may turn NaN."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import sat_amd  # noqa
from sat_amd import model as M

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = sys.argv[2] if len(sys.argv) > 2 else "c2"
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
hp, T, B, R = bench.hparams(cfg)
if len(sys.argv) > 4 and sys.argv[4] == "none":
    hp["decoder_tf"] = None
print("soak", cfg, prec, "decoder_tf", hp["decoder_tf"], flush=True)
torch.manual_seed(42)
model = M.SAT(**hp).cuda().train(); model.set_precision(prec)
model.__dict__["_sat_global_step"] = 2
opt = model.configure_optimizers()
img, caps, lengths = bench.synthetic_batch(B, R, T, hp["vocab_size"], 1234, True)
img, caps = img.cuda(), caps.cuda()
t0 = time.perf_counter()
for i in range(steps):
    opt.zero_grad(set_to_none=True)
    out = model.training_step((img, caps, lengths), 0)
    out["loss"].backward()
    opt.step()
    if not all(bool(torch.isfinite(p).all()) for p in model.parameters()):
        print("first non-finite parameter seen at step", i, [k for k, p in model.named_parameters() if not bool(torch.isfinite(p).all())][:5]); break
    if i % 50 == 0 or i == steps - 1:
        torch.cuda.synchronize()
        print("step %4d loss %.4f acc %.3f  allocated %.2f GB  reserved %.2f GB  %.1f s" % (
            i, float(out["loss"]), float(out["accuracy"]), torch.cuda.memory_allocated() / 2 ** 30, torch.cuda.memory_reserved() / 2 ** 30,
            time.perf_counter() - t0), flush=True)
bad = [(k, int((~torch.isfinite(p)).sum()), p.numel()) for k, p in model.named_parameters() if not bool(torch.isfinite(p).all())]
print("non-finite parameters:", bad)
badg = [(k, int((~torch.isfinite(p.grad)).sum())) for k, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
print("non-finite gradients:", badg)
for k, p in model.named_parameters():
    if not bool(torch.isfinite(p).all()):
        idx = (~torch.isfinite(p)).nonzero()[:5].tolist()
        st = opt.state.get(p, {})
        print(k, tuple(p.shape), "first bad", idx, {a: (float(b.flatten()[0]) if torch.is_tensor(b) else b) for a, b in st.items() if a != "step"})
        break
