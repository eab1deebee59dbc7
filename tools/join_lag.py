"""Dev tool: how far the side stream (encoder weight gradients) lags behind the main stream at each join of one C2 backward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import sat_amd  # noqa
from sat_amd import model as M, encoder as E
hp, T, B, R = bench.hparams("c2")
torch.manual_seed(42)
model = M.SAT(**hp).cuda().train(); model.set_precision("bf16")
model.__dict__["_sat_global_step"] = 2
opt = model.configure_optimizers()
img, caps, lengths = bench.synthetic_batch(B, R, T, hp["vocab_size"], 1234, False)
img, caps = img.cuda(), caps.cuda()
def step():
    opt.zero_grad(set_to_none=True)
    out = model.training_step((img, caps, lengths), 0); out["loss"].backward(); opt.step()
for _ in range(30): step()
torch.cuda.synchronize()
for rep in range(3):
    E._JOIN_LAG = []
    step(); torch.cuda.synchronize()
    print("step %d: side stream finished after the main stream reached the join by (ms):" % rep, ["%.3f" % em.elapsed_time(ev) for em, ev in E._JOIN_LAG])
E._JOIN_LAG = None
