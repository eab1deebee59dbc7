"""Dev tool: per-shape roofline table of the GEMM / conv kernels inside one C2 train step.
SAT_PROFILE_SHAPES=1 makes the in-library profiler key its entries by problem shape."""
import os, sys
os.environ["SAT_PROFILE_SHAPES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import sat_amd  # noqa
from sat_amd import _lib, model as M

hp, T, B, R = bench.hparams("c2")
if os.environ.get("TF") == "none":
    hp["decoder_tf"] = None
torch.manual_seed(42)
model = M.SAT(**hp).cuda().train(); model.set_precision("bf16")
model.__dict__["_sat_global_step"] = 2
opt = model.configure_optimizers()
img, caps, lengths = bench.synthetic_batch(B, R, T, hp["vocab_size"], 1234, False)
img, caps = img.cuda(), caps.cuda()

def step():
    opt.zero_grad(set_to_none=True)
    out = model.training_step((img, caps, lengths), 0); out["loss"].backward(); opt.step()

for _ in range(4): step()
torch.cuda.synchronize()
_lib.profile_start()
n = 4
for _ in range(n): step()
torch.cuda.synchronize()
ent = sorted(_lib.profile_stop(max_entries=1024), key=lambda e: -e["total_ms"])
tot = 0.0
print("%-62s %4s %8s %7s %7s %7s %7s" % ("kernel shape", "n", "us/call", "TF", "GB/s", "hbm_us", "mfma_us"))
for e in ent:
    if not e["name"].startswith("gemm"): continue
    us = e["total_ms"] * 1e3 / e["launches"]; fl = e["flops"] / e["launches"]; by = e["bytes"] / e["launches"]
    tot += e["total_ms"] / n
    print("%-62s %4d %8.1f %7.1f %7.0f %7.1f %7.1f" % (e["name"], e["launches"] // n, us, fl / us / 1e6, by / us / 1e3, by / 4.5e6, fl / 2.5e9))
print("GEMM total per step: %.2f ms" % tot)
if os.environ.get("ALL") == "1":          # every other instrumented family
    for e in ent:
        if e["name"].startswith("gemm"): continue
        print("%-62s %4d %8.1f us  %7.3f ms/step" % (e["name"], e["launches"] // n, e["total_ms"] * 1e3 / e["launches"], e["total_ms"] / n))
