"""Dev tool: event-timed attention kernels of one config's train step (the in-library profiler), + the step time.  CFG=c4|c2|c3|c1"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import sat_amd  # noqa
from sat_amd import _lib
cfg = os.environ.get("CFG", "c4")
dev = torch.device("cuda", 0)
model, opt, sync, batch, (hp, T, B, R) = bench._build_train(cfg, dev)
def step():
    opt.zero_grad(set_to_none=True)
    out = model.training_step(batch, 0); out["loss"].backward(); sync.finish(); opt.step()
for _ in range(12): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
_lib.profile_start(only="attention*")
for _ in range(3): step()
torch.cuda.synchronize()
print("%s step %.2f ms  SAT_ATT_FAST=%s" % (cfg, dt * 1e3, os.environ.get("SAT_ATT_FAST", "1")))
for e in sorted(_lib.profile_stop(), key=lambda e: e["name"]):
    us = e["total_ms"] * 1e3 / e["launches"]
    print("  %-24s %6.2f us/launch  %7.1f GB/s algorithmic (%.3f of HBM peak)" % (e["name"], us, e["bytes"] / e["launches"] / us / 1e3, e["bytes"] / e["launches"] / us / 1e3 / 8000))
