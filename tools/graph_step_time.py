"""Dev tool: one train step eager vs replayed from a hipGraph (sat_amd/graph.py): ms per step and host time to issue it.
usage: CFG=c1|c2|c3|c4|cli python tools/graph_step_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import sat_amd  # noqa
from sat_amd import model as M
from sat_amd.dist import GradSync
from sat_amd.graph import GraphedTrainStep

cfg = os.environ.get("CFG", "c2")
hp, T, B, R = bench.hparams(cfg)
if os.environ.get("TF") == "none" or cfg == "cli":
    hp["decoder_tf"] = None
if cfg == "cli":
    hp["deep_output"] = False
    hp["encoder_arch"] = os.environ.get("ARCH", hp["encoder_arch"])
torch.manual_seed(42)
model = M.SAT(**hp).cuda().train(); model.set_precision("bf16")
model.__dict__["_sat_global_step"] = 2
opt = model.configure_optimizers()
sync = GradSync(model)
img, caps, lengths = bench.synthetic_batch(B, R, T, hp["vocab_size"], 1234, False, px=hp["input_size"])
img, caps = img.cuda(), caps.cuda()
stepper = GraphedTrainStep(model, opt, sync=sync)


def eager():
    opt.zero_grad(set_to_none=True)
    out = model.training_step((img, caps, lengths), 0); out["loss"].backward(); sync.finish(); opt.step()
    return out


def timeit(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, out


n = int(os.environ.get("N", "20"))
for _ in range(max(10, int(1500 / 25))): out = eager()
del out
res = {}
for rnd in range(3):
    for name, fn in (("eager", eager), ("graph", lambda: stepper((img, caps, lengths), 0))):
        for _ in range(3): o = fn()
        del o
        issue, total, o = timeit(fn, n)
        del o
        res.setdefault(name, []).append((issue, total))
for name, v in res.items():
    v.sort(key=lambda x: x[1])
    print("%s %-5s host issue %.2f ms/step, step %.2f ms (median of %d rounds; min %.2f)" % (cfg, name, v[len(v) // 2][0], v[len(v) // 2][1], len(v), v[0][1]))
print(dict(stepper.stats))
