"""Dev tool: many train steps in one process - loss trend, device memory, host RSS (leak check).  usage: CFG=c2|c1|cli [GRAPH=1] [STEPS=600] python tools/soak.py"""
import os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import sat_amd  # noqa
from sat_amd import model as M
from sat_amd.dist import GradSync
from sat_amd.graph import GraphedTrainStep

cfg = os.environ.get("CFG", "c2")
hp, T, B, R = bench.hparams(cfg)
if cfg == "cli":
    hp["decoder_tf"] = None; hp["deep_output"] = False
torch.manual_seed(42)
model = M.SAT(**hp).cuda().train(); model.set_precision("bf16")
model.__dict__["_sat_global_step"] = 2
opt = model.configure_optimizers()
sync = GradSync(model)
batches = []
for s in range(4):          # four batches with different caption lengths: four packing plans / graphs
    img, caps, lengths = bench.synthetic_batch(B, R, T, hp["vocab_size"], 1234 + s, True, px=hp["input_size"])
    batches.append((img.cuda(), caps.cuda(), lengths))
stepper = GraphedTrainStep(model, opt, sync=sync) if os.environ.get("GRAPH") == "1" else None
n = int(os.environ.get("STEPS", "600"))
t0 = time.time()
for it in range(n):
    b = batches[it % 4]
    if stepper is not None:
        out = stepper(b, it)
    else:
        opt.zero_grad(set_to_none=True)
        out = model.training_step(b, it); out["loss"].backward(); sync.finish(); opt.step()
    if it % (n // 6) == 0 or it == n - 1:
        torch.cuda.synchronize()
        print("step %4d loss %.4f  device allocated %.1f MB reserved %.1f MB  host max RSS %.0f MB  %.1f s" % (
            it, float(out["loss"].detach()), torch.cuda.memory_allocated() / 1e6, torch.cuda.memory_reserved() / 1e6,
            resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3, time.time() - t0), flush=True)
if stepper is not None:
    print(dict(stepper.stats))
