"""Dev tool: host time to ISSUE one C2 train step (no synchronisation) against the time the GPU needs for it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import sat_amd  # noqa
from sat_amd import model as M
hp, T, B, R = bench.hparams(os.environ.get("CFG", "c2"))
if os.environ.get("CFG") == "cli":
    hp["decoder_tf"] = None; hp["deep_output"] = False
torch.manual_seed(42)
model = M.SAT(**hp).cuda().train(); model.set_precision("bf16")
model.__dict__["_sat_global_step"] = 2
opt = model.configure_optimizers()
img, caps, lengths = bench.synthetic_batch(B, R, T, hp["vocab_size"], 1234, False, px=hp["input_size"])
img, caps = img.cuda(), caps.cuda()
def step():
    opt.zero_grad(set_to_none=True)
    out = model.training_step((img, caps, lengths), 0); out["loss"].backward(); opt.step()
for _ in range(30): step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(4): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("issue %.2f ms/step on the host; %.2f ms/step until the GPU is done" % ((t1 - t0) / 4 * 1e3, (t2 - t0) / 4 * 1e3))
if os.environ.get("PROFILE"):
    import cProfile, pstats
    torch.autograd.set_multithreading_enabled(False)          # the backward's Python runs on this thread: the profiler sees it
    for _ in range(2): step()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(4): step()
    pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)

