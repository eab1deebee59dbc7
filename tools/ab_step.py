"""Dev tool: A/B of run-time switches on the whole C2 train step inside ONE process (variants alternate round by round, so clock /
thermal drift and box-to-box differences cancel; the per-process bench.py A/B has +-0.5 ms of noise).
usage: python tools/ab_step.py name=value[,name=value...] ...   (each argument = one variant; 'base' = defaults)
switches: sat_debug_option names (bn_ticket, wgrad3x3, reduce_z16, wide_tiles, glds_tile, ...) and py:bn_bwd_epilogue"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import sat_amd  # noqa
from sat_amd import _lib as L, model as M, encoder as E

cfg = os.environ.get("CFG", "c2")
hp, T, B, R = bench.hparams(cfg)
torch.manual_seed(42)
model = M.SAT(**hp).cuda().train(); model.set_precision("bf16")
model.__dict__["_sat_global_step"] = 2
opt = model.configure_optimizers()
img, caps, lengths = bench.synthetic_batch(B, R, T, hp["vocab_size"], 1234, False)
img, caps = img.cuda(), caps.cuda()
DEFAULTS = {"glds_tall_k": 1 << 30, "glds_tall_conv": 1152, "bn_onepass": 1, "bn_vpt": 2, "acc_prefetch": 0, "wgrad3x3": 1, "reduce_z16": 1, "wide_tiles": 0, "py:bn_bwd_epilogue": 1, "py:wgrad_stream": 1, "py:wgrad_streams": 1, "py:dgrad_join": 1, "py:fwd_res_bn": 1, "py:hi_prio": 0, "py:wgrad_side_only": 0}


def apply(settings):
    for k, v in {**DEFAULTS, **settings}.items():
        if k == "py:hi_prio":
            _hi[0] = bool(v)
        elif k.startswith("py:"):
            setattr(E, "_" + k[3:].upper(), int(v) if (k.endswith("streams") or k.endswith("side_only")) else bool(v))
        else:
            L.check(L.lib().sat_debug_option(k.encode(), int(v)), k)


HI = torch.cuda.Stream(priority=-1)      # py:hi_prio=1: the whole step on a high-priority stream (the side stream keeps the default priority)
_hi = [False]


def step():
    if _hi[0]:
        HI.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(HI):
            opt.zero_grad(set_to_none=True)
            out = model.training_step((img, caps, lengths), 0); out["loss"].backward(); opt.step()
        torch.cuda.current_stream().wait_stream(HI)
        return
    opt.zero_grad(set_to_none=True)
    out = model.training_step((img, caps, lengths), 0); out["loss"].backward(); opt.step()


variants = []
for a in sys.argv[1:]:
    variants.append((a, {} if a == "base" else {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.split(",")}))
for _ in range(30):
    step()
torch.cuda.synchronize()
rounds, per = int(os.environ.get("ROUNDS", "8")), int(os.environ.get("PER", "10"))
times = {n: [] for n, _ in variants}
for r in range(rounds):
    for n, s in variants:
        apply(s)
        for _ in range(3):
            step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(per):
            step()
        torch.cuda.synchronize(); times[n].append((time.perf_counter() - t0) / per * 1e3)
for n, ts in times.items():
    ts = sorted(ts)
    print("%-40s median %.3f ms  min %.3f  max %.3f" % (n, ts[len(ts) // 2], ts[0], ts[-1]))
