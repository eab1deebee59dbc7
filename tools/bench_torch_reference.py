"""Dev tool (context for DESIGN.md, not part of bench.py): the reference's own arithmetic -- the torch restatement in
oracle/sat_oracle.py, i.e. stock PyTorch ops (MIOpen / hipBLASLt kernels on ROCm) -- run on the same MI355X at config C2.
This is what the reference repository would deliver on this hardware."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
import bench
from oracle import sat_oracle as O

ap = argparse.ArgumentParser(); ap.add_argument("--amp", action="store_true"); ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--channels-last", action="store_true"); ap.add_argument("--math-lstm", action="store_true", default=True)
a = ap.parse_args()
hp, T, B, R = bench.hparams("c2")
ns = SimpleNamespace(**hp)
model = O.OracleSAT(ns, None, seed=42)
dev = torch.device("cuda")
model.encoder = model.encoder.to(dev)
if a.channels_last:
    model.encoder = model.encoder.to(memory_format=torch.channels_last)
model.sd = {k: v.detach().to(dev).requires_grad_() for k, v in model.sd.items()}
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
img, caps, lengths = bench.synthetic_batch(B, R, T, hp["vocab_size"], 1234, False)
img, caps, lengths = img.to(dev), caps.to(dev), lengths.to(dev)

def zeros_like_ref(*shape):
    return torch.zeros(*shape, device=dev)
torch.zeros_default = torch.zeros
_orig_zeros = torch.zeros
def _zeros(*args, **kw):
    kw.setdefault("device", dev) if not any(isinstance(x, torch.device) for x in args) else None
    return _orig_zeros(*args, **kw)
torch.zeros = _zeros          # the restatement allocates logits/alphas with torch.zeros(...) on the default device

def step():
    opt.zero_grad(set_to_none=True)
    x = img.clone()
    if a.channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=a.amp):      # AMP on the conv stack (the decoder stays fp32)
        ann = model.encoder(x)
    ann = ann.float()
    if True:
        loss, _ = O.training_loss(model.sd, model.hp, ann, caps, lengths, 1.0, draw=lambda: 0.0) if not a.math_lstm else None, None
        if a.math_lstm:
            out = O.decode_train(model.sd, model.hp, ann, caps, lengths, 1.0, lambda: 0.0, lstm_fn=O.lstm_step_math)
            lp, _ = O.pack_time_major(out['logits'], out['lengths'].cpu()); tp, _ = O.pack_time_major(out['targets'], out['lengths'].cpu())
            loss = O.label_smoothing_ce(lp, tp, 0.0) + O.doubly_stochastic(out['alphas'], 1.0)
    loss.backward()
    opt.step()
    return loss

for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(a.steps): l = step()
torch.cuda.synchronize(); dt = (time.time() - t0) / a.steps
print("torch restatement on MI355X (amp=%s, channels_last=%s): %.1f ms/step -> %.0f captions/s (loss %.3f)" % (a.amp, a.channels_last, dt * 1e3, B * R / dt, float(l)))
