"""The reference's encoder table (dev/encoder_summaries.txt, produced by its dev/dev_encoder.py) measured with this package's ``get_encoder`` on one
MI355X: forward latency of the train-mode encoder at 224 px, mean 0.5 / std 0.5, no projection, after 5 warm-up calls over 100 calls; "amp=True"
= the bf16 storage mode (the reference ran torch AMP fp16 on an unnamed CUDA GPU), grad = whether the activations are kept for a backward.
usage: python tools/encoder_summaries.py [batch] [arch ...]"""
import os, sys, time
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sat_amd  # noqa
from sat_amd import encoder as E
from sat_amd.encoder_shuffle import SHUFFLENETS

batch = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 32
archs = [a for a in sys.argv[1:] if not a.isdigit()] or (list(E.RESNETS) + list(SHUFFLENETS) + ["mobilenet_v2"])
warmup, n_trials = 5, 100
data = torch.rand(batch, 3, 224, 224, device="cuda")
print("ENCODER SUMMARIES (sat_amd on %s)" % torch.cuda.get_device_name(0))
for arch in archs:
    args = SimpleNamespace(encoder_arch=arch, pretrained=False, input_size=224, encoder_dim=None, encoder_size=None, mean=[0.5] * 3, std=[0.5] * 3)
    torch.manual_seed(0)
    model = E.get_encoder(args).cuda().train()
    params = sum(p.numel() for p in model.state_dict().values() if p.dim() > 0 and p.dtype.is_floating_point and p.requires_grad is not None) * 0
    params = sum(v.numel() for k, v in model.state_dict().items() if not ("running" in k or "num_batches" in k)) * 1e-6          # reference shapes (state dict)
    for amp, grad in ((True, True), (True, False), (False, True)):
        model.precision = "bf16" if amp else "fp32"
        with torch.set_grad_enabled(grad):
            for _ in range(warmup):
                yhat = model(data)
            torch.cuda.synchronize(); t0 = time.time()
            for _ in range(n_trials):
                yhat = model(data)
            torch.cuda.synchronize(); duration = time.time() - t0
        latency = 1e3 * duration / n_trials
        _, features, h, w = yhat.shape
        print("arch=%-18s features=%4d attention=%3d params=%6.2fM. amp=%-5s grad=%-5s Latency=%7.3f ms. batch=%4d. Batches/s=%5.1f. Imgs/s=%7.1f."
              % (arch, features, h * w, params, amp, grad, latency, batch, n_trials / duration, batch * n_trials / duration), flush=True)
    del model, yhat
    torch.cuda.empty_cache()
