"""Dev tool / BASELINE configs[4] (C5): caption() throughput, greedy (beam 1) vs beam 5, resnet50 encoder, 64 images, on one GPU.
Decode-only timing (annotations precomputed) and end-to-end timing (encoder included)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import sat_amd  # noqa
from sat_amd import model as M

hp, T, B, R = bench.hparams("c2")
torch.manual_seed(42)
model = M.SAT(**hp).cuda().eval(); model.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
img = torch.rand(64, 3, hp["input_size"], hp["input_size"], device="cuda")
with torch.no_grad():
    ann, hw = model.encode(img)
    ann = ann.contiguous()
    for beamk in (1, 5):
        for _ in range(2):
            model.beam_decode(ann[:8], hw, beamk=beamk, max_gen_length=20)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        caps, scores, alphas, ppl = model.beam_decode(ann, hw, beamk=beamk, max_gen_length=20)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        steps = sum(len(c) + 1 for c in caps)
        print("beam %d: decode-only %.1f images/s (%.1f ms/image, mean caption length %.1f, %.0f us per decode step of the kept hypothesis)"
              % (beamk, 64 / dt, dt / 64 * 1e3, steps / 64, dt / steps * 1e6))
        for _ in range(2):
            model.beam_decode_batched(ann, hw, beamk=beamk, max_gen_length=20)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            capsb, _, _, _ = model.beam_decode_batched(ann, hw, beamk=beamk, max_gen_length=20)
        torch.cuda.synchronize(); dtb = (time.perf_counter() - t0) / 5
        assert capsb == caps
        print("beam %d: batched search, decode-only %.0f images/s (%.2f ms per 64 images incl. host back-trace), %.1fx the per-image loop"
              % (beamk, 64 / dtb, dtb * 1e3, dt / dtb))
        for _ in range(2):
            model.beam_decode_batched(ann, hw, beamk=beamk, max_gen_length=20, graph=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            capsg, _, _, _ = model.beam_decode_batched(ann, hw, beamk=beamk, max_gen_length=20, graph=True)
        torch.cuda.synchronize(); dtg = (time.perf_counter() - t0) / 5
        assert capsg == caps
        print("beam %d: batched search replayed from a hipGraph, decode-only %.0f images/s (%.2f ms per 64 images incl. host back-trace)" % (beamk, 64 / dtg, dtg * 1e3))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            model.caption(img, beamk=beamk, max_gen_length=20)
        torch.cuda.synchronize(); dte = (time.perf_counter() - t0) / 3
        print("beam %d: caption() end to end (encoder + batched decode) %.0f images/s" % (beamk, 64 / dte))
