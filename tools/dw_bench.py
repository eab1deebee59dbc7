"""Dev tool: the depthwise 3x3 / shuffle kernels of csrc/depthwise.hip against the HBM roofline (algorithmic bytes / event time).
usage: python tools/dw_bench.py [images]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sat_amd  # noqa
from sat_amd import encoder_shuffle as S

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
PEAK = 8000.0


def timed(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3          # us


print("%-44s %9s %9s %8s" % ("kernel  (N=%d images, bf16)" % N, "us", "GB/s", "of peak"))
for (H, C, s) in ((56, 24, 2), (28, 24, 1), (28, 48, 2), (14, 48, 1), (14, 96, 2), (7, 96, 1), (112, 96, 2), (56, 144, 1), (28, 192, 1), (14, 384, 1), (7, 960, 1)):
    conv = torch.nn.Conv2d(C, C, 3, s, 1, bias=False, groups=C).cuda()
    x = torch.randn(N, H, H, C, device="cuda").to(torch.bfloat16)
    y = S.dw_fwd(x, conv)
    dy = torch.randn_like(y)
    by_in, by_out = x.numel() * 2, y.numel() * 2
    for name, fn, nbytes in (("dw3x3 fwd", lambda: S.dw_fwd(x, conv), by_in + by_out), ("dw3x3 dgrad", lambda: S.dw_dgrad(dy, conv, tuple(x.shape)), by_in + by_out),
                             ("dw3x3 wgrad", lambda: S.dw_wgrad(dy, x, conv), by_in + by_out)):
        us = timed(fn)
        gbs = nbytes / us / 1e3
        print("%-44s %9.1f %9.0f %8.3f" % ("%s %dx%dx%d s%d" % (name, H, H, C, s), us, gbs, gbs / PEAK))
for (H, Ch) in ((28, 24), (14, 48), (7, 96)):
    a = torch.randn(N, H, H, Ch, device="cuda").to(torch.bfloat16); b = torch.randn_like(a)
    us = timed(lambda: S.shuffle_join(a, b, True)); gbs = a.numel() * 2 * 4 / us / 1e3
    print("%-44s %9.1f %9.0f %8.3f" % ("shuffle join (halves) %dx%dx2x%d" % (H, H, Ch), us, gbs, gbs / PEAK))
