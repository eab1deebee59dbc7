"""Dev tool: weight-gradient GEMM timing for a few ResNet-50 shapes."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sat_amd  # noqa
from sat_amd import _lib as L
lib = L.lib()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
out = []
for N, H, W, C, K, R, st in [(128, 64, 64, 64, 64, 3, 1), (128, 32, 32, 128, 128, 3, 1), (128, 16, 16, 256, 256, 3, 1), (128, 64, 64, 64, 256, 1, 1), (128, 32, 32, 512, 128, 1, 1), (128, 16, 16, 1024, 256, 1, 1)]:
    pad = R // 2; P = (H + 2 * pad - R) // st + 1
    x = torch.randn(N, H, W, C, device="cuda").bfloat16(); y = torch.randn(N, P, P, K, device="cuda").bfloat16()
    dw = torch.empty(K, R, R, C, device="cuda"); slab = torch.empty(32 << 20, device="cuda")
    g = L.ConvGeom(N=N, H=H, W=W, C=C, K=K, R=R, S=R, stride=st, pad=pad)
    t = timeit(lambda: L.check(lib.sat_conv2d_wgrad_bf16(L.ptr(y), L.ptr(x), L.ptr(dw), ctypes.byref(g), L.ptr(slab), slab.numel(), L.stream_ptr()), "w"))
    fl = 2.0 * N * P * P * K * R * R * C / 1e12
    out.append("%s %.0fTF/%.0fus" % ((C, K, R), fl / t, t * 1e6))
print("  ".join(out))
