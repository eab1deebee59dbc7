"""Dev tool: time bf16 conv forms with cold caches.  args: list of form:N:H:C:K:R:stride separated by spaces"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sat_amd  # noqa
from sat_amd import _lib as L
lib = L.lib()
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for spec in sys.argv[1:]:
    form, N, H, C, K, R, st = spec.split(":"); N, H, C, K, R, st = map(int, (N, H, C, K, R, st)); W = H
    pad = R // 2; P = (H + 2 * pad - R) // st + 1
    x = torch.randn(N, H, W, C, device="cuda").bfloat16(); y = torch.randn(N, P, P, K, device="cuda").bfloat16()
    w = torch.randn(K, R, R, C, device="cuda").bfloat16()
    dw = torch.empty(K, R, R, C, device="cuda"); slab = torch.empty(32 << 20, device="cuda"); dx = torch.empty_like(x)
    g = L.ConvGeom(N=N, H=H, W=W, C=C, K=K, R=R, S=R, stride=st, pad=pad)
    ts = []
    for i in range(12):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if form == "wgrad":
            L.check(lib.sat_conv2d_wgrad_bf16(L.ptr(y), L.ptr(x), L.ptr(dw), ctypes.byref(g), L.ptr(slab), slab.numel(), L.stream_ptr()), "w")
        elif form == "fwd":
            L.check(lib.sat_conv2d_fwd_bf16(L.ptr(x), L.ptr(w), None, L.ptr(y), ctypes.byref(g), L.stream_ptr()), "f")
        else:
            L.check(lib.sat_conv2d_dgrad_bf16(L.ptr(y), L.ptr(w), L.ptr(dx), ctypes.byref(g), 0, L.stream_ptr()), "d")
        e1.record(); torch.cuda.synchronize()
        if i >= 4: ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    by = 2.0 * (N * H * W * C + N * P * P * K + K * R * R * C)
    print("%-28s median %7.1f us  min %7.1f us  %6.0f GB/s" % (spec, ts[len(ts) // 2], ts[0], by / ts[len(ts) // 2] / 1e3))
