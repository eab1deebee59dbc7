"""Dev tool: launch one bf16 conv form a few times (for rocprofv3 --pmc passes).  args: form N H W C K R stride [reps]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sat_amd  # noqa
from sat_amd import _lib as L
lib = L.lib()
form = sys.argv[1]; N, H, W, C, K, R, st = map(int, sys.argv[2:9]); reps = int(sys.argv[9]) if len(sys.argv) > 9 else 5
pad = R // 2; P = (H + 2 * pad - R) // st + 1
x = torch.randn(N, H, W, C, device="cuda").bfloat16(); y = torch.randn(N, P, P, K, device="cuda").bfloat16()
w = torch.randn(K, R, R, C, device="cuda").bfloat16()
dw = torch.empty(K, R, R, C, device="cuda"); slab = torch.empty(32 << 20, device="cuda"); dx = torch.empty_like(x)
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
g = L.ConvGeom(N=N, H=H, W=W, C=C, K=K, R=R, S=R, stride=st, pad=pad)
for _ in range(reps):
    flush.zero_()          # cold caches, like inside the train step
    if form == "wgrad":
        L.check(lib.sat_conv2d_wgrad_bf16(L.ptr(y), L.ptr(x), L.ptr(dw), ctypes.byref(g), L.ptr(slab), slab.numel(), L.stream_ptr()), "w")
    elif form == "fwd":
        L.check(lib.sat_conv2d_fwd_bf16(L.ptr(x), L.ptr(w), None, L.ptr(y), ctypes.byref(g), L.stream_ptr()), "f")
    else:
        L.check(lib.sat_conv2d_dgrad_bf16(L.ptr(y), L.ptr(w), L.ptr(dx), ctypes.byref(g), 0, L.stream_ptr()), "d")
torch.cuda.synchronize()
