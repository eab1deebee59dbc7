"""Dev tool: where a conv / GEMM launch of the direct-to-LDS kernel spends its time - the same launch timed with parts switched off
(sat_debug_option("glds_ablate", bits): 1 no operand loads, 2 no MFMAs, 4 no LDS fragment reads, 8 no result stores; results are garbage).
usage: python tools/ablate_gemm.py form:H:C:K:R:stride[:tile] ..."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sat_amd  # noqa
from sat_amd import _lib as L
lib = L.lib()
B = int(os.environ.get("BATCH", "128"))
flush = torch.empty(768 << 20, dtype=torch.uint8, device="cuda")


def timed(fn, n=7):
    ts = []
    for i in range(n + 2):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i >= 2:
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


COMBOS = [0, 1, 2, 4, 8, 1 | 8, 2 | 4, 1 | 2 | 4, 1 | 2 | 4 | 8, 2 | 4 | 8]
print("%-34s " % "launch" + " ".join("%8s" % ("a%d" % c) for c in COMBOS) + "    (a0 = everything on; 1 loads off, 2 MFMA off, 4 fragment reads off, 8 stores off)")
for spec in sys.argv[1:]:
    f = spec.split(":"); form = f[0]; H, C, K, R, st = map(int, f[1:6]); tile = int(f[6]) if len(f) > 6 else 0
    pad = R // 2; P = (H + 2 * pad - R) // st + 1
    x = (torch.randn(B, H, H, C, device="cuda") * 0.5).bfloat16(); y = (torch.randn(B, P, P, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(K, R, R, C, device="cuda") * 0.1).bfloat16()
    g = L.ConvGeom(N=B, H=H, W=H, C=C, K=K, R=R, S=R, stride=st, pad=pad)
    slab = torch.empty(48 << 20, device="cuda")
    L.check(lib.sat_debug_option(b"tile_override", tile if tile else -1), "opt")
    if form == "fwd":
        o = torch.empty(B, P, P, K, device="cuda", dtype=torch.bfloat16)
        fn = lambda: L.check(lib.sat_conv2d_fwd_bf16(L.ptr(x), L.ptr(w), None, L.ptr(o), ctypes.byref(g), L.stream_ptr()), "f")
    elif form == "dgrad":
        o = torch.empty(B, H, H, C, device="cuda", dtype=torch.bfloat16)
        fn = lambda: L.check(lib.sat_conv2d_dgrad_bf16(L.ptr(y), L.ptr(w), L.ptr(o), ctypes.byref(g), 0, L.stream_ptr()), "d")
    else:
        o = torch.empty(K, R, R, C, device="cuda")
        fn = lambda: L.check(lib.sat_conv2d_wgrad_bf16(L.ptr(y), L.ptr(x), L.ptr(o), ctypes.byref(g), L.ptr(slab), slab.numel(), L.stream_ptr()), "w")
    row = []
    for c in COMBOS:
        L.check(lib.sat_debug_option(b"glds_ablate", c), "opt")
        row.append(timed(fn))
    L.check(lib.sat_debug_option(b"glds_ablate", 0), "opt")
    print("%-34s " % spec + " ".join("%8.1f" % t for t in row))
