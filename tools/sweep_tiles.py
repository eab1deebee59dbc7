"""Dev tool: time every conv of the C2 encoder (forward / data gradient / weight gradient) under each tile form of the GEMM launcher
(sat_debug_option("tile_override", mode): 0 = the launcher's own choice, 128 = 128x128, 257 = 256x128, 256 = 256x256) and check the
forms against each other.  usage: python tools/sweep_tiles.py [resnet50|wrn101|decoder] [stages8]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sat_amd  # noqa
from sat_amd import _lib as L
lib = L.lib()
which = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
if len(sys.argv) > 2:
    L.check(lib.sat_debug_option(b"glds_stages8", int(sys.argv[2])), "opt")
B = 128
# (H, C, K, R, stride): distinct conv geometries of resnet50 at 256 px
R50 = [(64, 64, 64, 1, 1), (64, 64, 64, 3, 1), (64, 64, 256, 1, 1), (64, 256, 64, 1, 1), (64, 256, 128, 1, 1), (64, 128, 128, 3, 2), (32, 128, 512, 1, 1),
       (64, 256, 512, 1, 2), (32, 512, 128, 1, 1), (32, 128, 128, 3, 1), (32, 512, 256, 1, 1), (32, 256, 256, 3, 2), (16, 256, 1024, 1, 1), (32, 512, 1024, 1, 2),
       (16, 1024, 256, 1, 1), (16, 256, 256, 3, 1), (16, 1024, 512, 1, 1), (16, 512, 512, 3, 2), (8, 512, 2048, 1, 1), (16, 1024, 2048, 1, 2), (8, 2048, 512, 1, 1),
       (8, 512, 512, 3, 1)]
W101 = [(64, 64, 128, 1, 1), (64, 128, 128, 3, 1), (64, 128, 256, 1, 1), (64, 256, 128, 1, 1), (32, 256, 256, 3, 1), (32, 256, 512, 1, 1), (32, 512, 256, 1, 1),
        (16, 512, 512, 3, 1), (16, 512, 1024, 1, 1), (16, 1024, 512, 1, 1), (8, 1024, 1024, 3, 1), (8, 1024, 2048, 1, 1), (8, 2048, 1024, 1, 1)]
shapes = R50 if which == "resnet50" else W101
if which == "wrn101":
    B = 64
flush = torch.empty(768 << 20, dtype=torch.uint8, device="cuda")
MODES = [0, 128, 257, 256]


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


print("%-30s %-6s | %s | best" % ("conv (H,C,K,R,stride)", "form", " ".join("%9s" % ("m%d" % m) for m in MODES)))
tot = {m: 0.0 for m in MODES}; best_tot = 0.0
for (H, C, K, R, st) in shapes:
    pad = R // 2; P = (H + 2 * pad - R) // st + 1
    x = (torch.randn(B, H, H, C, device="cuda") * 0.5).bfloat16(); y = (torch.randn(B, P, P, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(K, R, R, C, device="cuda") * 0.1).bfloat16()
    g = L.ConvGeom(N=B, H=H, W=H, C=C, K=K, R=R, S=R, stride=st, pad=pad)
    slab = torch.empty(48 << 20, device="cuda")
    for form in ("fwd", "dgrad", "wgrad"):
        outs, row = {}, []
        for m in MODES:
            L.check(lib.sat_debug_option(b"tile_override", m if m else -1), "opt")
            if form == "fwd":
                o = torch.empty(B, P, P, K, device="cuda", dtype=torch.bfloat16)
                fn = lambda: L.check(lib.sat_conv2d_fwd_bf16(L.ptr(x), L.ptr(w), None, L.ptr(o), ctypes.byref(g), L.stream_ptr()), "f")
            elif form == "dgrad":
                o = torch.empty(B, H, H, C, device="cuda", dtype=torch.bfloat16)
                fn = lambda: L.check(lib.sat_conv2d_dgrad_bf16(L.ptr(y), L.ptr(w), L.ptr(o), ctypes.byref(g), 0, L.stream_ptr()), "d")
            else:
                o = torch.empty(K, R, R, C, device="cuda")
                fn = lambda: L.check(lib.sat_conv2d_wgrad_bf16(L.ptr(y), L.ptr(x), L.ptr(o), ctypes.byref(g), L.ptr(slab), slab.numel(), L.stream_ptr()), "w")
            t = timed(fn)
            outs[m] = o.float().clone(); row.append(t); tot[m] += t
        ref = outs[0]; scale = float(ref.abs().max()) + 1e-9
        bad = [m for m in MODES if float((outs[m] - ref).abs().max()) > (2e-2 if form != "wgrad" else 2e-3) * scale]
        best = min(range(len(MODES)), key=lambda i: row[i]); best_tot += row[best]
        fl = 2.0 * B * P * P * K * R * R * C
        print("%-30s %-6s | %s | m%-3d %6.0f TF %s" % ((H, C, K, R, st), form, " ".join("%9.1f" % t for t in row), MODES[best], fl / row[best] / 1e6,
                                                       ("MISMATCH " + str(bad)) if bad else ""))
print("sum (us):", {m: round(v) for m, v in tot.items()}, " best-per-shape:", round(best_tot))
