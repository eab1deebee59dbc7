"""Dev tool: TFLOP/s of the GEMM / conv kernels on representative shapes."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sat_amd  # noqa
from sat_amd import _lib as L, decoder as dk

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

print("== dense NT (M,N,K)")
for M, N, K in [(640, 2688, 512), (13440, 6400, 256), (524288, 64, 64), (524288, 256, 64), (131072, 512, 128), (8192, 2048, 512), (4096, 4096, 4096)]:
    A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); out = torch.empty(M, N, device="cuda")
    t32 = timeit(lambda: dk.gemm(A, B, out=out))
    tb = timeit(lambda: dk.gemm(A, B, out=out, bf16_mfma=True))
    Ab, Bb = A.bfloat16(), B.bfloat16(); ob = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    tbb = timeit(lambda: dk.gemm(Ab, Bb, out=ob, bf16_mfma=True))
    fl = 2.0 * M * N * K / 1e12
    print("%8d %6d %6d  fp32 %7.1f TF  bf16mfma(f32 mem) %7.1f TF  bf16 mem %7.1f TF" % (M, N, K, fl / t32, fl / tb, fl / tbb))

print("== conv bf16 (N,H,W,C,K,R,stride)  fwd / dgrad / wgrad TF")
lib = L.lib()
for N, H, W, C, K, R, st in [(128, 64, 64, 64, 64, 3, 1), (128, 32, 32, 128, 128, 3, 1), (128, 16, 16, 256, 256, 3, 1), (128, 8, 8, 512, 512, 3, 1),
                             (128, 64, 64, 256, 64, 1, 1), (128, 16, 16, 1024, 256, 1, 1), (128, 64, 64, 128, 128, 3, 2), (128, 256, 256, 8, 64, 7, 2)]:
    pad = R // 2
    P = (H + 2 * pad - R) // st + 1
    x = torch.randn(N, H, W, C, device="cuda").bfloat16(); w = torch.randn(K, R, R, C, device="cuda").bfloat16()
    y = torch.empty(N, P, P, K, device="cuda", dtype=torch.bfloat16); dx = torch.empty_like(x); dw = torch.empty(K, R, R, C, device="cuda")
    slab = torch.empty(16 << 20, device="cuda")
    g = L.ConvGeom(N=N, H=H, W=W, C=C, K=K, R=R, S=R, stride=st, pad=pad)
    fl = 2.0 * N * P * P * K * R * R * C / 1e12
    tf = timeit(lambda: L.check(lib.sat_conv2d_fwd_bf16(L.ptr(x), L.ptr(w), None, L.ptr(y), ctypes.byref(g), L.stream_ptr()), "f"))
    td = timeit(lambda: L.check(lib.sat_conv2d_dgrad_bf16(L.ptr(y), L.ptr(w), L.ptr(dx), ctypes.byref(g), 0, L.stream_ptr()), "d"))
    tw = timeit(lambda: L.check(lib.sat_conv2d_wgrad_bf16(L.ptr(y), L.ptr(x), L.ptr(dw), ctypes.byref(g), L.ptr(slab), slab.numel(), L.stream_ptr()), "w"))
    print("%s  fwd %7.1f  dgrad %7.1f  wgrad %7.1f   (%.2f ms / %.2f / %.2f)" % ((N, H, W, C, K, R, st), fl / tf, fl / td, fl / tw, tf * 1e3, td * 1e3, tw * 1e3))
