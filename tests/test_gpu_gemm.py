"""GPU: the MFMA GEMM family (csrc/gemm.hip) through the C ABI against torch fp32 on the CPU.
Tolerance: fp32 accumulation in a different order than the CPU BLAS -> 2e-5 * K-scaled magnitude."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tol(ref, K):
    return 3e-6 * max(1.0, float(ref.abs().max())) * max(1.0, K ** 0.5)


@pytest.fixture(scope="module")
def dk():
    import sat_amd  # noqa: F401
    from sat_amd import decoder
    return decoder


SHAPES = [(1, 1, 1), (5, 7, 3), (64, 64, 16), (33, 65, 17), (130, 70, 129), (257, 300, 64), (640, 2688, 512), (37, 23, 10), (512, 512, 8)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("amode,bmode", [(0, 0), (0, 1), (1, 1)])
def test_dense_modes(dk, M, N, K, amode, bmode):
    g = torch.Generator().manual_seed(M * 1000 + N * 10 + K)
    A = torch.randn(M, K, generator=g); B = torch.randn(K, N, generator=g)
    ref = A @ B
    Ad = (A if amode == 0 else A.t().contiguous()).cuda()
    Bd = (B.t().contiguous() if bmode == 0 else B).cuda()
    out = dk.gemm(Ad, Bd, amode=amode, bmode=bmode)
    assert (out.cpu() - ref).abs().max().item() <= _tol(ref, K)


def test_split_k_and_accumulate(dk):
    g = torch.Generator().manual_seed(1)
    M, N, K = 96, 40, 8192
    A = torch.randn(K, M, generator=g); B = torch.randn(K, N, generator=g); C0 = torch.randn(M, N, generator=g)
    ref = A.t() @ B + C0
    slab = torch.empty(4 << 20, device="cuda")
    out = dk.gemm(A.cuda(), B.cuda(), amode=1, bmode=1, out=C0.cuda().clone(), accumulate=True, slab=slab)
    assert (out.cpu() - ref).abs().max().item() <= _tol(ref, K)


def test_gather_scatter_and_epilogues(dk):
    g = torch.Generator().manual_seed(2)
    M, N, K, R = 50, 36, 24, 80
    src = torch.randn(R, K, generator=g); W = torch.randn(N, K, generator=g); bias = torch.randn(N, generator=g)
    rows = torch.randint(0, R, (M,), generator=g).to(torch.int32); rows[3] = -1
    Ag = src[rows.clamp(min=0).long()] * (rows >= 0).float()[:, None]
    ref = Ag @ W.t() + bias
    out = dk.gemm(src.cuda(), W.cuda(), a_rows=rows.cuda(), bias=bias.cuda(), epi=1)
    assert (out.cpu() - ref).abs().max().item() <= _tol(ref, K)
    # sigmoid on a column range
    ref2 = ref.clone(); ref2[:, 5:20] = torch.sigmoid(ref2[:, 5:20])
    out2 = dk.gemm(src.cuda(), W.cuda(), a_rows=rows.cuda(), bias=bias.cuda(), epi=2, c0=5, c1=20)
    assert (out2.cpu() - ref2).abs().max().item() <= 2e-5
    # scatter rows
    perm = torch.randperm(M, generator=g).to(torch.int32); perm[7] = -1
    out3 = dk.gemm(Ag.cuda(), W.cuda(), c_rows=perm.cuda(), out_rows=M)
    ref3 = torch.zeros(M, N)
    for r in range(M):
        if perm[r] >= 0:
            ref3[perm[r]] = (Ag @ W.t())[r]
    assert (out3.cpu() - ref3).abs().max().item() <= _tol(ref, K)
    # tanh(v + e0[arow]) and v * (1 - e0^2)
    e0 = torch.randn(R, N, generator=g)
    ref4 = torch.tanh(Ag @ W.t() + e0[rows.clamp(min=0).long()])
    r2 = rows.clone(); r2[3] = 0
    Ag2 = src[r2.long()]
    ref4 = torch.tanh(Ag2 @ W.t() + e0[r2.long()])
    out4 = dk.gemm(src.cuda(), W.cuda(), a_rows=r2.cuda(), e0=e0.cuda(), epi=3)
    assert (out4.cpu() - ref4).abs().max().item() <= 2e-5
    u = torch.tanh(torch.randn(M, N, generator=g))
    out5 = dk.gemm(Ag2.cuda(), W.cuda(), e0=u.cuda(), epi=4)
    assert (out5.cpu() - (Ag2 @ W.t()) * (1 - u * u)).abs().max().item() <= _tol(ref, K)


# ----------------------------------------------------------------------------- bf16 MFMA kernel (csrc/gemm_bf16.hip)
def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


BSHAPES = [(64, 64, 32), (128, 128, 64), (640, 2688, 512), (200, 72, 96), (136, 264, 40), (1024, 512, 2048), (8, 8, 8),
           (4096, 1024, 256), (2048, 2048, 512), (3000, 1536, 192)]      # >= 192 tiles of 128x128: the kernels the C2 step mostly runs


@pytest.mark.parametrize("M,N,K", BSHAPES)
@pytest.mark.parametrize("amode,bmode", [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("store", ["f32", "bf16"])
def test_bf16_mfma_dense_modes(dk, M, N, K, amode, bmode, store):
    """fp32 operands rounded to bf16 in-kernel, and bf16 operands in HBM: both equal the fp32 product of the
    bf16-rounded inputs up to accumulation order."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g); B = torch.randn(K, N, generator=g)
    ref = _bf(A) @ _bf(B)
    dt = torch.float32 if store == "f32" else torch.bfloat16
    Ad = (A if amode == 0 else A.t().contiguous()).to(dt).cuda()
    Bd = (B.t().contiguous() if bmode == 0 else B).to(dt).cuda()
    out = dk.gemm(Ad, Bd, amode=amode, bmode=bmode, bf16_mfma=True)
    assert (out.cpu() - ref).abs().max().item() <= _tol(ref, K)
    if store == "bf16" and not (amode == 1 and bmode == 1):
        outb = dk.gemm(Ad, Bd, amode=amode, bmode=bmode, bf16_mfma=True, out_dtype=torch.bfloat16) if (amode, bmode) != (0, 1) or True else None
        assert (outb.float().cpu() - ref).abs().max().item() <= 1e-2 * max(1.0, float(ref.abs().max()))


def test_bf16_mfma_splitk_gather_epilogue(dk):
    g = torch.Generator().manual_seed(11)
    M, N, K = 96, 40, 4096
    A = torch.randn(K, M, generator=g); B = torch.randn(K, N, generator=g)
    slab = torch.empty(4 << 20, device="cuda")
    out = dk.gemm(A.cuda(), B.cuda(), amode=1, bmode=1, slab=slab, bf16_mfma=True)
    ref = _bf(A).t() @ _bf(B)
    assert (out.cpu() - ref).abs().max().item() <= _tol(ref, K)
    src = torch.randn(300, 64, generator=g); W = torch.randn(48, 64, generator=g); bias = torch.randn(48, generator=g)
    rows = torch.randint(0, 300, (130,), generator=g).to(torch.int32); rows[5] = -1
    Ag = _bf(src)[rows.clamp(min=0).long()] * (rows >= 0).float()[:, None]
    ref2 = Ag @ _bf(W).t() + bias; ref2[:, 8:24] = torch.sigmoid(ref2[:, 8:24])
    out2 = dk.gemm(src.cuda(), W.cuda(), a_rows=rows.cuda(), bias=bias.cuda(), epi=2, c0=8, c1=24, bf16_mfma=True)
    assert (out2.cpu() - ref2).abs().max().item() <= 3e-5


def test_bf16_request_falls_back_to_fp32_kernel_on_odd_shapes(dk):
    g = torch.Generator().manual_seed(12)
    A = torch.randn(9, 7, generator=g); B = torch.randn(5, 7, generator=g)
    out = dk.gemm(A.cuda(), B.cuda(), bf16_mfma=True)
    assert (out.cpu() - A @ B.t()).abs().max().item() <= 1e-5
