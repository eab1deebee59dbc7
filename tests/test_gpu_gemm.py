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
