"""GPU parity of the reference's sub-modules called on their own (InitLSTM / SoftAttention / DeepOutput / beta / embedding /
lstm, reference model.py:66-131, 158-192) on the step entry points of the C ABI, against the fixtures the REFERENCE produced
(G1 attention forward + gradients, G2 InitLSTM incl. the F3 reshape, G3 one full decode step in four variants, G6 LabelSmoothing)
and against torch's own modules on the CPU for the gradients the fixtures do not hold.  Tolerance (north_star): 1e-4."""
import ctypes
import os

import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu
TOL = 1e-4


def close(a, b, tol=TOL, what=""):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1.0, float(np.abs(b).max())) if b.size else 1.0
    err = float(np.abs(a - b).max()) if b.size else 0.0
    assert err <= tol * scale, "%s: max|d|=%.3e (scale %.3g, tol %.1e)" % (what, err, scale, tol)


@pytest.fixture(scope="module")
def M():
    import sat_amd  # noqa: F401
    from sat_amd import model
    return model


def small_hp(**over):
    from oracle import sat_oracle as O
    base = dict(vocab_size=23, encoder_dim=12, embed_dim=10, attention_dim=7, decoder_dim=9, input_size=64)
    base.update(over)
    return O.default_hparams(**base)


def test_g1_soft_attention_forward_and_gradients(M, golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_attention.npz"))
    att = M.SoftAttention(small_hp()).cuda()
    with torch.no_grad():
        att.encoder_att.weight.copy_(torch.from_numpy(g["We"])); att.decoder_att.weight.copy_(torch.from_numpy(g["Wd"]))
        att.f_att.weight.copy_(torch.from_numpy(g["wf"]))
    ann = torch.from_numpy(g["ann"]).cuda().requires_grad_()             # (N, D, h, w) with h != w, as the reference passes it
    hid = torch.from_numpy(g["hid"]).cuda().requires_grad_()
    z, alpha = att(ann, hid)                                             # model.py:94: self.attention(annotations, h)
    assert alpha.shape == g["alpha"].shape
    close(z, g["z"], what="z"); close(alpha, g["alpha"], what="alpha")
    ((z * torch.from_numpy(g["gz"]).cuda()).sum() + (alpha * torch.from_numpy(g["ga"]).cuda()).sum()).backward()
    close(ann.grad, g["d_ann"], what="d_ann"); close(hid.grad, g["d_hid"], what="d_hid")
    close(att.encoder_att.weight.grad, g["d_We"], what="d_We"); close(att.decoder_att.weight.grad, g["d_Wd"], what="d_Wd")
    close(att.f_att.weight.grad, g["d_wf"], what="d_wf")


@pytest.mark.parametrize("layers", [1, 2])
@pytest.mark.parametrize("N", [4, 5])
def test_g2_init_lstm_reproduces_the_raw_reshape(M, golden_dir, layers, N):
    g = np.load(os.path.join(golden_dir, "g2_initlstm.npz"))
    mod = M.InitLSTM(small_hp(decoder_layers=layers)).cuda()
    with torch.no_grad():
        for k in ("factorize.weight", "factorize.bias", "init.weight", "init.bias"):
            dict(mod.named_parameters())[k].copy_(torch.from_numpy(g["L%d_init_lstm.%s" % (layers, k)]))
    tag = "L%d_N%d_" % (layers, N)
    ann_cpu = torch.from_numpy(g[tag + "ann"])
    ann = ann_cpu.cuda().requires_grad_()
    h0, c0 = mod(ann)                                                    # model.py:269 / 498
    close(h0, g[tag + "h0"], what="h0"); close(c0, g[tag + "c0"], what="c0")
    # gradients against torch's modules on the CPU (the fixture holds none)
    ref = nn.Sequential(nn.Linear(12, 10), nn.Linear(10, 2 * 9 * layers))
    with torch.no_grad():
        ref[0].weight.copy_(mod.factorize.weight.cpu()); ref[0].bias.copy_(mod.factorize.bias.cpu())
        ref[1].weight.copy_(mod.init.weight.cpu()); ref[1].bias.copy_(mod.init.bias.cpu())
    ann_r = ann_cpu.clone().requires_grad_()
    init_r = ref(ann_r.mean((2, 3))).reshape(2 * layers, N, 9)
    gw = torch.randn(2 * layers, N, 9, generator=torch.Generator().manual_seed(3))
    (init_r * gw).sum().backward()
    (torch.cat([h0, c0], 0) * gw.cuda()).sum().backward()
    close(ann.grad, ann_r.grad, what="d_ann")
    close(mod.factorize.weight.grad, ref[0].weight.grad, what="dW_f"); close(mod.factorize.bias.grad, ref[0].bias.grad, what="db_f")
    close(mod.init.weight.grad, ref[1].weight.grad, what="dW_i"); close(mod.init.bias.grad, ref[1].bias.grad, what="db_i")


@pytest.mark.parametrize("tag", ["deep", "shallow", "tied", "layers2"])
def test_g3_one_decode_step_through_the_sub_modules(M, golden_dir, tag):
    """model.py:526-547 written exactly as the reference writes it, on this package's modules."""
    g = np.load(os.path.join(golden_dir, "g3_step_%s.npz" % tag))
    sd = {k[3:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith("sd.")}
    hp = small_hp(deep_output=(tag != "shallow"), weight_tying=(tag == "tied"), decoder_layers=(2 if tag == "layers2" else 1))
    model = M.SATDecoder(hp).cuda()
    model.load_decoder_state(sd)
    ann, h, c, tok = (torch.from_numpy(g[k]).cuda() for k in ("ann", "h", "c", "tok"))
    with torch.no_grad():
        y = model.embedding_dropout(model.embedding(tok))
        z, alpha = model.attention(ann, h[-1])
        beta = model.beta(h[-1])
        h_in = torch.cat([y, beta * z], dim=1).unsqueeze(0)
        _, (hn, cn) = model.lstm(h_in, (h, c))
        logit = model.output(y, hn[-1], z)
    for name, val in (("z", z), ("alpha", alpha), ("beta", beta), ("hn", hn), ("cn", cn), ("logit", logit)):
        close(val, g[name], what="%s/%s" % (tag, name))


def test_step_gradients_match_torch_modules_on_the_cpu(M):
    """embedding -> gate -> LSTM (2 layers) -> DeepOutput with dropout off: every gradient against torch's nn modules on the CPU."""
    hp = small_hp(decoder_layers=2, deep_output=True)
    torch.manual_seed(5)
    dev = M.SATDecoder(hp).cuda()
    N = 6
    gen = torch.Generator().manual_seed(9)
    tok = torch.randint(0, 23, (N,), generator=gen)
    z0 = torch.randn(N, 12, generator=gen)
    h0 = torch.randn(2, N, 9, generator=gen) * 0.5; c0 = torch.randn(2, N, 9, generator=gen) * 0.5
    gl = torch.randn(N, 23, generator=gen); gh = torch.randn(2, N, 9, generator=gen); gc = torch.randn(2, N, 9, generator=gen)
    # reference modules on the CPU with the same weights
    emb = nn.Embedding(23, 10, padding_idx=0); lstm = nn.LSTM(22, 9, 2); gate = nn.Sequential(nn.Linear(9, 12), nn.Sigmoid())
    Wh, Wc, Wo = nn.Linear(9, 10, bias=False), nn.Linear(12, 10, bias=False), nn.Linear(10, 23)
    with torch.no_grad():
        emb.weight.copy_(dev.embedding.weight.cpu()); gate[0].weight.copy_(dev.beta[0].weight.cpu()); gate[0].bias.copy_(dev.beta[0].bias.cpu())
        for k, p in lstm.named_parameters():
            p.copy_(getattr(dev.lstm, k).cpu())
        Wh.weight.copy_(dev.output.hidden.weight.cpu()); Wc.weight.copy_(dev.output.context.weight.cpu())
        Wo.weight.copy_(dev.output.output.weight.cpu()); Wo.bias.copy_(dev.output.output.bias.cpu())

    def run(embedding, beta, rnn, out_fn, to):
        z = to(z0).requires_grad_(); h = to(h0).requires_grad_(); c = to(c0).requires_grad_()
        y = embedding(to(tok))
        x = torch.cat([y, beta(h[-1]) * z], 1).unsqueeze(0)
        _, (hn, cn) = rnn(x, (h, c))
        logit = out_fn(y, hn[-1], z)
        ((logit * to(gl)).sum() + (hn * to(gh)).sum() + (cn * to(gc)).sum()).backward()
        return z.grad, h.grad, c.grad

    ref = run(emb, gate, lstm, lambda y, hh, z: Wo(torch.tanh(y + Wh(hh) + Wc(z))), lambda t: t.clone())
    got = run(dev.embedding, dev.beta, dev.lstm, dev.output, lambda t: t.cuda())
    for name, a, b in zip(("dz", "dh", "dc"), got, ref):
        close(a, b, what=name)
    close(dev.embedding.weight.grad, emb.weight.grad, what="dE")
    close(dev.beta[0].weight.grad, gate[0].weight.grad, what="dW_beta"); close(dev.beta[0].bias.grad, gate[0].bias.grad, what="db_beta")
    for k, p in lstm.named_parameters():
        close(getattr(dev.lstm, k).grad, p.grad, what="lstm." + k)
    close(dev.output.hidden.weight.grad, Wh.weight.grad, what="dW_h"); close(dev.output.context.weight.grad, Wc.weight.grad, what="dW_c")
    close(dev.output.output.weight.grad, Wo.weight.grad, what="dW_o"); close(dev.output.output.bias.grad, Wo.bias.grad, what="db_o")


def test_shallow_output_and_dropout_masks(M):
    """shallow DeepOutput ignores the embedding and the context; with dropout the masks of forward and backward agree
    (gradient of sum(logits) w.r.t. hidden equals the finite effect of the same masks: checked through linearity)."""
    hp = small_hp(deep_output=False, dropout=0.4)
    torch.manual_seed(1)
    out = M.DeepOutput(hp).cuda().train()
    hdn = torch.randn(8, 9, device="cuda", requires_grad=True)
    torch.manual_seed(77)
    logit = out(None, hdn, None)
    logit.sum().backward()
    # logits are linear in `hidden` for a fixed mask: logit == J hidden + b with the J backward used
    torch.manual_seed(77)
    again = out(None, hdn.detach(), None)
    assert torch.equal(again, logit.detach())                            # same seed -> same masks
    lin = (hdn.grad * hdn.detach()).sum() + out.output.bias.sum() * 8
    assert abs(float(lin) - float(logit.sum())) <= 1e-3 * max(1.0, abs(float(logit.sum())))
    out.eval()
    ref = (hdn.detach().cpu() @ out.hidden.weight.detach().cpu().t()) @ out.output.weight.detach().cpu().t() + out.output.bias.detach().cpu()
    close(out(None, hdn.detach(), None), ref, what="shallow eval")


def test_embedding_max_norm_renormalises_in_place_like_torch(M):
    hp = small_hp(embed_norm=0.7)
    torch.manual_seed(2)
    emb = M.Embedding(23, 10, max_norm=0.7, padding_idx=0).cuda()
    ref = nn.Embedding(23, 10, max_norm=0.7, padding_idx=0)
    with torch.no_grad():
        ref.weight.copy_(emb.weight.cpu())
    tok = torch.tensor([[3, 5, 3], [0, 22, 7]])
    y, yr = emb(tok.cuda()), ref(tok)
    close(y, yr, what="rows"); close(emb.weight, ref.weight, what="table after renorm")
    y.sum().backward(); yr.sum().backward()
    close(emb.weight.grad, ref.weight.grad, what="dE")


@pytest.mark.parametrize("smoothing", [0.0, 0.15, 0.3])
def test_g6_label_smoothing_kernel_against_the_reference_fixture(M, golden_dir, smoothing):
    g = np.load(os.path.join(golden_dir, "g6_label_smoothing.npz"))
    x = torch.from_numpy(g["x"]).cuda().requires_grad_()
    t = torch.from_numpy(g["t"]).cuda()
    crit = M.LabelSmoothing(smoothing)
    loss = crit(x, t)
    loss.backward()
    close(loss, g["loss_%g" % smoothing], tol=1e-5, what="loss"); close(x.grad, g["grad_%g" % smoothing], tol=1e-5, what="grad")
    if smoothing == 0.0:
        close(loss, g["ce_torch"], tol=1e-5, what="== F.cross_entropy (dev/dev_label_smoothing.py)")
    acc = float((x.detach().argmax(1) == t).float().mean())
    assert abs(float(crit.last_accuracy) - acc) < 1e-6


@pytest.mark.parametrize("P,V", [(7, 10000), (5, 1003), (3, 4), (9, 4100), (2, 1)])
def test_label_smoothing_kernel_vector_and_scalar_paths(M, P, V):
    """util.py:105-112 at vocabulary sizes on both sides of the 16-byte path (V % 4 == 0) and with tied maxima (argmax = first index,
    as torch): loss / gradient against float64 torch on the CPU, tolerance 5e-6 relative to the largest entry."""
    torch.manual_seed(P * 131 + V)
    x = (torch.randn(P, V) * 3.0)
    if V >= 4:
        x[0, V // 2] = x[0].max() + 1.0; x[0, V - 1] = x[0, V // 2]              # a tie: the first one counts
    t = torch.randint(0, V, (P,)); t[0] = V // 2
    xg = x.cuda().requires_grad_()
    crit = M.LabelSmoothing(0.1)
    loss = crit(xg, t.cuda()); (loss * 1.7).backward()
    xd = x.double().requires_grad_()
    lp = torch.log_softmax(xd, -1)
    ref = (0.9 * (-lp.gather(1, t[:, None])[:, 0]) + 0.1 * (-lp.mean(-1))).mean()
    (ref * 1.7).backward()
    close(loss, ref.detach().float(), tol=5e-6, what="loss"); close(xg.grad, xd.grad.float(), tol=5e-6, what="gradient")
    acc = float((x.argmax(1) == t).float().mean())
    assert abs(float(crit.last_accuracy) - acc) < 1e-6


def test_attention_step_fwd_bound_exactly_as_integration_md(golden_dir):
    """INTEGRATION.md section 2, verbatim: a bare ctypes binding of sat_attention_precompute / sat_attention_step_fwd inside a module
    with the reference's SoftAttention signature, checked against fixture G1."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _sat = ctypes.CDLL(os.path.join(root, "show-attend-and-tell-pytorch-lightning_amd", "libsat_hip.so"))
    _sat.sat_last_error.restype = ctypes.c_char_p

    def _check(rc):
        if rc:
            raise RuntimeError(_sat.sat_last_error().decode())

    _p = lambda t: ctypes.c_void_p(t.data_ptr())                          # noqa: E731
    _stream = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)   # noqa: E731

    class SoftAttention(nn.Module):                       # model.py:84-109
        def __init__(self, D, A, n):
            super().__init__()
            self.encoder_att = nn.Linear(D, A, bias=False); self.decoder_att = nn.Linear(n, A, bias=False); self.f_att = nn.Linear(A, 1, bias=False)

        def forward(self, annotations, decoder_hidden):
            b, c, h, w = annotations.shape
            ann = annotations.permute(0, 2, 3, 1).reshape(b, h * w, c).contiguous()      # (B, L, D)
            A = self.encoder_att.weight.shape[0]
            U = torch.empty(b, h * w, A, device=ann.device)
            _check(_sat.sat_attention_precompute(_p(ann), _p(self.encoder_att.weight), _p(U), b, h * w, c, A, _stream()))
            hc = torch.cat([decoder_hidden @ self.decoder_att.weight.t(), torch.ones(b, c, device=ann.device)], 1).contiguous()
            alphas = torch.empty(b, 1, h * w, device=ann.device); z = torch.empty(b, c, device=ann.device); xz = torch.empty_like(z)
            lengths = torch.ones(b, dtype=torch.int32, device=ann.device)
            _check(_sat.sat_attention_step_fwd(_p(ann), _p(U), _p(hc), hc.shape[1], _p(self.f_att.weight), _p(lengths), 0,
                                               _p(alphas), 1, _p(z), _p(xz), b, 1, h * w, c, A, _stream()))
            return z, alphas.reshape(b, h, w)

    g = np.load(os.path.join(golden_dir, "g1_attention.npz"))
    att = SoftAttention(12, 7, 9).cuda()
    with torch.no_grad():
        att.encoder_att.weight.copy_(torch.from_numpy(g["We"])); att.decoder_att.weight.copy_(torch.from_numpy(g["Wd"]))
        att.f_att.weight.copy_(torch.from_numpy(g["wf"]))
        z, alpha = att(torch.from_numpy(g["ann"]).cuda(), torch.from_numpy(g["hid"]).cuda())
    torch.cuda.synchronize()
    close(z, g["z"], what="z"); close(alpha, g["alpha"], what="alpha")
