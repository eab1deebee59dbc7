"""The reference's decoder sub-modules as callable modules ("sub-module call signatures the C ABI must serve",
rows a2-a7): ``InitLSTM.forward`` (model.py:76-81), ``SoftAttention.forward`` (model.py:94-109), ``DeepOutput.forward``
(model.py:125-131), the beta gate ``Sequential(Linear, Sigmoid)`` (model.py:187-192), ``nn.Embedding`` (model.py:158-164) and
``nn.LSTM`` called one step at a time (model.py:175-180; calls at 326, 544) -- same constructor arguments, parameter names and
state-dict keys, with ``forward`` on the step entry points of ``libsat_hip.so`` (include/sat_hip.h) under
``torch.autograd.Function`` wrappers.  Exact fp32 MFMA.  The fused whole-loop path (``SAT.train_batch``) does not go through
these: they serve callers that drive the sub-modules themselves (the reference's own ``forward`` / notebooks do).
GPU tensors only: like everywhere in this package there is no CPU fallback."""
import ctypes as C

import torch
from torch import nn

from . import _lib as L


def _f32(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


def _empty(*shape, like):
    return torch.empty(*shape, dtype=torch.float32, device=like.device)


def _nld(ann):
    """(N, D, h, w) annotations as the reference passes them -> (N, L, D) row-major (free for this package's NHWC encoder output)"""
    N, D, h, w = ann.shape
    return _f32(ann.permute(0, 2, 3, 1).reshape(N, h * w, D)), (h, w)


def _seed(p, training):
    if not training or p == 0.0:
        return 0.0, 0
    return float(p), int(torch.randint(0, 2 ** 62, (1,)))


def _gemm(A, B, out, amode=0, bmode=0, M=None, N=None, K=None, accumulate=False, epi=0, bias=None, c0=0, c1=0):
    from .decoder import gemm
    return gemm(A, B, amode=amode, bmode=bmode, M=M, N=N, K=K, out=out, accumulate=accumulate, epi=epi, bias=bias, c0=c0, c1=c1)


# ----------------------------------------------------------------------------- InitLSTM (model.py:66-81)
class _InitLSTMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ann, w_f, b_f, w_i, b_i, p, seed):
        L.require_gpu(ann, w_f, b_f, w_i, b_i)
        N, Lc, D = ann.shape
        m, n2 = w_f.shape[0], w_i.shape[0]
        mean, f, init = _empty(N, D, like=ann), _empty(N, m, like=ann), _empty(N, n2, like=ann)
        L.check(L.lib().sat_init_lstm_fwd(L.ptr(ann), L.ptr(w_f), L.ptr(b_f), L.ptr(w_i), L.ptr(b_i), p, seed, L.ptr(mean), L.ptr(f), L.ptr(init),
                                          N, Lc, D, m, n2, L.stream_ptr()), "sat_init_lstm_fwd")
        ctx.save_for_backward(mean, f, w_f, w_i)
        ctx.dims, ctx.drop = (N, Lc, D, m, n2), (p, seed)
        return init

    @staticmethod
    def backward(ctx, dinit):
        mean, f, w_f, w_i = ctx.saved_tensors
        N, Lc, D, m, n2 = ctx.dims
        dinit = _f32(dinit)
        dw_f, db_f, dw_i, db_i = torch.empty_like(w_f), _empty(m, like=f), torch.empty_like(w_i), _empty(n2, like=f)
        dann, df, dmean = _empty(N, Lc, D, like=f), _empty(N, m, like=f), _empty(N, D, like=f)
        scratch = _empty(((N + 255) // 256) * max(n2, m), like=f)
        L.check(L.lib().sat_init_lstm_bwd(L.ptr(dinit), L.ptr(mean), L.ptr(f), L.ptr(w_f), L.ptr(w_i), ctx.drop[0], ctx.drop[1], L.ptr(dw_f), L.ptr(db_f),
                                          L.ptr(dw_i), L.ptr(db_i), L.ptr(dann), L.ptr(df), L.ptr(dmean), L.ptr(scratch), N, Lc, D, m, n2, L.stream_ptr()),
                "sat_init_lstm_bwd")
        return dann, dw_f, db_f, dw_i, db_i, None, None


class InitLSTM(nn.Module):
    """model.py:66-81.  ``forward(annotations (N, D, h, w))`` -> ``(init_h, init_c)``, each ``(layers, N, n)``: the reference's raw
    ``reshape`` without a permute is reproduced as a reinterpretation of the (N, 2 n layers) result."""

    def __init__(self, args, bias=True):
        super().__init__()
        self.decoder_dim, self.decoder_layers = args.decoder_dim, args.decoder_layers
        self.factorize = nn.Linear(args.encoder_dim, args.embed_dim, bias=bias)
        self.init = nn.Linear(args.embed_dim, 2 * args.decoder_dim * args.decoder_layers, bias=bias)
        self.dropout = nn.Dropout(p=args.dropout)

    def forward(self, annotations):
        ann, _ = _nld(annotations)
        p, seed = _seed(self.dropout.p, self.training)
        init = _InitLSTMFn.apply(ann, self.factorize.weight, self.factorize.bias, self.init.weight, self.init.bias, p, seed)
        init = init.reshape(2 * self.decoder_layers, ann.shape[0], self.decoder_dim)          # model.py:79
        return init[:self.decoder_layers], init[self.decoder_layers:]


# ----------------------------------------------------------------------------- SoftAttention (model.py:84-109)
class _AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ann, hid, w_e, w_d, w_f):
        L.require_gpu(ann, hid, w_e, w_d, w_f)
        lib = L.lib()
        N, Lc, D = ann.shape
        A, n = w_d.shape
        U = _empty(N, Lc, A, like=ann)
        L.check(lib.sat_attention_precompute(L.ptr(ann), L.ptr(w_e), L.ptr(U), N, Lc, D, A, L.stream_ptr()), "sat_attention_precompute")
        hc = torch.ones(N, A + D, dtype=torch.float32, device=ann.device)       # [q | gate]: the module has no gate, so the gate columns are 1
        _gemm(hid, w_d, hc, M=N, N=A, K=n)                                       # q = h W_d^T into hc[:, :A]
        live = torch.ones(N, dtype=torch.int32, device=ann.device)
        alpha, Z, XZ = _empty(N, Lc, like=ann), _empty(N, D, like=ann), _empty(N, D, like=ann)
        L.check(lib.sat_attention_step_fwd(L.ptr(ann), L.ptr(U), L.ptr(hc), A + D, L.ptr(w_f), L.ptr(live), 0, L.ptr(alpha), 1, L.ptr(Z), L.ptr(XZ),
                                           N, 1, Lc, D, A, L.stream_ptr()), "sat_attention_step_fwd")
        ctx.save_for_backward(ann, hid, w_e, w_d, w_f, U, hc, alpha, Z, live)
        return Z, alpha

    @staticmethod
    def backward(ctx, dZ, dalpha):
        ann, hid, w_e, w_d, w_f, U, hc, alpha, Z, live = ctx.saved_tensors
        lib = L.lib()
        N, Lc, D = ann.shape
        A, n = w_d.shape
        dev = ann.device
        dZ = torch.zeros(N, D, dtype=torch.float32, device=dev) if dZ is None else _f32(dZ)
        dalpha = None if dalpha is None else _f32(dalpha)
        zero = torch.zeros(N, D, dtype=torch.float32, device=dev)                # no gated-context consumer here
        DZ, dhc = _empty(N, D, like=ann), _empty(N, A + D, like=ann)
        dU = torch.zeros(N, Lc, A, dtype=torch.float32, device=dev)
        dwf_part = torch.zeros(N, A, dtype=torch.float32, device=dev)
        da = _empty(N, Lc, like=ann)
        L.check(lib.sat_attention_step_bwd(L.ptr(ann), L.ptr(U), L.ptr(hc), A + D, L.ptr(w_f), L.ptr(live), 0, L.ptr(alpha), L.ptr(dalpha), 1, L.ptr(Z),
                                           L.ptr(dZ), L.ptr(zero), L.ptr(DZ), L.ptr(dhc), A + D, L.ptr(dU), L.ptr(dwf_part), L.ptr(da), N, 1, Lc, D, A,
                                           L.stream_ptr()), "sat_attention_step_bwd")
        dann = _empty(N, Lc, D, like=ann)
        L.check(lib.sat_attention_context_bwd(L.ptr(alpha), L.ptr(DZ), L.ptr(live), L.ptr(dann), 0, N, 1, 1, Lc, D, L.stream_ptr()), "sat_attention_context_bwd")
        _gemm(dU.view(N * Lc, A), w_e, dann.view(N * Lc, D), bmode=1, M=N * Lc, N=D, K=A, accumulate=True)          # + dU W_e
        dw_e = torch.empty_like(w_e); dw_d = torch.empty_like(w_d); dw_f = torch.empty_like(w_f)
        _gemm(dU.view(N * Lc, A), ann.view(N * Lc, D), dw_e, amode=1, bmode=1, M=A, N=D, K=N * Lc)
        dq = dhc[:, :A]                                                          # strided view: row stride A + D
        dhid = _empty(N, n, like=ann)
        _gemm(dq, w_d, dhid, bmode=1, M=N, N=n, K=A)
        _gemm(dq, hid, dw_d, amode=1, bmode=1, M=A, N=n, K=N)
        scratch = _empty(((N + 255) // 256) * A, like=ann)
        L.check(lib.sat_colsum(L.ptr(dwf_part), A, N, A, L.ptr(dw_f), L.ptr(scratch), L.stream_ptr()), "sat_colsum")
        return dann, dhid, dw_e, dw_d, dw_f


class SoftAttention(nn.Module):
    """model.py:84-109.  ``forward(annotations (N, D, h, w), decoder_hidden (N, n))`` -> ``(zt (N, D), alpha (N, h, w))``."""

    def __init__(self, args):
        super().__init__()
        self.encoder_att = nn.Linear(args.encoder_dim, args.attention_dim, bias=False)
        self.decoder_att = nn.Linear(args.decoder_dim, args.attention_dim, bias=False)
        self.f_att = nn.Linear(args.attention_dim, 1, bias=False)

    def forward(self, annotations, decoder_hidden):
        ann, (h, w) = _nld(annotations)
        z, alpha = _AttentionFn.apply(ann, _f32(decoder_hidden), self.encoder_att.weight, self.decoder_att.weight, self.f_att.weight)
        return z, alpha.reshape(ann.shape[0], h, w)


# ----------------------------------------------------------------------------- DeepOutput (model.py:112-131)
class _DeepOutputFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prev_embed, hidden, context, w_h, w_c, w_o, b_o, p, seed):
        L.require_gpu(hidden, w_h, w_o)
        deep = w_c is not None
        N, n = hidden.shape
        V, m = w_o.shape
        D = context.shape[1] if deep else 0
        u = _empty(N, m, like=hidden)
        udrop = _empty(N, m, like=hidden) if p > 0 else None
        logits = _empty(N, V, like=hidden)
        L.check(L.lib().sat_deep_output_fwd(L.ptr(prev_embed if deep else None), L.ptr(hidden), L.ptr(context if deep else None), L.ptr(w_h), L.ptr(w_c), L.ptr(w_o),
                                            L.ptr(b_o), p, seed, L.ptr(u), L.ptr(udrop), L.ptr(logits), N, m, n, D, V, L.stream_ptr()), "sat_deep_output_fwd")
        ctx.save_for_backward(hidden, context if deep else None, u, udrop, w_h, w_c, w_o)
        ctx.cfg = (deep, N, m, n, D, V, p, seed, b_o is not None, prev_embed is not None, context is not None)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        hidden, context, u, udrop, w_h, w_c, w_o = ctx.saved_tensors
        deep, N, m, n, D, V, p, seed, has_b, has_e, has_c = ctx.cfg
        dlogits = _f32(dlogits)
        d_e, d_h = _empty(N, m, like=hidden), _empty(N, n, like=hidden)
        d_c = _empty(N, D, like=hidden) if deep else None
        dw_h, dw_o = torch.empty_like(w_h), torch.empty_like(w_o)
        dw_c = torch.empty_like(w_c) if deep else None
        db_o = _empty(V, like=hidden) if has_b else None
        scratch = _empty(((N + 255) // 256) * V, like=hidden)
        L.check(L.lib().sat_deep_output_bwd(L.ptr(dlogits), L.ptr(hidden), L.ptr(context), L.ptr(u), L.ptr(udrop), L.ptr(w_h), L.ptr(w_c), L.ptr(w_o), p, seed,
                                            L.ptr(d_e), L.ptr(d_h), L.ptr(d_c), L.ptr(dw_h), L.ptr(dw_c), L.ptr(dw_o), L.ptr(db_o), L.ptr(scratch),
                                            N, m, n, D, V, L.stream_ptr()), "sat_deep_output_bwd")
        return (d_e if (deep and has_e) else None), d_h, (d_c if (deep and has_c) else None), dw_h, dw_c, dw_o, db_o, None, None


class DeepOutput(nn.Module):
    """model.py:112-131.  ``forward(prev_embed (N, m), hidden (N, n), context (N, D))`` -> logits (N, V)."""

    def __init__(self, args):
        super().__init__()
        self.deep = args.deep_output
        self.dropout = nn.Dropout(p=args.dropout)
        self.hidden = nn.Linear(args.decoder_dim, args.embed_dim, bias=False)
        if self.deep:
            self.context = nn.Linear(args.encoder_dim, args.embed_dim, bias=False)
        self.output = nn.Linear(args.embed_dim, args.vocab_size, bias=(not args.weight_tying))

    def forward(self, prev_embed, hidden, context):
        p, seed = _seed(self.dropout.p, self.training)
        if self.deep:
            return _DeepOutputFn.apply(_f32(prev_embed), _f32(hidden), _f32(context), self.hidden.weight, self.context.weight, self.output.weight,
                                       self.output.bias, p, seed)
        return _DeepOutputFn.apply(None, _f32(hidden), None, self.hidden.weight, None, self.output.weight, self.output.bias, p, seed)


# ----------------------------------------------------------------------------- beta gate (model.py:187-192)
class _GateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, w, b):
        L.require_gpu(h, w, b)
        N, n = h.shape
        D = w.shape[0]
        y = _empty(N, D, like=h)
        _gemm(h, w, y, M=N, N=D, K=n, epi=2, bias=b, c0=0, c1=D)                 # sigmoid(h W^T + b)
        ctx.save_for_backward(h, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        h, w, y = ctx.saved_tensors
        N, n = h.shape
        D = w.shape[0]
        dy = _f32(dy)
        dpre = torch.empty_like(y)
        L.check(L.lib().sat_sigmoid_bwd(L.ptr(dy), L.ptr(y), L.ptr(dpre), dpre.numel(), L.stream_ptr()), "sat_sigmoid_bwd")
        dh, dw, db = torch.empty_like(h), torch.empty_like(w), _empty(D, like=h)
        _gemm(dpre, w, dh, bmode=1, M=N, N=n, K=D)
        _gemm(dpre, h, dw, amode=1, bmode=1, M=D, N=n, K=N)
        scratch = _empty(((N + 255) // 256) * D, like=h)
        L.check(L.lib().sat_colsum(L.ptr(dpre), D, N, D, L.ptr(db), L.ptr(scratch), L.stream_ptr()), "sat_colsum")
        return dh, dw, db


class Gate(nn.Sequential):
    """``nn.Sequential(nn.Linear(n, D), nn.Sigmoid())`` of model.py:187-192 (state-dict keys ``0.weight`` / ``0.bias``); ``forward`` is one
    GEMM with the bias + sigmoid epilogue."""

    def forward(self, h):
        return _GateFn.apply(_f32(h), self[0].weight, self[0].bias)


# ----------------------------------------------------------------------------- embedding (model.py:158-164)
def _embedding_gather(tok, table, max_norm):
    """rows of `table` for the token ids (any shape); max_norm > 0 first renormalises the used rows of `table` in place"""
    L.require_gpu(tok, table)
    V, m = table.shape
    flat = tok.reshape(-1).to(torch.int32).contiguous()
    out = _empty(flat.numel(), m, like=table)
    flags = torch.empty(V, dtype=torch.int32, device=table.device) if max_norm else None
    L.check(L.lib().sat_embedding_fwd(L.ptr(table), L.ptr(flat), L.ptr(out), flat.numel(), V, m, float(max_norm or 0.0), L.ptr(flags), L.stream_ptr()),
            "sat_embedding_fwd")
    return out, flat


class _EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tok, table, padding_idx):
        out, flat = _embedding_gather(tok, table, None)
        ctx.save_for_backward(flat)
        ctx.cfg = (table.shape[0], table.shape[1], int(-1 if padding_idx is None else padding_idx))
        return out.reshape(*tok.shape, table.shape[1])

    @staticmethod
    def backward(ctx, dy):
        (flat,) = ctx.saved_tensors
        V, m, pad = ctx.cfg
        dy = _f32(dy).reshape(-1, m)
        dtable = _empty(V, m, like=dy)
        scratch = torch.empty(3 * V + 1 + flat.numel(), dtype=torch.int32, device=dy.device)
        L.check(L.lib().sat_embedding_bwd(L.ptr(dy), L.ptr(flat), L.ptr(dtable), flat.numel(), V, m, pad, L.ptr(scratch), L.stream_ptr()), "sat_embedding_bwd")
        return None, dtable, None


class Embedding(nn.Embedding):
    """``nn.Embedding(V, m, max_norm=embed_norm, padding_idx=<PAD>)`` of model.py:158-164 with the gather (and the in-place max-norm
    renormalisation) on the library's kernels; the gradient is accumulated in a fixed order."""

    def forward(self, tokens):
        if self.max_norm:                           # torch renormalises the used rows in place, outside autograd (embedding_renorm_)
            with torch.no_grad():
                _embedding_gather(tokens, self.weight.data, self.max_norm)
        return _EmbeddingFn.apply(tokens, self.weight, self.padding_idx)


# ----------------------------------------------------------------------------- nn.LSTM, one step at a time (model.py:175-180)
class _LSTMCellFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h, c, w_ih, w_hh, b_ih, b_hh):
        L.require_gpu(x, h, c, w_ih, w_hh, b_ih, b_hh)
        N, inp = x.shape
        n = h.shape[1]
        gates, hn, cn, bs = _empty(N, 4 * n, like=x), _empty(N, n, like=x), _empty(N, n, like=x), _empty(4 * n, like=x)
        L.check(L.lib().sat_lstm_cell_fwd(L.ptr(x), inp, L.ptr(h), L.ptr(c), L.ptr(w_ih), L.ptr(w_hh), L.ptr(b_ih), L.ptr(b_hh), L.ptr(gates), L.ptr(hn),
                                          L.ptr(cn), L.ptr(bs), N, n, L.stream_ptr()), "sat_lstm_cell_fwd")
        ctx.save_for_backward(x, h, c, cn, gates, w_ih, w_hh)
        return hn, cn

    @staticmethod
    def backward(ctx, dh, dc):
        x, h, c, cn, gates, w_ih, w_hh = ctx.saved_tensors
        N, inp = x.shape
        n = h.shape[1]
        dh = torch.zeros_like(h) if dh is None else _f32(dh)
        dc = None if dc is None else _f32(dc)
        dx, dhp, dcp = torch.empty_like(x), torch.empty_like(h), torch.empty_like(c)
        dw_ih, dw_hh, db_ih, db_hh = torch.empty_like(w_ih), torch.empty_like(w_hh), _empty(4 * n, like=x), _empty(4 * n, like=x)
        dg, scratch = _empty(N, 4 * n, like=x), _empty(((N + 255) // 256) * 4 * n, like=x)
        L.check(L.lib().sat_lstm_cell_bwd(L.ptr(x), inp, L.ptr(h), L.ptr(c), L.ptr(cn), L.ptr(gates), L.ptr(dh), L.ptr(dc), L.ptr(w_ih), L.ptr(w_hh), L.ptr(dx),
                                          L.ptr(dhp), L.ptr(dcp), L.ptr(dw_ih), L.ptr(dw_hh), L.ptr(db_ih), L.ptr(db_hh), L.ptr(dg), L.ptr(scratch), N, n,
                                          L.stream_ptr()), "sat_lstm_cell_bwd")
        return dx, dhp, dcp, dw_ih, dw_hh, db_ih, db_hh


class LSTM(nn.LSTM):
    """``nn.LSTM(input, hidden, num_layers, bias=True)`` of model.py:175-180 (torch's parameter names ``weight_ih_l{k}`` ...).
    ``forward(x (T, N, in), (h, c) (layers, N, n))`` runs the stacked cells step by step on ``sat_lstm_cell_fwd``: the reference
    only ever calls it with T = 1 (model.py:326, 544)."""

    def forward(self, x, state=None):
        if self.batch_first or self.bidirectional or self.proj_size or not self.bias or self.dropout:
            raise NotImplementedError("sat_amd.LSTM: only the configuration of model.py:175-180 (time-major, unidirectional, biased, no dropout)")
        T, N, _ = x.shape
        n, NL = self.hidden_size, self.num_layers
        if state is None:
            z = torch.zeros(NL, N, n, dtype=torch.float32, device=x.device)
            state = (z, z.clone())
        hs, cs = [_f32(state[0][l]) for l in range(NL)], [_f32(state[1][l]) for l in range(NL)]
        outs = []
        for t in range(T):
            inp = _f32(x[t])
            for l in range(NL):
                hs[l], cs[l] = _LSTMCellFn.apply(inp, hs[l], cs[l], getattr(self, "weight_ih_l%d" % l), getattr(self, "weight_hh_l%d" % l),
                                                 getattr(self, "bias_ih_l%d" % l), getattr(self, "bias_hh_l%d" % l))
                inp = hs[l]
            outs.append(inp)
        return torch.stack(outs, 0), (torch.stack(hs, 0), torch.stack(cs, 0))
