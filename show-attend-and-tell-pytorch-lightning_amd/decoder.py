"""Host side of the decoder hot path: packing plan, workspace and the
``torch.autograd.Function`` wrappers that hand raw device pointers to
``libsat_hip.so``.  Mirrors the decoder half of ``SAT.train_batch``
(reference model.py:487-557) and the loss lines model.py:592-597.
"""
import collections
import ctypes as C

import numpy as np
import torch

from . import _lib as L


def _upload(t, dev):
    if dev.type != "cuda":
        return t.to(dev)
    return t.pin_memory().to(dev, non_blocking=True)


class PackPlan:
    """Index maps of ``pack_padded_sequence(x, lengths, batch_first=True, enforce_sorted=False)``
    (model.py:553-554) built once per batch on the host, as the reference does through
    ``lengths.tolist()``: batch order = ``torch.sort(lengths, descending=True)``, rows time-major.
    """

    def __init__(self, lengths, T, device):
        lens = torch.as_tensor(lengths, dtype=torch.int64, device="cpu").reshape(-1)
        self.N, self.T, self.T1 = lens.numel(), int(T), int(T) - 1
        if (lens < 0).any() or (lens > self.T1).any():
            raise ValueError("caption lengths must lie in [0, T-1]")
        _, order = torch.sort(lens, descending=True)          # the call pack_padded_sequence makes
        self.sorted_indices = order
        order_np, lens_np = order.numpy(), lens.numpy()
        self.max_len = int(lens_np.max()) if self.N else 0
        prow = np.full((self.T1, self.N), -1, np.int32)
        offsets = np.zeros(self.T1 + 1, np.int32)
        src, batch_sizes = [], []
        off = 0
        for t in range(self.T1):
            k = int((lens_np > t).sum())
            offsets[t] = off
            if k:
                rows = order_np[:k]
                prow[t, rows] = off + np.arange(k, dtype=np.int32)
                src.append(t * self.N + rows.astype(np.int32))
                batch_sizes.append(k)
            off += k
        offsets[self.T1] = off
        self.P = off
        self.batch_sizes = torch.tensor(batch_sizes, dtype=torch.int64)
        self.src_row_np = np.concatenate(src).astype(np.int32) if src else np.zeros(0, np.int32)
        self.offsets_host = np.ascontiguousarray(offsets)
        self.lengths_cpu = lens
        dev = torch.device(device)
        # pinned staging + asynchronous copies: a copy out of pageable memory blocks the host until the stream has drained, i.e. one
        # host/GPU synchronisation per train step (the host then issues the rest of the step into an empty queue)
        self.prow = _upload(torch.from_numpy(prow), dev)
        self.src_row = _upload(torch.from_numpy(self.src_row_np), dev)
        self.lengths = _upload(lens.to(torch.int32), dev)
        unsorted = torch.empty_like(order); unsorted[order] = torch.arange(self.N)
        self.unsorted_indices = unsorted                     # PackedSequence's index pair (model.py:553-554), host and device
        self.sorted_indices_dev, self.unsorted_indices_dev = _upload(order, dev), _upload(unsorted, dev)

    _cache = collections.OrderedDict()

    @classmethod
    def cached(cls, lengths, T, device):
        """Plans are read-only: batches with the same lengths (bucketed sampling repeats them often) share one."""
        lens = torch.as_tensor(lengths, dtype=torch.int64, device="cpu").reshape(-1)
        key = (int(T), str(torch.device(device)), lens.numpy().tobytes())
        plan = cls._cache.get(key)
        if plan is None:
            plan = cls._cache[key] = cls(lens, T, device)
            if len(cls._cache) > 32:
                cls._cache.popitem(last=False)
        else:
            cls._cache.move_to_end(key)
        return plan

    def pack(self, x_ntx):
        """(N, T-1, ...) padded -> packed rows (P, ...), same order as the library writes."""
        flat = x_ntx.transpose(0, 1).reshape(self.T1 * self.N, *x_ntx.shape[2:])
        return flat.index_select(0, self.src_row.to(torch.int64))

    def teacher_flags(self, epsilon, draw=None):
        """Scheduled sampling decisions (model.py:518): steps 0..2 always feed the caption; later steps
        draw one uniform sample each -- from the CPU generator, like the reference (F7) -- while any
        caption is still running."""
        if draw is None:
            draw = lambda: float(torch.rand(1))
        flags = np.ones(self.T1, np.int32)
        for step in range(min(self.T1, self.max_len)):
            if step <= 2 or draw() <= float(epsilon):
                flags[step] = 1
            else:
                flags[step] = 0
        return flags


def decoder_dims(B, R, T, Lc, D, A, m, n, V, P, deep, padding_idx, precision=0, embed_max_norm=0.0, dropout=0.0,
                 embedding_dropout=0.0, dropout_seed=0, layers=1):
    if not 1 <= int(layers) <= L.MAX_LSTM_LAYERS:
        raise ValueError("decoder_layers=%d: the library is built for 1..%d stacked LSTM layers" % (layers, L.MAX_LSTM_LAYERS))
    return L.DecoderDims(B=B, R=R, T=T, L=Lc, D=D, A=A, m=m, n=n, V=V, P=P, deep_output=int(bool(deep)), padding_idx=padding_idx,
                         precision=int(precision), layers=int(layers), embed_max_norm=float(embed_max_norm or 0.0), dropout=float(dropout),
                         embedding_dropout=float(embedding_dropout), dropout_seed=int(dropout_seed))


def _params_struct(tensors, layers=1):
    s = L.DecoderParams()
    for k in L.PARAM_FIELDS:
        t = tensors.get(k)
        setattr(s, k, None if t is None else t.data_ptr())
    for k in L.UP_FIELDS:
        arr = getattr(s, k)
        for l in range(1, layers):
            arr[l - 1] = tensors["%s_l%d" % (k[3:], l)].data_ptr()
    return s


def layers_of(n_params):
    """number of LSTM layers implied by the length of a decoder parameter list"""
    extra = n_params - len(L.PARAM_FIELDS)
    if extra < 0 or extra % len(L.UP_FIELDS):
        raise ValueError("decoder parameter list has %d entries; expected 18 + 4 per stacked layer" % n_params)
    return 1 + extra // len(L.UP_FIELDS)


def _check_param(name, t, shape):
    if t is None:
        return
    if tuple(t.shape) != tuple(shape):
        raise ValueError("decoder parameter %s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError("decoder parameter %s must be contiguous fp32" % name)


class DecoderTrainFn(torch.autograd.Function):
    """logits_packed (P,V), alphas (N,T-1,L) = decoder(ann (B,L,D), captions) with BPTT backward."""

    @staticmethod
    def forward(ctx, ann, caps_i32, plan, teacher, deep, padding_idx, R, precision, embed_max_norm, dropout, *params):
        lib = L.lib()
        L.require_gpu(ann, caps_i32, *[p for p in params if p is not None])
        layers = layers_of(len(params))
        names = L.param_names(layers)
        tens = dict(zip(names, params))
        ann = ann.contiguous()
        B, Lc, D = ann.shape
        V, m = tens["embedding"].shape
        n = tens["w_hh"].shape[1]
        A = tens["att_dec"].shape[0]
        N, T = caps_i32.shape
        if N != B * R or plan.N != N or plan.T != T:
            raise ValueError("caption batch (%d,%d) does not match B*R=%d / plan (%d,%d)" % (N, T, B * R, plan.N, plan.T))
        shapes = dict(embedding=(V, m), init_f_w=(m, D), init_f_b=(m,), init_i_w=(2 * n * layers, m), init_i_b=(2 * n * layers,), w_ih=(4 * n, m + D),
                      w_hh=(4 * n, n), b_ih=(4 * n,), b_hh=(4 * n,), att_enc=(A, D), att_dec=(A, n), att_f=(1, A), beta_w=(D, n),
                      beta_b=(D,), out_hidden=(m, n), out_context=(m, D), out_w=(V, m), out_b=(V,))
        for l in range(1, layers):
            shapes.update({"w_ih_l%d" % l: (4 * n, n), "w_hh_l%d" % l: (4 * n, n), "b_ih_l%d" % l: (4 * n,), "b_hh_l%d" % l: (4 * n,)})
        for k in names:
            _check_param(k, tens[k], shapes[k])
            if tens[k] is None and k not in ("out_context", "out_b"):
                raise ValueError("decoder parameter %s is missing" % k)
        dims = decoder_dims(B, R, T, Lc, D, A, m, n, V, plan.P, deep, padding_idx, precision, embed_max_norm, *dropout, layers=layers)
        ws_bytes = lib.sat_decoder_workspace_bytes(C.byref(dims))
        if ws_bytes == 0:
            raise L.SatHipError("sat_decoder_workspace_bytes: %s" % lib.sat_last_error().decode())
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=ann.device)
        logits = torch.empty(max(plan.P, 1), V, dtype=torch.float32, device=ann.device)[:plan.P]
        alphas = torch.empty(N, T - 1, Lc, dtype=torch.float32, device=ann.device)
        teacher = np.ascontiguousarray(teacher, np.int32)
        batch = L.DecoderBatch(ann=ann.data_ptr(), caps=caps_i32.data_ptr(), lengths=plan.lengths.data_ptr(), prow=plan.prow.data_ptr(),
                               src_row=plan.src_row.data_ptr(), step_offsets_host=plan.offsets_host.ctypes.data,
                               teacher_host=teacher.ctypes.data)
        w = _params_struct(tens, layers)
        L.check(lib.sat_decoder_train_fwd(C.byref(dims), C.byref(w), C.byref(batch), L.ptr(logits), L.ptr(alphas), L.ptr(ws), ws_bytes,
                                          L.stream_ptr()), "sat_decoder_train_fwd")
        ctx.dims, ctx.plan, ctx.teacher, ctx.ws, ctx.ws_bytes = dims, plan, teacher, ws, ws_bytes
        ctx.caps, ctx.layers = caps_i32, layers
        ctx.save_for_backward(ann, alphas, *[p for p in params if p is not None])
        ctx.present = [p is not None for p in params]
        return logits, alphas

    @staticmethod
    def backward(ctx, dlogits, dalphas):
        lib = L.lib()
        saved = list(ctx.saved_tensors)
        ann, alphas = saved[0], saved[1]
        it = iter(saved[2:])
        params = [next(it) if present else None for present in ctx.present]
        names = L.param_names(ctx.layers)
        tens = dict(zip(names, params))
        plan, dims = ctx.plan, ctx.dims
        taken = set()

        def gbuf(t):                  # bucket slice of the data-parallel exchange when there is one; a tied weight appears twice: once only
            if id(t) in taken:
                return torch.empty_like(t)
            taken.add(id(t))
            return L.grad_buffer(t)

        grads = {k: (None if t is None else gbuf(t)) for k, t in tens.items()}
        dann = torch.empty_like(ann)
        if dlogits is None:
            dlogits = torch.zeros(plan.P, dims.V, dtype=torch.float32, device=ann.device)
        dlogits = dlogits.contiguous()
        dalphas = None if dalphas is None else dalphas.contiguous()
        batch = L.DecoderBatch(ann=ann.data_ptr(), caps=ctx.caps.data_ptr(), lengths=plan.lengths.data_ptr(), prow=plan.prow.data_ptr(),
                               src_row=plan.src_row.data_ptr(), step_offsets_host=plan.offsets_host.ctypes.data,
                               teacher_host=ctx.teacher.ctypes.data)
        w, g = _params_struct(tens, ctx.layers), _params_struct(grads, ctx.layers)
        L.check(lib.sat_decoder_train_bwd(C.byref(dims), C.byref(w), C.byref(batch), L.ptr(dlogits), L.ptr(alphas), L.ptr(dalphas),
                                          C.byref(g), L.ptr(dann), L.ptr(ctx.ws), ctx.ws_bytes, L.stream_ptr()), "sat_decoder_train_bwd")
        return (dann, None, None, None, None, None, None, None, None, None, *[grads[k] for k in names])


class LabelSmoothingFn(torch.autograd.Function):
    """util.py:105-112 on packed rows; returns (loss, accuracy) -- accuracy as in model.py:596-597."""

    @staticmethod
    def forward(ctx, logits, targets_i32, smoothing):
        lib = L.lib()
        L.require_gpu(logits, targets_i32)
        logits = logits.contiguous()
        P, V = logits.shape
        lse = torch.empty(P, dtype=torch.float32, device=logits.device)
        rows = torch.empty(P, dtype=torch.float32, device=logits.device)
        correct = torch.empty(P, dtype=torch.int32, device=logits.device)
        out = torch.empty(2, dtype=torch.float32, device=logits.device)
        L.check(lib.sat_ce_label_smooth_fwd(L.ptr(logits), L.ptr(targets_i32), P, V, float(smoothing), L.ptr(lse), L.ptr(rows),
                                            L.ptr(correct), L.ptr(out), L.stream_ptr()), "sat_ce_label_smooth_fwd")
        ctx.save_for_backward(logits, targets_i32, lse)
        ctx.smoothing = float(smoothing)
        loss, acc = out[0].clone(), out[1].clone()
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    def backward(ctx, gloss, _gacc):
        lib = L.lib()
        logits, targets, lse = ctx.saved_tensors
        P, V = logits.shape
        dlogits = torch.empty_like(logits)
        g = gloss.reshape(1).to(torch.float32).contiguous()
        L.check(lib.sat_ce_label_smooth_bwd(L.ptr(logits), L.ptr(targets), L.ptr(lse), P, V, ctx.smoothing, L.ptr(g), L.ptr(dlogits),
                                            L.stream_ptr()), "sat_ce_label_smooth_bwd")
        return dlogits, None, None


class DoublyStochasticFn(torch.autograd.Function):
    """model.py:594: gamma * ((1 - alphas.sum(dim=1)) ** 2).mean(), fixed-order reduction."""

    @staticmethod
    def forward(ctx, alphas, gamma):
        lib = L.lib()
        L.require_gpu(alphas)
        alphas = alphas.contiguous()
        N, T1, Lc = alphas.shape
        asum = torch.empty(N, Lc, dtype=torch.float32, device=alphas.device)
        part = torch.empty((N * Lc + 255) // 256, dtype=torch.float32, device=alphas.device)
        out = torch.empty(1, dtype=torch.float32, device=alphas.device)
        L.check(lib.sat_doubly_stochastic_fwd(L.ptr(alphas), N, T1, Lc, float(gamma), L.ptr(asum), L.ptr(part), L.ptr(out),
                                              L.stream_ptr()), "sat_doubly_stochastic_fwd")
        ctx.save_for_backward(asum)
        ctx.shape, ctx.gamma = (N, T1, Lc), float(gamma)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        lib = L.lib()
        (asum,) = ctx.saved_tensors
        N, T1, Lc = ctx.shape
        dal = torch.empty(N, T1, Lc, dtype=torch.float32, device=asum.device)
        gs = g.reshape(1).to(torch.float32).contiguous()
        L.check(lib.sat_doubly_stochastic_bwd(L.ptr(asum), L.ptr(gs), N, T1, Lc, ctx.gamma, L.ptr(dal), L.stream_ptr()),
                "sat_doubly_stochastic_bwd")
        return dal, None


def dropout_scale_reference(seed, stream, idx, p):
    """numpy replica of csrc/decoder_kernels.h:dropout_scale (tests rebuild the masks with it on the host)."""
    M = (1 << 64) - 1
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (np.uint64(seed & M) + np.uint64((0x9E3779B97F4A7C15 * (stream + 1)) & M) + idx * np.uint64(0xD1342543DE82EF95))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return np.where(u >= np.float32(p), np.float32(1.0) / (np.float32(1.0) - np.float32(p)), np.float32(0.0)).astype(np.float32)


def gemm(A, B, *, amode=0, bmode=0, M=None, N=None, K=None, out=None, accumulate=False, epi=0, bias=None, e0=None, c0=0, c1=0,
         a_rows=None, c_rows=None, out_rows=None, slab=None, bf16_mfma=False, out_dtype=torch.float32):
    """Thin test/driver entry to sat_gemm_f32 (dense modes).  A: (M,K) if amode=0 else (K,M); B: (N,K) if bmode=0 else (K,N)."""
    lib = L.lib()
    L.require_gpu(A, B)
    if M is None:
        M = (a_rows.numel() if a_rows is not None else A.shape[0]) if amode == 0 else A.shape[1]
    if K is None:
        K = A.shape[1] if amode == 0 else A.shape[0]
    if N is None:
        N = B.shape[0] if bmode == 0 else B.shape[1]
    if out is None:
        out = torch.zeros(out_rows if out_rows is not None else M, N, dtype=out_dtype, device=A.device)
    d = L.GemmDesc(A=A.data_ptr(), lda=A.stride(0), a_rows=None if a_rows is None else a_rows.data_ptr(),
                   B=B.data_ptr(), ldb=B.stride(0), C=out.data_ptr(), ldc=out.stride(0),
                   c_rows=None if c_rows is None else c_rows.data_ptr(), M=M, N=N, K=K, amode=amode, bmode=bmode,
                   accumulate=int(accumulate), epi=epi, bias=None if bias is None else bias.data_ptr(),
                   e0=None if e0 is None else e0.data_ptr(), lde0=0 if e0 is None else e0.stride(0), c0=c0, c1=c1,
                   slab=None if slab is None else slab.data_ptr(), slab_elems=0 if slab is None else slab.numel())
    bf = torch.bfloat16
    if bf16_mfma or A.dtype == bf or B.dtype == bf or out.dtype == bf:
        t = L.GemmTypes(a_bf16=int(A.dtype == bf), b_bf16=int(B.dtype == bf), c_bf16=int(out.dtype == bf), bf16_mfma=int(bf16_mfma))
        L.check(lib.sat_gemm_ex(C.byref(d), C.byref(t), L.stream_ptr()), "sat_gemm_ex")
    else:
        L.check(lib.sat_gemm_f32(C.byref(d), L.stream_ptr()), "sat_gemm_f32")
    return out
