"""Drop-in mirror of the reference's ``model.py`` surface for the train-step hot path:
``get_encoder``, ``InitLSTM``, ``SoftAttention``, ``DeepOutput`` and ``SAT`` with the same
constructor kwargs, attribute names and state-dict keys (reference model.py:16-199),
so ``train.py``-style callers and checkpoints keep working.  The sub-modules (modules.py) hold the
parameters (created in the reference's order, so a given seed yields the same initial weights) and
are callable with the reference's signatures; ``train_batch`` runs the whole decode loop fused in
the library instead of calling them step by step.  All arithmetic goes through ``libsat_hip.so``.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F
from torch import nn

from torch.nn.utils.rnn import PackedSequence
from torch.optim.lr_scheduler import MultiStepLR, ReduceLROnPlateau, ExponentialLR, CosineAnnealingWarmRestarts, OneCycleLR

from . import _lib as L
from . import decoder as Dk
from .modules import DeepOutput, Embedding, Gate, InitLSTM, LSTM, SoftAttention  # noqa: F401  (the reference's module names, model.py:66-131)
from .encoder import get_encoder  # noqa: F401  (module-level name, as in the reference)

try:                                     # Lightning is optional plumbing (absent in the build image)
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:                        # pragma: no cover - exercised where Lightning is missing
    pl = None
    _Base = nn.Module


class _HParams(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    __setattr__ = dict.__setitem__


class LabelSmoothing(nn.Module):
    """util.py:91-112 on the HIP kernel; ``forward`` returns the loss, ``last_accuracy`` the
    argmax accuracy of the same rows (model.py:596-597) computed in the same pass."""

    def __init__(self, smoothing=0.0):
        super().__init__()
        self.confidence, self.smoothing = 1.0 - smoothing, smoothing
        self.last_accuracy = None

    def forward(self, x, target):
        loss, acc = Dk.LabelSmoothingFn.apply(x, target.to(torch.int32), self.smoothing)
        self.last_accuracy = acc
        return loss


class SATDecoder(nn.Module):
    """Everything of ``SAT`` except the encoder (model.py:146-199): the decoder parameters under
    the reference's names and the fused train-time decode."""

    def __init__(self, hp):
        super().__init__()
        self._build_decoder(hp)

    def _build_decoder(self, hp, encoder_factory=None):
        if isinstance(hp, dict) and not isinstance(hp, _HParams):
            hp = _HParams(hp)
        self.hp = hp
        if not 1 <= hp.decoder_layers <= L.MAX_LSTM_LAYERS:
            raise NotImplementedError("HIP decoder: decoder_layers=%d (the library stacks 1..%d LSTM layers)" % (hp.decoder_layers, L.MAX_LSTM_LAYERS))
        assert 0 <= hp.label_smoothing < (hp.vocab_size - 1) / hp.vocab_size
        self.criterion = LabelSmoothing(hp.label_smoothing)
        self.pad_idx = int(hp.vocab_stoi["<PAD>"])
        if encoder_factory is not None:                     # registered here to keep the reference's module order
            self.encoder = encoder_factory(hp)
        self.embedding = Embedding(hp.vocab_size, hp.embed_dim, max_norm=getattr(hp, "embed_norm", None), padding_idx=self.pad_idx)
        self.embedding_dropout = nn.Dropout(p=hp.embedding_dropout)
        self.init_lstm = InitLSTM(hp, bias=True)
        self.lstm = LSTM(input_size=hp.embed_dim + hp.encoder_dim, hidden_size=hp.decoder_dim, num_layers=hp.decoder_layers, bias=True)
        self.attention = SoftAttention(hp)
        self.beta = Gate(nn.Linear(hp.decoder_dim, hp.encoder_dim, bias=True), nn.Sigmoid())
        fan_in = self.beta[0].weight.shape[1]
        self.beta[0].bias.data.fill_(1 / fan_in)
        self.output = DeepOutput(hp)
        if hp.weight_tying and hp.deep_output:
            self.output.output.weight = self.embedding.weight

    # -- parameters in the order of sat_decoder_params (include/sat_hip.h): 18 base tensors, then 4 per stacked LSTM layer
    def param_list(self):
        o = self.output
        up = [getattr(self.lstm, "%s_l%d" % (k, l)) for l in range(1, self.hp.decoder_layers) for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        return [self.embedding.weight, self.init_lstm.factorize.weight, self.init_lstm.factorize.bias, self.init_lstm.init.weight,
                self.init_lstm.init.bias, self.lstm.weight_ih_l0, self.lstm.weight_hh_l0, self.lstm.bias_ih_l0, self.lstm.bias_hh_l0,
                self.attention.encoder_att.weight, self.attention.decoder_att.weight, self.attention.f_att.weight,
                self.beta[0].weight, self.beta[0].bias, o.hidden.weight, (o.context.weight if o.deep else None),
                o.output.weight, o.output.bias] + up

    def load_decoder_state(self, sd):
        own = self.state_dict()
        for k, v in sd.items():
            if k in own:
                own[k].copy_(torch.as_tensor(v))

    # ------------------------------------------------------------------ inference (model.py:214-472)
    def _params_struct(self):
        layers = self.hp.decoder_layers
        tens = dict(zip(L.param_names(layers), self.param_list()))
        return Dk._params_struct(tens, layers), tens

    @torch.no_grad()
    def beam_decode(self, ann_bld, hw, beamk=3, max_gen_length=32, temperature=1.0, sample_method="beam", sample_topk=3,
                    decoder_noise=None, rescore_method=None, rescore_reward=0.5, return_all=False, multinomial=None, randn=None):
        """SAT.forward's per-image beam search (model.py:260-472) on annotations (B, L, D).  The decode step,
        log-softmax / masking and top-k run in the library; the beam bookkeeping (which hypotheses to keep, finished
        lists, rescoring) stays on the host like in the reference.  ``sample_method`` "multinomial" / "topk"
        (model.py:360-379) draw the continuing hypotheses with ``multinomial(probs, k)`` and ``decoder_noise``
        (model.py:322-324) perturbs the recurrent state with ``randn(shape)``: both default to the torch samplers on the
        annotations' device and can be replaced (tests feed both sides the same draws)."""
        import ctypes as C
        assert sample_method in ("beam", "multinomial", "topk")
        multinomial = multinomial or torch.multinomial
        lib = L.lib()
        L.require_gpu(ann_bld)
        hp = self.hp
        dev = ann_bld.device
        B, Lc, D = ann_bld.shape
        Hh, Ww = hw
        V, m = self.embedding.weight.shape
        n = self.lstm.weight_hh_l0.shape[1]
        A = self.attention.decoder_att.weight.shape[0]
        START, PAD = int(hp.vocab_stoi["<START>"]), int(hp.vocab_stoi["<PAD>"])
        END, UNK = int(hp.vocab_stoi["<END>"]), int(hp.vocab_stoi["<UNK>"])
        temps = temperature if isinstance(temperature, list) else [temperature]
        NL = int(hp.decoder_layers)
        randn = randn or (lambda shape: torch.randn(shape, device=dev))
        dims = Dk.decoder_dims(1, beamk, 2, Lc, D, A, m, n, V, 0, hp.deep_output, self.pad_idx,
                               int(getattr(self, "sat_precision", "fp32") == "bf16"), layers=NL)
        w, _keep = self._params_struct()
        ws_bytes = lib.sat_decoder_infer_workspace_bytes(C.byref(dims), beamk)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        mask_rest = torch.tensor([START, PAD], dtype=torch.int32, device=dev)
        mask_first = torch.tensor([START, PAD, END, UNK], dtype=torch.int32, device=dev)
        work = torch.empty(beamk * V, **f32)
        captions, cap_scores, cap_alphas, cap_ppl = [], [], [], []
        st = L.stream_ptr
        for idx in range(B):
            k = beamk
            ann = ann_bld[idx].contiguous()
            h = torch.empty(NL, k, n, **f32); c = torch.empty(NL, k, n, **f32)
            L.check(lib.sat_decoder_infer_begin(C.byref(dims), C.byref(w), L.ptr(ann), k, beamk, L.ptr(h), L.ptr(c), L.ptr(ws), ws_bytes, st()),
                    "sat_decoder_infer_begin")
            top_preds = torch.full((1, k), START, dtype=torch.int64, device=dev)
            top_scores = torch.zeros(k, **f32)
            alphas = torch.zeros(1, k, Lc, **f32)
            fin_caps, fin_alphas, fin_scores, fin_ppl = [], [], [], []

            def rescore(s, step):
                if rescore_method == "LN":
                    return s / step
                if rescore_method == "WR":
                    return s + rescore_reward * step
                if rescore_method == "BAR":
                    return s + rescore_reward * (-top_scores.mean())
                return s

            step = 0
            while True:
                T = float(temps[step % len(temps)])
                tok = top_preds[step].to(torch.int32).contiguous()
                logits = torch.empty(k, V, **f32); alpha = torch.empty(k, Lc, **f32)
                noise = None
                if decoder_noise is not None and decoder_noise != 0.0:
                    noise = (randn((NL, k, n)).to(device=dev, dtype=torch.float32) * (decoder_noise / (step + 1))).contiguous()
                L.check(lib.sat_decoder_infer_step(C.byref(dims), C.byref(w), L.ptr(ann), L.ptr(tok), k, beamk, L.ptr(h), L.ptr(c), L.ptr(logits),
                                                   L.ptr(alpha), L.ptr(noise), L.ptr(ws), ws_bytes, st()), "sat_decoder_infer_step")
                scores = torch.empty(k, V, **f32)
                vals = torch.empty(k, **f32); inds = torch.empty(k, dtype=torch.int32, device=dev)
                if step == 0:
                    L.check(lib.sat_beam_scores(L.ptr(logits), k, V, T, L.ptr(mask_first), 4, None, L.ptr(scores), st()), "sat_beam_scores")
                    L.check(lib.sat_topk(L.ptr(scores), L.ptr(work), V, k, L.ptr(vals), L.ptr(inds), st()), "sat_topk")       # row 0 only (model.py:343)
                    top_scores = vals
                    top_preds = torch.cat([top_preds, inds.to(torch.int64).unsqueeze(0)], 0)
                    alphas = torch.cat([alphas, alpha.unsqueeze(0)], 0)
                else:
                    L.check(lib.sat_beam_scores(L.ptr(logits), k, V, T, L.ptr(mask_rest), 2, L.ptr(top_scores.contiguous()), L.ptr(scores), st()),
                            "sat_beam_scores")
                    if sample_method == "beam":
                        L.check(lib.sat_topk(L.ptr(scores), L.ptr(work), k * V, k, L.ptr(vals), L.ptr(inds), st()), "sat_topk")
                        top_scores = vals
                        pred = inds.to(torch.int64)
                    else:
                        if sample_method == "multinomial":                                 # model.py:360-364
                            pred = multinomial(torch.softmax(20 * scores / step, dim=1).reshape(-1), k)
                        else:                                                              # model.py:365-379
                            _, cand = torch.topk(scores, sample_topk, dim=1)
                            cand = (cand + (torch.arange(k, device=dev) * V).unsqueeze(1)).reshape(-1)
                            choice = multinomial(torch.softmax(scores.reshape(-1)[cand] / step, dim=0), k)
                            pred = cand[choice.to(cand.device)]
                        pred = pred.to(device=dev, dtype=torch.int64)
                        top_scores = scores.reshape(-1)[pred]
                    keep = torch.div(pred, V, rounding_mode="floor")
                    word = torch.remainder(pred, V).unsqueeze(0)
                    top_preds = torch.cat([top_preds[:, keep], word], 0)
                    alphas = torch.cat([alphas[:, keep], alpha.unsqueeze(0)[:, keep]], 0)
                    h, c = h[:, keep].contiguous(), c[:, keep].contiguous()
                complete = top_preds[step + 1] == END
                done = complete.tolist()
                if any(done):
                    for i, flag in enumerate(done):
                        if flag:
                            fin_caps.append(top_preds[:, i][1:-1].tolist())
                            fin_alphas.append(alphas[:, i][1:-1].reshape(-1, Hh, Ww).cpu())
                            fin_scores.append(float(rescore(top_scores[i], step)))
                            fin_ppl.append(float(torch.exp(-top_scores[i] / step)))
                    inc = ~complete
                    top_preds, alphas, top_scores = top_preds[:, inc], alphas[:, inc], top_scores[inc]
                    h, c = h[:, inc].contiguous(), c[:, inc].contiguous()
                    k = int(inc.sum())
                    if k == 0:
                        break
                if step >= max_gen_length:
                    for i in range(top_preds.shape[1]):
                        fin_caps.append(top_preds[:, i][1:-1].tolist())
                        fin_alphas.append(alphas[:, i][1:-1].reshape(-1, Hh, Ww).cpu())
                        fin_scores.append(float(rescore(top_scores[i], step)))
                        fin_ppl.append(float(torch.exp(-top_scores[i] / step)))
                    break
                step += 1
            if return_all:
                order = [i for _, i in sorted([[fin_scores[i], i] for i in range(len(fin_scores))], reverse=True)]
                captions.append([fin_caps[i] for i in order]); cap_alphas.append([fin_alphas[i] for i in order])
                cap_scores.append([fin_scores[i] for i in order]); cap_ppl.append([fin_ppl[i] for i in order])
            else:
                best = fin_scores.index(max(fin_scores))
                captions.append(fin_caps[best]); cap_alphas.append(fin_alphas[best])
                cap_scores.append(fin_scores[best]); cap_ppl.append(fin_ppl[best])
        return captions, cap_scores, cap_alphas, cap_ppl

    @torch.no_grad()
    def beam_decode_batched(self, ann_bld, hw, beamk=3, max_gen_length=32, temperature=1.0, rescore_method=None, rescore_reward=0.5,
                            return_all=False, sample_method="beam", sample_topk=3, decoder_noise=None, seed=None, gumbel=None, normals=None, graph=False):
        """The same beam search as ``beam_decode`` ("beam" sampling, no decoder noise) for ALL images of the batch at once
        (SURVEY 8f row 2): one library call enqueues every decode step for the (B, beamk) hypothesis rows -- per-image top-k,
        completed hypotheses leaving their image's beam, cut at ``max_gen_length`` -- without a host round trip; the host reads
        the back-trace once and rebuilds the reference's four lists (model.py:449-472).
        ``sample_method`` "multinomial" / "topk" (model.py:360-379) and ``decoder_noise`` (model.py:322-324) run in the same call:
        the hypotheses are drawn on the device as the top k of log p + Gumbel noise (an ordered sample without replacement, like
        ``torch.multinomial``) from a counter-based generator seeded by ``seed`` (default: one draw from torch's CPU generator);
        ``gumbel`` / ``normals`` replace the generator by tables (layouts: include/sat_hip.h, sat_beam_sampling).
        ``graph=True`` ("beam" sampling without noise): the call's ~25 launches per decode step are captured once per (batch
        shape, beam, length, temperatures, weights) into a hipGraph over static buffers and replayed - the same kernels with the
        same arguments, one submission."""
        import ctypes as C
        import numpy as np
        assert sample_method in ("beam", "multinomial", "topk")
        lib = L.lib()
        L.require_gpu(ann_bld)
        hp = self.hp
        dev = ann_bld.device
        ann_bld = ann_bld.contiguous()
        B, Lc, D = ann_bld.shape
        Hh, Ww = hw
        V, m = self.embedding.weight.shape
        n = self.lstm.weight_hh_l0.shape[1]
        A = self.attention.decoder_att.weight.shape[0]
        K, S = int(beamk), int(max_gen_length)
        temps = temperature if isinstance(temperature, list) else [temperature]
        tarr = (C.c_float * len(temps))(*[float(t) for t in temps])
        ids = (C.c_int32 * 4)(int(hp.vocab_stoi["<START>"]), int(hp.vocab_stoi["<PAD>"]), int(hp.vocab_stoi["<END>"]), int(hp.vocab_stoi["<UNK>"]))
        dims = Dk.decoder_dims(B, K, 2, Lc, D, A, m, n, V, 0, hp.deep_output, self.pad_idx, int(getattr(self, "sat_precision", "fp32") == "bf16"),
                               layers=int(hp.decoder_layers))
        w, _keep = self._params_struct()
        ws_bytes = lib.sat_beam_search_workspace_bytes(C.byref(dims), K)
        i32 = dict(dtype=torch.int32, device=dev); f32 = dict(dtype=torch.float32, device=dev)
        smp = None
        if sample_method != "beam" or decoder_noise:
            if seed is None:
                seed = int(torch.randint(0, 2 ** 62, (1,)))
            for t in (gumbel, normals):
                if t is not None:
                    L.require_gpu(t)
                    assert t.dtype == torch.float32 and t.is_contiguous()
            smp = L.BeamSampling(method={"beam": 0, "multinomial": 1, "topk": 2}[sample_method], sample_topk=int(sample_topk), seed=int(seed),
                                 gumbel=(gumbel.data_ptr() if gumbel is not None else None), decoder_noise=float(decoder_noise or 0.0),
                                 normals=(normals.data_ptr() if normals is not None else None))

        def buffers():
            return dict(ws=torch.empty(ws_bytes, dtype=torch.uint8, device=dev), tok_in=torch.empty(S + 2, B, K, **i32),
                        prev_row=torch.empty(S + 2, B, K, **i32), alpha_hist=torch.empty(S + 1, B, K, Lc, **f32), fin_count=torch.empty(B, **i32),
                        fin_step=torch.empty(B, K, **i32), fin_row=torch.empty(B, K, **i32), fin_score=torch.empty(B, K, **f32),
                        fin_mean=torch.empty(B, K, **f32))

        def enqueue(ann, o):
            L.check(lib.sat_beam_search_sampled(C.byref(dims), C.byref(w), L.ptr(ann), K, S, tarr, len(temps), ids, C.byref(smp) if smp is not None else None,
                                                L.ptr(o["tok_in"]), L.ptr(o["prev_row"]), L.ptr(o["alpha_hist"]), L.ptr(o["fin_count"]), L.ptr(o["fin_step"]),
                                                L.ptr(o["fin_row"]), L.ptr(o["fin_score"]), L.ptr(o["fin_mean"]), L.ptr(o["ws"]), ws_bytes, L.stream_ptr()),
                    "sat_beam_search_sampled")

        if graph and smp is None:
            cache = self.__dict__.setdefault("_beam_graphs", {})
            key = (B, Lc, D, K, S, tuple(float(t) for t in temps), int(dims.precision), str(dev), tuple(t.data_ptr() for t in _keep.values() if torch.is_tensor(t)))
            ent = cache.get(key)
            if ent is None:
                o = buffers()
                ann_s = ann_bld.clone()
                enqueue(ann_s, o)                       # eager once: validates the arguments and sets the kernel attributes before the capture
                torch.cuda.synchronize(dev)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    enqueue(ann_s, o)
                while len(cache) >= 4:                  # a few shapes at most: every entry pins its buffers
                    cache.pop(next(iter(cache)))
                ent = cache[key] = (g, ann_s, o, _keep)
            g, ann_s, o = ent[0], ent[1], ent[2]
            ann_s.copy_(ann_bld)
            g.replay()
        else:
            o = buffers()
            enqueue(ann_bld, o)
        tok_in, prev_row, alpha_hist = o["tok_in"], o["prev_row"], o["alpha_hist"]
        fin_count, fin_step, fin_row, fin_score, fin_mean = o["fin_count"], o["fin_step"], o["fin_row"], o["fin_score"], o["fin_mean"]
        tok_in, prev_row = tok_in.cpu().numpy(), prev_row.cpu().numpy()
        alpha_np = alpha_hist.cpu().numpy()
        fin_count, fin_step, fin_row = fin_count.cpu().numpy(), fin_step.cpu().numpy(), fin_row.cpu().numpy()
        fin_score, fin_mean = fin_score.cpu(), fin_mean.cpu()
        # back-trace of every finished hypothesis at once: walk the parent rows from the step it ended at down to step 0
        bidx = np.repeat(np.arange(B), K)[(np.arange(K)[None, :] < fin_count[:, None]).reshape(-1)]
        fidx = np.tile(np.arange(K), B)[(np.arange(K)[None, :] < fin_count[:, None]).reshape(-1)]
        hstep, cur = fin_step[bidx, fidx].astype(np.int64), fin_row[bidx, fidx].astype(np.int64)
        nh = len(bidx)
        rows = np.zeros((S + 1, nh), np.int64)
        for s_ in range(S, -1, -1):
            act = hstep >= s_
            rows[s_, act] = cur[act]
            if s_ > 0:
                cur[act] = prev_row[s_, bidx[act], cur[act]]
        sidx = np.arange(S + 1)[:, None]
        toks_all = tok_in[sidx, bidx[None, :], rows]                       # (S+1, nh): token fed at step s to the hypothesis' ancestor
        als_all = torch.from_numpy(alpha_np[sidx, bidx[None, :], rows])    # (S+1, nh, L)
        if rescore_method == "LN":
            resc = fin_score[bidx, fidx] / torch.from_numpy(hstep).float()
        elif rescore_method == "WR":
            resc = fin_score[bidx, fidx] + rescore_reward * torch.from_numpy(hstep).float()
        elif rescore_method == "BAR":
            resc = fin_score[bidx, fidx] + rescore_reward * (-fin_mean[bidx, fidx])
        else:
            resc = fin_score[bidx, fidx]
        ppl_all = torch.exp(-fin_score[bidx, fidx] / torch.from_numpy(hstep).float())
        resc, ppl_all = resc.tolist(), ppl_all.tolist()
        captions, cap_scores, cap_alphas, cap_ppl = [], [], [], []
        hpos = 0
        for b in range(B):
            fin_caps, fin_alphas, fin_scores, fin_ppl = [], [], [], []
            for f in range(int(fin_count[b])):
                hh, step = hpos + f, int(hstep[hpos + f])
                # top_preds[:, i][1:-1] / alphas[:, i][1:-1] (model.py:412-413): without START / the zero map and without the last step
                fin_caps.append(toks_all[1:step + 1, hh].tolist())
                fin_alphas.append(als_all[:step, hh].reshape(-1, Hh, Ww).clone())
                fin_scores.append(resc[hh]); fin_ppl.append(ppl_all[hh])
            hpos += int(fin_count[b])
            if return_all:
                order = [i for _, i in sorted([[fin_scores[i], i] for i in range(len(fin_scores))], reverse=True)]
                captions.append([fin_caps[i] for i in order]); cap_alphas.append([fin_alphas[i] for i in order])
                cap_scores.append([fin_scores[i] for i in order]); cap_ppl.append([fin_ppl[i] for i in order])
            else:
                best = fin_scores.index(max(fin_scores))
                captions.append(fin_caps[best]); cap_alphas.append(fin_alphas[best])
                cap_scores.append(fin_scores[best]); cap_ppl.append(fin_ppl[best])
        return captions, cap_scores, cap_alphas, cap_ppl

    def _dropout_args(self, seed=None):
        """(p, p_embedding, seed) of this call: nn.Dropout is the identity in eval mode; the seed comes from a generator of
        its own so that the CPU generator stream of scheduled sampling (F7) stays exactly the reference's."""
        p, pe = float(self.hp.dropout), float(self.hp.embedding_dropout)
        if not self.training or (p == 0.0 and pe == 0.0):
            return (0.0, 0.0, 0)
        if seed is None:
            gen = self.__dict__.get("_dropout_gen")
            if gen is None:
                gen = self.__dict__["_dropout_gen"] = torch.Generator().manual_seed(torch.initial_seed() & 0x7FFFFFFFFFFFFFFF)
            seed = int(torch.randint(0, 2 ** 62, (1,), generator=gen))
        return (p, pe, int(seed))

    def train_decode(self, ann_bld, caps, lengths, epsilon=0, draw=None, with_loss=True, dropout_seed=None, teacher=None):
        """Decoder half of train_batch + the loss terms (model.py:487-557, 592-597).

        ann_bld (B, L, D) on the GPU; caps (B, R, T) int64; lengths (B, R) int64 (host or device).  ``teacher``: the scheduled-sampling
        decisions of this batch when the caller has drawn them already (``PackPlan.teacher_flags``; graph.py keys its graphs by them)."""
        B, R, T = caps.shape
        plan = Dk.PackPlan.cached(lengths.reshape(-1).cpu(), T, ann_bld.device)
        if teacher is None:
            teacher = plan.teacher_flags(float(epsilon), draw)
        caps2 = caps.reshape(B * R, T)
        caps_i32 = caps2.to(device=ann_bld.device, dtype=torch.int32).contiguous()
        logits_packed, alphas = Dk.DecoderTrainFn.apply(ann_bld, caps_i32, plan, teacher, bool(self.hp.deep_output), self.pad_idx, R,
                                                        int(getattr(self, "sat_precision", "fp32") == "bf16"), getattr(self.hp, "embed_norm", None) or 0.0,
                                                        self._dropout_args(dropout_seed), *self.param_list())
        targets_packed = plan.pack(caps2[:, 1:].to(ann_bld.device).unsqueeze(-1)).squeeze(-1)
        if not with_loss:
            return dict(logits_packed=logits_packed, targets_packed=targets_packed, alphas=alphas, plan=plan)
        ce = self.criterion(logits_packed, targets_packed)
        ds = Dk.DoublyStochasticFn.apply(alphas, float(self.hp.att_gamma))
        return dict(logits_packed=logits_packed, targets_packed=targets_packed, alphas=alphas, ce=ce, ds=ds,
                    acc=self.criterion.last_accuracy, plan=plan)


class SAT(SATDecoder, _Base):
    """``class SAT(pl.LightningModule)`` of the reference (model.py:134-199) for the train-step hot path.

    Same constructor kwargs (train.py:264 passes ``**vars(args)``), same sub-module attribute names and
    state-dict keys; ``train_batch`` / ``training_step`` / ``configure_optimizers`` keep their signatures.
    Construction order follows the reference (criterion, encoder, embedding, init_lstm, lstm, attention, beta,
    output) so that a seed produces the same parameter stream.  ``caption`` / ``forward`` run the reference's per-image
    beam search (beam / multinomial / topk sampling, decoder noise) on the HIP step kernels; ``score_captions`` / ``val_batch`` /
    ``validation_step`` (model.py:646-718) use metrics.py, the nltk algorithms restated."""

    def __init__(self, **kwargs):
        nn.Module.__init__(self)
        hp = _HParams(kwargs)
        for k, v in dict(encoder_size=None, embed_norm=None, pretrained_embedding=None, pretrained=False, decoder_tf=None,
                         decoder_tf_min=0.5, epochs=10, encoder_finetune_after=-1, att_gamma=1.0, label_smoothing=0.0,
                         dropout=0.0, embedding_dropout=0.0, weight_tying=False, deep_output=False, decoder_layers=1,
                         lr_warmup_steps=0).items():
            hp.setdefault(k, v)
        self.scheduler = None
        self.opt_init_lr = None
        self.__dict__["_sat_global_step"] = 0           # optimizer steps taken (Lightning's trainer.global_step when no trainer is attached)
        self.__dict__["_sat_micro_batches"] = 0         # training_step calls inside the current accumulation window
        # reference order: criterion -> encoder -> decoder parts (model.py:148-195); get_encoder writes
        # hp.encoder_dim back when no projection is needed (model.py:56)
        self._build_decoder(hp, encoder_factory=get_encoder)
        # train.py:31-32 `--precision 16` asks Lightning for torch AMP (fp16 autocast + GradScaler).  The MI355X-native reduced-precision mode is
        # bf16 storage / bf16 MFMA with fp32 accumulation, master weights and losses (fp32 exponent range: no loss scaling needed); there is no
        # fp16 path.  `hip_precision` overrides.
        amp = str(hp.get("precision", 32)) in ("16", "bf16", "16-mixed", "bf16-mixed")
        self.set_precision(hp.get("hip_precision", "bf16" if amp else "fp32"))
        if hp.pretrained_embedding is not None:
            import numpy as np
            self.embedding.weight = nn.Parameter(torch.tensor(np.load(hp.pretrained_embedding), dtype=torch.float32))
        self.special_idxs = [self.stoi("<PAD>"), self.stoi("<START>"), self.stoi("<END>")]

    def set_precision(self, mode):
        """"fp32": exact fp32 MFMA everywhere (parity mode).  "bf16": bf16 matrix cores with fp32 accumulation --
        bf16 activations/filter copies in the encoder, fp32 decoder state rounded on the way into LDS; master
        weights, gradients of parameters, statistics and losses stay fp32."""
        if mode not in ("fp32", "bf16"):
            raise ValueError("hip_precision must be 'fp32' or 'bf16'")
        self.sat_precision = mode
        self.encoder.precision = mode

    # Lightning gives `hparams`; without Lightning keep the same attribute
    @property
    def hparams(self):
        return self.hp

    def stoi(self, s):
        return int(self.hp.vocab_stoi.get(s, self.hp.vocab_stoi["<UNK>"]))

    def itos(self, i):
        return str(self.hp.vocab_itos.get(int(i), "<UNK>"))

    def decode_seq(self, seq, remove_special=False):
        return [str(self.itos(t)) for t in seq if not (remove_special and t in self.special_idxs)]

    # ------------------------------------------------------------------ train_batch (model.py:474-557)
    def encode(self, img):
        ann = self.encoder(img)                                  # (B, D, h, w), channels-last memory
        B, D, h, w = ann.shape
        return ann.permute(0, 2, 3, 1).reshape(B, h * w, D), (h, w)

    @torch.no_grad()
    def caption(self, img_tensor, beamk=3, max_gen_length=32, temperature=1.0, sample_method="beam", sample_topk=3,
                decoder_noise=None, rescore_method=None, rescore_reward=0.5, return_all=False):
        """model.py:214-235: eval mode, then forward."""
        self.eval()
        return self.forward(img_tensor, beamk, max_gen_length, temperature, sample_method, sample_topk, decoder_noise,
                            rescore_method, rescore_reward, return_all)

    def forward(self, img, beamk=3, max_gen_length=32, temperature=1.0, sample_method="beam", sample_topk=3, decoder_noise=None,
                rescore_method=None, rescore_reward=0.5, return_all=False):
        """Inference only (model.py:237-472): encode the batch once, then beam-search every image."""
        assert sample_method in ["beam", "multinomial", "topk"]
        with torch.no_grad():
            ann_bld, hw = self.encode(img)
            if max_gen_length >= 1:        # every image at once; sampled continuations and decoder noise draw from the device generator
                return self.beam_decode_batched(ann_bld.contiguous(), hw, beamk, max_gen_length, temperature, rescore_method, rescore_reward, return_all,
                                                sample_method=sample_method, sample_topk=sample_topk, decoder_noise=decoder_noise,
                                                graph=bool(self.__dict__.get("beam_graph", False)))       # model.beam_graph = True: hipGraph replay
            return self.beam_decode(ann_bld.contiguous(), hw, beamk, max_gen_length, temperature, sample_method, sample_topk,
                                    decoder_noise, rescore_method, rescore_reward, return_all)

    def train_batch(self, batch, epsilon=0, draw=None, teacher=None):
        img, encoded_captions, lengths = batch
        ann_bld, _ = self.encode(img)
        res = self.train_decode(ann_bld, encoded_captions, lengths, float(epsilon), draw, with_loss=False, teacher=teacher)
        plan = res["plan"]
        lp = PackedSequence(res["logits_packed"], plan.batch_sizes, plan.sorted_indices_dev, plan.unsorted_indices_dev)
        tp = PackedSequence(res["targets_packed"], plan.batch_sizes, plan.sorted_indices_dev, plan.unsorted_indices_dev)
        return lp, tp, res["alphas"]

    def teacher_forcing_epsilon(self, current_epoch=0):
        """model.py:565-582 (host scalars)."""
        hp = self.hp
        if hp.decoder_tf is None:
            return 0.0
        if hp.decoder_tf == "always":
            return 1.0
        if hp.decoder_tf == "linear":
            return 1 - (1 - hp.decoder_tf_min) * current_epoch / hp.epochs
        if hp.decoder_tf == "inv_sigmoid":
            l = -math.log(hp.decoder_tf_min / (1 - hp.decoder_tf_min)); g = 5.0
            b = (1 / ((l / g) + 1)) * hp.epochs
            return 1 / (1 + math.exp((g / b) * (current_epoch - b)))
        if hp.decoder_tf == "exp":
            return math.exp(math.log(hp.decoder_tf_min) / hp.epochs) ** current_epoch
        raise ValueError("decoder_tf=%r" % (hp.decoder_tf,))

    def sat_global_step(self):
        """Lightning's ``trainer.global_step`` (optimizer steps taken so far); without a trainer, the module's own count:
        one more every ``hparams.accumulate`` calls of ``training_step`` (train.py:70-71, 267)."""
        tr = getattr(self, "_trainer", None) if pl is not None else None
        if tr is not None:
            return int(tr.global_step)
        return int(self.__dict__["_sat_global_step"])

    def _train_optimizer(self):
        """the optimizer ``configure_optimizers`` built: Lightning's ``self.optimizers()`` under a trainer"""
        if pl is not None and getattr(self, "_trainer", None) is not None:
            return self.optimizers()
        return self.__dict__.get("_sat_optimizer")

    def step_learning_rate(self, gstep):
        """model.py:614-626, run at the end of every ``training_step`` with the optimizer steps taken so far: linear warm-up of
        every group's LR from ``opt_init_lr`` while ``gstep < lr_warmup_steps``; afterwards the per-batch schedulers
        (CosineAnnealingWarmRestarts, OneCycleLR) advance once per call."""
        hp = self.hp
        opt = self._train_optimizer()
        if opt is None or self.opt_init_lr is None:          # the caller drives its own optimizer (configure_optimizers never ran)
            return
        if gstep < hp.lr_warmup_steps:
            lr_scale = min(1, float(gstep + 1) / hp.lr_warmup_steps)
            for pg, init_lr in zip(opt.param_groups, self.opt_init_lr):
                pg["lr"] = lr_scale * init_lr
        elif gstep > 0:
            if type(self.scheduler) in [CosineAnnealingWarmRestarts, OneCycleLR]:
                self.scheduler.step()

    def _step_begin(self):
        """host half of ``training_step`` in front of the device work (model.py:559-586): (epsilon, optimizer steps so far); unfreezes the
        encoder at ``encoder_finetune_after``"""
        hp = self.hp
        epoch = int(getattr(self, "current_epoch", 0) or 0)        # Lightning's property, or a plain attribute set by the caller's loop
        epsilon = self.teacher_forcing_epsilon(epoch)
        gstep = self.sat_global_step()
        if gstep == hp.encoder_finetune_after and hp.encoder_finetune_after >= 0:
            for p in self.encoder.parameters():
                p.requires_grad = True
        return epsilon, gstep

    def _step_losses(self, batch, epsilon, teacher=None):
        """device half: forward + the two loss terms (model.py:588-597)"""
        lp, tp, alphas = self.train_batch(batch, epsilon, teacher=teacher)
        loss = self.criterion(lp.data, tp.data)                                   # model.py:592
        loss = loss + Dk.DoublyStochasticFn.apply(alphas, float(self.hp.att_gamma))      # model.py:594
        return loss

    def _step_end(self, gstep):
        """host half behind the device work (model.py:614-626): learning-rate recipe, optimizer-step count"""
        hp = self.hp
        self.step_learning_rate(gstep)                                            # model.py:614-626
        micro = self.__dict__["_sat_micro_batches"] + 1
        if micro >= max(1, int(getattr(hp, "accumulate", 1) or 1)):                # the trainer steps the optimizer after this batch
            self.__dict__["_sat_global_step"] += 1
            micro = 0
        self.__dict__["_sat_micro_batches"] = micro

    def training_step(self, batch, batch_idx=0):
        """model.py:559-628: returns the metrics dict whose "loss" the trainer back-propagates.  (Split in three so that
        ``graph.GraphedTrainStep`` can replay the device half from a hipGraph and still run the two host halves every step.)"""
        epsilon, gstep = self._step_begin()
        loss = self._step_losses(batch, epsilon)
        self._step_end(gstep)
        return {"loss": loss, "accuracy": self.criterion.last_accuracy, "epsilon_tf": float(epsilon)}

    # ------------------------------------------------------------------ validation (model.py:630-718)
    def _log_scalar(self, key, val, step):
        """tensorboard scalar when a Lightning-style logger is attached (model.py:611, 635, 708); a no-op otherwise"""
        logger = getattr(self, "logger", None)
        exp = getattr(logger, "experiment", None)
        if exp is not None and hasattr(exp, "add_scalar"):
            exp.add_scalar(key, val, global_step=step)

    def training_epoch_end(self, outputs):
        """model.py:630-644: epoch means of the step metrics, learning rate, per-epoch scheduler step."""
        epoch = int(getattr(self, "current_epoch", 0))
        means = {}
        for k in outputs[0].keys():
            vals = [float(x[k]) for x in outputs]
            means[k] = sum(vals) / len(vals) if vals else 0
            self._log_scalar("{}/train_epoch".format(k), means[k], epoch + 1)
        if isinstance(getattr(self, "scheduler", None), (MultiStepLR, ExponentialLR)):
            self.scheduler.step()
        return means

    @torch.no_grad()
    def score_captions(self, captions, encoded_captions, lengths, perplexities=None):
        """model.py:646-682: corpus BLEU-1..4 and GLEU of the generated captions against the R references of each image, and the
        best cosine similarity between mean embeddings.  BLEU / GLEU: metrics.py (nltk's algorithms restated; nltk is a host-side
        dependency of the reference and runs on token-id lists); the embedding part runs on the device."""
        from . import metrics
        lens = lengths.tolist() if torch.is_tensor(lengths) else lengths
        references = [[c[1:l] for c, l in zip(refs, lens[i])] for i, refs in enumerate(encoded_captions.tolist())]
        out = {"bleu1": metrics.corpus_bleu(references, captions, weights=(1, 0, 0, 0)),
               "bleu2": metrics.corpus_bleu(references, captions, weights=(0.5, 0.5, 0, 0)),
               "bleu3": metrics.corpus_bleu(references, captions, weights=(0.33, 0.33, 0.33, 0)),
               "bleu4": metrics.corpus_bleu(references, captions, weights=(0.25, 0.25, 0.25, 0.25))}
        dev = self.embedding.weight.device
        enc = encoded_captions.to(dev)
        E = self.embedding.weight
        cossims = torch.zeros(enc.shape[0], dtype=torch.float, device=dev)
        for i in range(enc.shape[0]):
            cv = E[torch.as_tensor(captions[i], dtype=torch.long, device=dev)].mean(0).unsqueeze(0)
            rvs = torch.zeros(enc.shape[1], dtype=torch.float, device=dev)
            for j, l in enumerate(lens[i]):
                rv = E[enc[i][j][1:l]].mean(0).unsqueeze(0)
                rvs[j] = F.cosine_similarity(rv, cv)
            cossims[i] = rvs.max()
        out["cosine_similarity"] = cossims.mean().item()
        out["gleu"] = metrics.corpus_gleu(references, captions)
        if type(perplexities) == list:
            out["perplexity"] = sum(perplexities) / len(perplexities)
        return out

    def val_batch(self, batch, beamk=3, max_gen_length=32, temperature=0.5, sample_method="beam", sample_topk=3, decoder_noise=None,
                  rescore_method=None, rescore_reward=0.5):
        """model.py:684-691"""
        img, encoded_captions, lengths = batch
        captions, scores, alphas, perplexities = self.caption(img, beamk, max_gen_length, temperature, sample_method, sample_topk, decoder_noise,
                                                              rescore_method, rescore_reward, return_all=False)
        return self.score_captions(captions, encoded_captions, lengths, perplexities)

    def validation_step(self, batch, batch_idx=0):
        """model.py:693-697"""
        return self.val_batch(batch, beamk=self.hp.val_beamk, max_gen_length=self.hp.val_max_len, temperature=1.0, rescore_method="LN")

    def validation_epoch_end(self, outputs):
        """model.py:699-718: epoch means; the monitored ones go to ``self.log`` when a trainer is attached; plateau scheduler step."""
        epoch = int(getattr(self, "current_epoch", 0))
        means, plateau_val = {}, None
        for k in outputs[0].keys():
            vals = [x[k] for x in outputs]
            try:
                val = sum(vals) / len(vals)
            except Exception:
                val = 0
            means[k] = val
            if epoch != 0:
                self._log_scalar("{}/val_epoch".format(k), val, epoch + 1)
            if hasattr(self, "log") and getattr(self, "_trainer", None) is not None and k in (getattr(self.hp, "save_monitor", None), getattr(self.hp, "early_stop_monitor", None)):
                self.log(k, val)
            if k == getattr(self.hp, "plateau_monitor", None):
                plateau_val = val
        if isinstance(getattr(self, "scheduler", None), ReduceLROnPlateau) and plateau_val is not None:
            self.scheduler.step(plateau_val)
        return means

    # ------------------------------------------------------------------ configure_optimizers (model.py:720-817)
    def configure_optimizers(self):
        hp = self.hp

        def groups(modules, weight_decay, lr):
            decay, no_decay = [], []
            for mod in modules:
                for _, p in mod.named_parameters():
                    if p.requires_grad:
                        (no_decay if p.dim() == 1 else decay).append(p)
            return [{"params": no_decay, "lr": lr, "weight_decay": 0.0}, {"params": decay, "lr": lr, "weight_decay": weight_decay}]

        params = groups([self.init_lstm, self.lstm, self.attention, self.beta, self.output], hp.weight_decay, hp.decoder_lr)
        if hp.embedding_lr > 0 and not hp.weight_tying:
            params += [{"params": self.embedding.parameters(), "lr": hp.embedding_lr, "weight_decay": 0.0}]
        if hp.encoder_finetune_after > 0 and hp.encoder_lr > 0:          # F10: mirrored as written
            params += groups([self.encoder], hp.weight_decay, hp.encoder_lr)
        if hp.opt not in ("sgd", "adam", "adamw"):
            raise ValueError("opt=%r" % (hp.opt,))
        if getattr(hp, "fused_optimizer", True):
            # one launch over every parameter tensor (csrc/optimizer.hip), gradient clipping of train.py:93-96 folded in
            from .optim import FusedOptimizer
            opt = FusedOptimizer(params, kind=hp.opt, lr=hp.decoder_lr, betas=(hp.adam_b1, hp.adam_b2), momentum=hp.momentum, nesterov=hp.nesterov,
                                 grad_clip=getattr(hp, "grad_clip", None), clip_value=getattr(hp, "clip_value", 0.0))
        elif hp.opt == "sgd":
            opt = torch.optim.SGD(params, lr=hp.decoder_lr, momentum=hp.momentum, nesterov=hp.nesterov)
        elif hp.opt == "adam":
            opt = torch.optim.Adam(params, lr=hp.decoder_lr, betas=(hp.adam_b1, hp.adam_b2))
        else:
            opt = torch.optim.AdamW(params, lr=hp.decoder_lr, betas=(hp.adam_b1, hp.adam_b2))
        self.opt_init_lr = [pg["lr"] for pg in opt.param_groups]
        self.__dict__["_sat_optimizer"] = opt
        sched = getattr(hp, "scheduler", None)
        if sched == "step":
            self.scheduler = MultiStepLR(opt, milestones=hp.milestones, gamma=hp.lr_gamma)
        elif sched == "plateau":
            self.scheduler = ReduceLROnPlateau(opt, mode="max", factor=hp.lr_gamma, patience=hp.plateau_patience, min_lr=hp.min_lr)
        elif sched == "exp":
            self.scheduler = ExponentialLR(opt, gamma=hp.lr_gamma)
        elif sched == "cosine":
            adj = hp.epochs * hp.train_loader_len - hp.lr_warmup_steps
            t0, tm = hp.cosine_iterations, hp.cosine_multi
            if tm != 1:
                restarts = math.floor(math.log(1 - (adj * (1 - tm) / t0)) / math.log(tm))
                t0 = adj + hp.accumulate if restarts == 0 else math.ceil((adj + hp.accumulate) / ((1 - tm ** restarts) / (1 - tm)))
            else:
                restarts = math.floor(adj / t0)
                t0 = adj + hp.accumulate if restarts == 0 else math.ceil((adj + hp.accumulate) / restarts)
            self.scheduler = CosineAnnealingWarmRestarts(opt, T_0=int(t0), T_mult=int(tm), eta_min=hp.min_lr)
        elif sched == "one_cycle":
            hp.lr_warmup_steps = 0
            self.scheduler = OneCycleLR(opt, self.opt_init_lr, epochs=hp.epochs, steps_per_epoch=hp.train_loader_len,
                                        pct_start=hp.one_cycle_pct, cycle_momentum=False, div_factor=hp.one_cycle_div,
                                        final_div_factor=hp.one_cycle_fdiv)
        return opt
