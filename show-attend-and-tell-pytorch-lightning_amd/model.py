"""Drop-in mirror of the reference's ``model.py`` surface for the train-step hot path:
``get_encoder``, ``InitLSTM``, ``SoftAttention``, ``DeepOutput`` and ``SAT`` with the same
constructor kwargs, attribute names and state-dict keys (reference model.py:16-199),
so ``train.py``-style callers and checkpoints keep working.  The sub-modules only *hold*
parameters (created in the reference's order, so a given seed yields the same initial
weights); all arithmetic goes through ``libsat_hip.so``.
"""
import math
from types import SimpleNamespace

import torch
from torch import nn

from . import _lib as L
from . import decoder as Dk

try:                                     # Lightning is optional plumbing (absent in the build image)
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:                        # pragma: no cover - exercised where Lightning is missing
    pl = None
    _Base = nn.Module


class _HParams(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class InitLSTM(nn.Module):
    """Parameter holder for model.py:66-81 (factorize: D->m, init: m->2*n*layers)."""

    def __init__(self, args, bias=True):
        super().__init__()
        self.decoder_dim, self.decoder_layers = args.decoder_dim, args.decoder_layers
        self.factorize = nn.Linear(args.encoder_dim, args.embed_dim, bias=bias)
        self.init = nn.Linear(args.embed_dim, 2 * args.decoder_dim * args.decoder_layers, bias=bias)
        self.dropout = nn.Dropout(p=args.dropout)


class SoftAttention(nn.Module):
    """Parameter holder for model.py:84-109 (three bias-free Linears)."""

    def __init__(self, args):
        super().__init__()
        self.encoder_att = nn.Linear(args.encoder_dim, args.attention_dim, bias=False)
        self.decoder_att = nn.Linear(args.decoder_dim, args.attention_dim, bias=False)
        self.f_att = nn.Linear(args.attention_dim, 1, bias=False)


class DeepOutput(nn.Module):
    """Parameter holder for model.py:112-131."""

    def __init__(self, args):
        super().__init__()
        self.deep = args.deep_output
        self.dropout = nn.Dropout(p=args.dropout)
        self.hidden = nn.Linear(args.decoder_dim, args.embed_dim, bias=False)
        if self.deep:
            self.context = nn.Linear(args.encoder_dim, args.embed_dim, bias=False)
        self.output = nn.Linear(args.embed_dim, args.vocab_size, bias=(not args.weight_tying))


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
        if isinstance(hp, dict):
            hp = _HParams(hp)
        self.hp = hp
        if hp.decoder_layers != 1:
            raise NotImplementedError("HIP decoder: decoder_layers=%d (only 1 layer is built this round)" % hp.decoder_layers)
        if float(hp.dropout) != 0.0 or float(hp.embedding_dropout) != 0.0:
            raise NotImplementedError("HIP decoder: dropout > 0 is not built this round (parity runs use 0, train.py:140-143)")
        if getattr(hp, "embed_norm", None) is not None:
            raise NotImplementedError("HIP decoder: embedding max_norm is not built this round")
        assert 0 <= hp.label_smoothing < (hp.vocab_size - 1) / hp.vocab_size
        self.criterion = LabelSmoothing(hp.label_smoothing)
        self.pad_idx = int(hp.vocab_stoi["<PAD>"])
        self.embedding = nn.Embedding(hp.vocab_size, hp.embed_dim, max_norm=None, padding_idx=self.pad_idx)
        self.embedding_dropout = nn.Dropout(p=hp.embedding_dropout)
        self.init_lstm = InitLSTM(hp, bias=True)
        self.lstm = nn.LSTM(input_size=hp.embed_dim + hp.encoder_dim, hidden_size=hp.decoder_dim, num_layers=hp.decoder_layers, bias=True)
        self.attention = SoftAttention(hp)
        self.beta = nn.Sequential(nn.Linear(hp.decoder_dim, hp.encoder_dim, bias=True), nn.Sigmoid())
        fan_in = self.beta[0].weight.shape[1]
        self.beta[0].bias.data.fill_(1 / fan_in)
        self.output = DeepOutput(hp)
        if hp.weight_tying and hp.deep_output:
            self.output.output.weight = self.embedding.weight

    # -- parameters in the order of sat_decoder_params (include/sat_hip.h)
    def param_list(self):
        o = self.output
        return [self.embedding.weight, self.init_lstm.factorize.weight, self.init_lstm.factorize.bias, self.init_lstm.init.weight,
                self.init_lstm.init.bias, self.lstm.weight_ih_l0, self.lstm.weight_hh_l0, self.lstm.bias_ih_l0, self.lstm.bias_hh_l0,
                self.attention.encoder_att.weight, self.attention.decoder_att.weight, self.attention.f_att.weight,
                self.beta[0].weight, self.beta[0].bias, o.hidden.weight, (o.context.weight if o.deep else None),
                o.output.weight, o.output.bias]

    def load_decoder_state(self, sd):
        own = self.state_dict()
        for k, v in sd.items():
            if k in own:
                own[k].copy_(torch.as_tensor(v))

    def train_decode(self, ann_bld, caps, lengths, epsilon=0, draw=None):
        """Decoder half of train_batch + the loss terms (model.py:487-557, 592-597).

        ann_bld (B, L, D) on the GPU; caps (B, R, T) int64; lengths (B, R) int64 (host or device)."""
        B, R, T = caps.shape
        plan = Dk.PackPlan(lengths.reshape(-1).cpu(), T, ann_bld.device)
        teacher = plan.teacher_flags(float(epsilon), draw)
        caps2 = caps.reshape(B * R, T)
        caps_i32 = caps2.to(device=ann_bld.device, dtype=torch.int32).contiguous()
        logits_packed, alphas = Dk.DecoderTrainFn.apply(ann_bld, caps_i32, plan, teacher, bool(self.hp.deep_output), self.pad_idx, R,
                                                        *self.param_list())
        targets_packed = plan.pack(caps2[:, 1:].to(ann_bld.device).unsqueeze(-1)).squeeze(-1)
        ce = self.criterion(logits_packed, targets_packed)
        ds = Dk.DoublyStochasticFn.apply(alphas, float(self.hp.att_gamma))
        return dict(logits_packed=logits_packed, targets_packed=targets_packed, alphas=alphas, ce=ce, ds=ds,
                    acc=self.criterion.last_accuracy, plan=plan)
