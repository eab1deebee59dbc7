"""Build-container-only helper: import the reference's ``model.py`` unmodified.

Three of its imports are absent from this image (``nltk``,
``pytorch_lightning``, ``torchvision``).  None of them contributes
arithmetic to the decoder path, so placeholders are registered in
``sys.modules`` before the import:

* ``nltk.translate.*``      - BLEU/GLEU, never called on the train-step path;
* ``pytorch_lightning``     - ``LightningModule`` is an ``nn.Module`` that keeps
                              ``save_hyperparameters()`` / ``hparams`` / ``device``;
* ``torchvision``           - ``models.<resnet>`` come from ``oracle.sat_oracle``
                              (the repo's own ResNet definition) so that the
                              reference's *own* ``get_encoder`` slicing, 1x1 conv
                              and Normalize logic runs; encoder arithmetic is
                              therefore "unpinned at the reference level".

This module reads /root/reference and is never shipped to, or run on, the GPU
box; only the numeric fixtures it helps produce are committed.
"""
import inspect
import os
import sys
import types

import torch
from torch import nn

REFERENCE_DIR = os.environ.get("SAT_REFERENCE_DIR", "/root/reference")
_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _REPO not in sys.path:
    sys.path.insert(0, _REPO)


class _AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _install_placeholders():
    from oracle import sat_oracle as O

    nltk = types.ModuleType("nltk")
    tr = types.ModuleType("nltk.translate")
    bl = types.ModuleType("nltk.translate.bleu_score")
    gl = types.ModuleType("nltk.translate.gleu_score")
    bl.corpus_bleu = lambda *a, **k: 0.0
    gl.corpus_gleu = lambda *a, **k: 0.0
    nltk.translate, tr.bleu_score, tr.gleu_score = tr, bl, gl
    sys.modules.update({"nltk": nltk, "nltk.translate": tr,
                        "nltk.translate.bleu_score": bl, "nltk.translate.gleu_score": gl})

    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        def save_hyperparameters(self):
            frame = inspect.currentframe().f_back
            self.hparams = _AttrDict(frame.f_locals.get("kwargs", {}))

        @property
        def device(self):
            return next(self.parameters()).device

    cb = types.ModuleType("pytorch_lightning.callbacks")
    cb.ModelCheckpoint = type("ModelCheckpoint", (), {"__init__": lambda self, *a, **k: None})
    pl.LightningModule, pl.Trainer, pl.callbacks = LightningModule, type("Trainer", (), {}), cb
    sys.modules.update({"pytorch_lightning": pl, "pytorch_lightning.callbacks": cb})

    tv = types.ModuleType("torchvision")
    models = types.ModuleType("torchvision.models")
    for arch in O.RESNET_TABLE:
        setattr(models, arch, O.resnet_factory(arch))
    tf = types.ModuleType("torchvision.transforms")
    tf.Normalize = O.NormalizeInplace
    tv.models, tv.transforms = models, tf
    sys.modules.update({"torchvision": tv, "torchvision.models": models, "torchvision.transforms": tf})


def load_reference():
    """Returns the reference's ``model`` module (imports ``util`` too)."""
    if not os.path.isdir(REFERENCE_DIR):
        raise RuntimeError("reference tree not present: %s" % REFERENCE_DIR)
    _install_placeholders()
    sys.dont_write_bytecode = True
    if REFERENCE_DIR not in sys.path:
        sys.path.insert(0, REFERENCE_DIR)
    import model  # noqa: the reference's model.py
    return model


def make_sat(ref, hp_namespace, seed=42):
    torch.manual_seed(seed)
    return ref.SAT(**vars(hp_namespace))
