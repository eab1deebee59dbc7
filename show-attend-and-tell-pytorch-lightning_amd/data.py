"""Input pipeline with the per-picture transforms on the GPU.

Host-side mirror of the reference's data path:

* ``CocoCaptionDataset``   util.py:16-45 - same JSON (``vocab_stoi``, ``<split>.img_paths / encoded_captions / lengths``,
  preprocess.ipynb cell 17); ``__getitem__`` returns the *decoded bytes* (H, W, 3) uint8 instead of a transformed tensor.
* ``BucketSampler``        util.py:48-87 - same grouping and shuffle; additionally rank/world_size aware.
* ``BatchTransform``       train.py:208-233 - RandomResizedCrop | Resize+CenterCrop, RandomHorizontalFlip, ToTensor,
  AddGaussianNoise (util.py:121-130) for a whole batch in three kernel launches (``sat_image_batch_transform``): the
  pictures are resampled with Pillow's BILINEAR arithmetic bit for bit, the random draws are made on the host in the
  reference's order (crop parameters, then the flip coin, per picture).
* ``DeviceLoader``         train.py:244-259 DataLoader(pin_memory=True): decode threads -> one pinned staging buffer per
  batch -> one H2D copy on a side stream -> transform; batches arrive as ``(img, caps, lengths)`` on the device.

ColorJitter / RandomPerspective / RandomAffine / RandomRotation (train.py:223-231, off by default) are not built.
There is no CPU path: the transform needs libsat_hip.so and a GPU.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
import queue
import threading
from collections import OrderedDict
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _lib as L


def json_loader(path):
    """util.py:132-133"""
    with open(path) as f:
        return json.load(f)


def decode_rgb(path):
    """util.py:136-137 pil_loader, returned as an (H, W, 3) uint8 array."""
    from PIL import Image
    with open(path, "rb") as f:
        return np.asarray(Image.open(f).convert("RGB"))


class CocoCaptionDataset:
    """util.py:16-45.  ``root`` is prepended to relative ``img_paths``."""

    def __init__(self, jsonpath, split="train", root=None, decode=decode_rgb):
        self.json = json_loader(jsonpath) if isinstance(jsonpath, (str, os.PathLike)) else jsonpath
        self.split = split
        self.vocab_stoi = self.json["vocab_stoi"]
        self.vocab_itos = {v: k for k, v in self.vocab_stoi.items()}
        self.img_paths = self.json[split]["img_paths"]
        self.encoded_captions = self.json[split]["encoded_captions"]
        self.lengths = self.json[split]["lengths"]
        assert len(self.img_paths) == len(self.encoded_captions) == len(self.lengths)
        self.root, self.decode = root, decode

    def stoi(self, s):
        return int(self.vocab_stoi.get(s, self.vocab_stoi["<UNK>"]))

    def itos(self, i):
        return str(self.vocab_itos.get(int(i), "<UNK>"))

    def __len__(self):
        return len(self.img_paths)

    def __getitem__(self, idx):
        path = self.img_paths[idx]
        if self.root is not None and not os.path.isabs(path):
            path = os.path.join(self.root, path)
        return self.decode(path), torch.LongTensor(self.encoded_captions[idx]), torch.LongTensor(self.lengths[idx])


class BucketSampler:
    """util.py:48-87: indices grouped by the sample's total target count (sum of its caption lengths), groups in
    decreasing count, each group shuffled in place with ``np.random.shuffle`` at every ``__iter__``.

    ``world_size > 1``: every rank builds the same order (``seed`` + epoch drive a private RandomState instead of the
    global numpy generator), pads it to a multiple of ``batch_size * world_size`` by wrapping, and yields, of every run of
    ``batch_size * world_size`` indices, the ``batch_size`` that belong to ``rank``: the ranks' batches hold neighbouring
    lengths, and every rank sees the same number of batches."""

    def __init__(self, lengths, batch_size, indices=None, rank=0, world_size=1, seed=None):
        self.lengths, self.batch_size = lengths, batch_size
        self.indices = indices if indices else list(range(len(lengths)))
        assert 0 <= rank < world_size
        self.rank, self.world_size, self.seed, self.epoch = rank, world_size, seed, 0
        if world_size > 1 and seed is None:
            self.seed = 0
        len_map = OrderedDict()
        for i, length_list in zip(self.indices, self.lengths):
            len_map.setdefault(sum(length_list), []).append(i)
        self.grouped_indices = [idxs for _, idxs in reversed(sorted(len_map.items()))]

    def set_epoch(self, epoch):
        self.epoch = epoch

    def global_order(self):
        rs = np.random if self.seed is None else np.random.RandomState(self.seed + self.epoch)
        order = []
        for idxs in self.grouped_indices:
            if self.seed is not None:
                idxs = list(idxs)          # a seeded epoch does not depend on the epochs before it
            rs.shuffle(idxs)
            order.extend(idxs)
        return order

    def __iter__(self):
        order = self.global_order()
        if self.world_size == 1:
            return iter(order)
        run = self.batch_size * self.world_size
        padded = int(math.ceil(len(order) / run)) * run
        order = (order * (padded // len(order) + 1))[:padded]
        mine = []
        for start in range(0, padded, run):
            mine.extend(order[start + self.rank * self.batch_size: start + (self.rank + 1) * self.batch_size])
        return iter(mine)

    def __len__(self):
        if self.world_size == 1:
            return len(self.lengths)
        run = self.batch_size * self.world_size
        return int(math.ceil(len(self.lengths) / run)) * self.batch_size


# ----------------------------------------------------------------------------------------------------------------------
def resize_rule(h, w, size):
    """T.Resize(int): the smaller edge becomes ``size`` (torchvision 0.10 F.resize)."""
    if w <= h:
        return int(size * h / w), size
    return size, int(size * w / h)


def random_resized_crop_params(h, w, scale, ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """T.RandomResizedCrop.get_params (torchvision 0.10), drawing from the global torch CPU generator."""
    area = h * w
    log_ratio = torch.log(torch.tensor(ratio))
    for _ in range(10):
        target_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
        aspect = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
        cw = int(round(math.sqrt(target_area * aspect)))
        ch = int(round(math.sqrt(target_area / aspect)))
        if 0 < cw <= w and 0 < ch <= h:
            i = torch.randint(0, h - ch + 1, size=(1,)).item()
            j = torch.randint(0, w - cw + 1, size=(1,)).item()
            return i, j, ch, cw
    in_ratio = float(w) / float(h)
    if in_ratio < min(ratio):
        cw = w
        ch = int(round(cw / min(ratio)))
    elif in_ratio > max(ratio):
        ch = h
        cw = int(round(ch * max(ratio)))
    else:
        cw, ch = w, h
    return (h - ch) // 2, (w - cw) // 2, ch, cw


def box_desc(h, w, box, size, flip=False):
    """descriptor fields of: crop ``box`` = (top, left, height, width) -> resize to size x size (-> flip)"""
    t, l, ch, cw = box
    return dict(height=h, width=w, crop_top=t, crop_left=l, crop_h=ch, crop_w=cw, resized_h=size, resized_w=size, out_top=0, out_left=0, flip=int(flip))


def center_desc(h, w, size):
    """descriptor fields of T.Resize(size) -> T.CenterCrop(size)"""
    rh, rw = resize_rule(h, w, size)
    return dict(height=h, width=w, crop_top=0, crop_left=0, crop_h=h, crop_w=w, resized_h=rh, resized_w=rw,
                out_top=int(round((rh - size) / 2.0)), out_left=int(round((rw - size) / 2.0)), flip=0)


class StagedBatch:
    """The decoded pictures of one batch in one pinned host buffer: [descriptors | pixels]."""

    def __init__(self, images, descs):
        n = len(images)
        self.n = n
        self.desc = (L.ImageDesc * n)()
        head = (C.sizeof(L.ImageDesc) * n + 255) // 256 * 256
        total = head + sum(int(im.shape[0]) * int(im.shape[1]) * 3 for im in images)
        self.host = torch.empty(total, dtype=torch.uint8).pin_memory() if torch.cuda.is_available() else torch.empty(total, dtype=torch.uint8)
        buf = self.host.numpy()
        off = 0
        for i, (im, d) in enumerate(zip(images, descs)):
            if im.ndim != 3 or im.shape[2] != 3 or im.dtype != np.uint8:
                raise ValueError("picture %d: expected (H, W, 3) uint8, got %s %s" % (i, im.shape, im.dtype))
            nb = im.shape[0] * im.shape[1] * 3
            buf[head + off: head + off + nb] = im.reshape(-1)
            e = self.desc[i]
            e.offset = off
            for k, v in d.items():
                setattr(e, k, int(v))
            off += nb
        buf[:C.sizeof(L.ImageDesc) * n] = np.frombuffer(self.desc, dtype=np.uint8)
        self.head, self.pixels_bytes = head, off


class BatchTransform:
    """train.py:208-233 for a batch.  ``train=False``: Resize + CenterCrop + ToTensor (valid_transforms).
    ``train=True``: aug_scale == 1 -> Resize + CenterCrop, else RandomResizedCrop(scale=(aug_scale, 1)); a flip with
    probability aug_hflip when 0 < aug_hflip < 1; ToTensor; + N(0,1) * aug_noise_std.
    ``randn(shape, device)`` supplies the noise draws (default: ``torch.randn`` on the device)."""

    def __init__(self, input_size, train=True, aug_scale=0.9, aug_hflip=0.5, aug_noise_std=0.01, randn=None):
        if train and not (0 <= aug_scale <= 1.0):
            raise ValueError("Invalid value for aug_scale. Choose in the range {0,1}.")       # train.py:219-220
        self.size, self.train = int(input_size), train
        self.aug_scale, self.aug_hflip, self.noise_std, self.randn = aug_scale, aug_hflip, aug_noise_std, randn
        self._ws = None

    def draw(self, shapes):
        """one descriptor per picture; consumes the torch CPU generator in the reference's per-sample order"""
        out = []
        for h, w in shapes:
            if not self.train or self.aug_scale == 1.0:
                d = center_desc(h, w, self.size)
            else:
                d = box_desc(h, w, random_resized_crop_params(h, w, (self.aug_scale, 1.0)), self.size)
            if self.train and 0 < self.aug_hflip < 1.0:
                d["flip"] = int(torch.rand(1).item() < self.aug_hflip)            # T.RandomHorizontalFlip.forward
            out.append(d)
        return out

    def stage(self, images, descs=None):
        return StagedBatch(images, descs if descs is not None else self.draw([im.shape[:2] for im in images]))

    def run(self, staged, device, stream=None, noise=None, want_bytes=False):
        """H2D copy of the staged batch + the kernels, on ``stream`` (default: current).  Returns the (n, 3, S, S) fp32
        batch (and the (n, S, S, 3) bytes PIL would hold when ``want_bytes``)."""
        lib = L.lib()
        device = torch.device(device)
        if device.type != "cuda":
            raise L.SatHipError("sat_amd computes on the GPU only: the batch transform got device %s (no CPU fallback)" % device)
        stream = stream if stream is not None else torch.cuda.current_stream(device)
        S, n = self.size, staged.n
        with torch.cuda.stream(stream):
            dev = staged.host.to(device, non_blocking=True)
            need = lib.sat_image_batch_workspace_bytes(C.cast(staged.desc, C.c_void_p), n, S, S)
            if need == 0:
                L.check(1, "sat_image_batch_workspace_bytes")
            if self._ws is None or self._ws.numel() < need or self._ws.device != dev.device:
                self._ws = torch.empty(need, dtype=torch.uint8, device=device)
            out = torch.empty(n, 3, S, S, dtype=torch.float32, device=device)
            raw = torch.empty(n, S, S, 3, dtype=torch.uint8, device=device) if want_bytes else None
            if noise is None and self.train and self.noise_std:
                noise = (self.randn or (lambda shape, device: torch.randn(shape, device=device)))((n, 3, S, S), device)
            if noise is not None:
                L.require_gpu(noise)
                assert noise.shape == out.shape and noise.dtype == torch.float32 and noise.is_contiguous()
            L.check(lib.sat_image_batch_transform(dev.data_ptr() + staged.head, staged.pixels_bytes, C.cast(staged.desc, C.c_void_p), dev.data_ptr(), n, S, S,
                                                  L.ptr(noise) if noise is not None else None, float(self.noise_std if noise is not None else 0.0),
                                                  L.ptr(out), L.ptr(raw) if raw is not None else None, L.ptr(self._ws), self._ws.numel(),
                                                  C.c_void_p(stream.cuda_stream)), "sat_image_batch_transform")
            dev.record_stream(stream)
        return (out, raw) if want_bytes else out

    def __call__(self, images, device="cuda", descs=None, noise=None):
        return self.run(self.stage(images, descs), torch.device(device), noise=noise)


class DeviceLoader:
    """train.py:244-259: ``DataLoader(dataset, sampler=BucketSampler | shuffle, batch_size, num_workers, pin_memory=True)``
    with the transform moved behind the H2D copy.  Decoding runs in ``workers`` threads (PIL releases the GIL), staging and
    the copy + kernels of batch i+1 overlap the consumer's work on batch i (side stream, ``prefetch`` batches in flight)."""

    def __init__(self, dataset, batch_size, transform, sampler=None, shuffle=False, workers=4, prefetch=2, device="cuda", drop_last=False):
        self.ds, self.batch_size, self.tf, self.sampler, self.shuffle = dataset, batch_size, transform, sampler, shuffle
        self.workers, self.prefetch, self.device, self.drop_last = max(1, workers), max(1, prefetch), torch.device(device), drop_last

    def __len__(self):
        n = len(self.sampler) if self.sampler is not None else len(self.ds)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _index_batches(self):
        if self.sampler is not None:
            order = list(iter(self.sampler))
        elif self.shuffle:
            order = torch.randperm(len(self.ds)).tolist()
        else:
            order = list(range(len(self.ds)))
        for s in range(0, len(order), self.batch_size):
            b = order[s:s + self.batch_size]
            if len(b) == self.batch_size or not self.drop_last:
                yield b

    def __iter__(self):
        L.lib()
        side = torch.cuda.Stream(self.device)
        q = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()

        def produce():
            try:
                with ThreadPoolExecutor(self.workers) as pool:
                    for idxs in self._index_batches():
                        if stop.is_set():
                            return
                        samples = list(pool.map(self.ds.__getitem__, idxs))
                        staged = self.tf.stage([s[0] for s in samples])
                        caps = torch.stack([s[1] for s in samples])
                        lens = torch.stack([s[2] for s in samples])
                        img = self.tf.run(staged, self.device, stream=side)
                        with torch.cuda.stream(side):
                            caps_d = caps.pin_memory().to(self.device, non_blocking=True)
                            lens_d = lens.pin_memory().to(self.device, non_blocking=True)
                            done = torch.cuda.Event()
                            done.record(side)
                        q.put((img, caps_d, lens_d, done, staged))
                q.put(None)
            except BaseException as e:          # surfaces in the consumer
                q.put(e)

        th = threading.Thread(target=produce, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                img, caps, lens, done, _staged = item
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(done)
                for t in (img, caps, lens):
                    t.record_stream(cur)
                yield img, caps, lens
        finally:
            stop.set()
            while th.is_alive():
                try:
                    q.get(timeout=0.05)
                except queue.Empty:
                    pass
