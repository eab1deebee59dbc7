"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's input pipeline
(SURVEY.md section 8 row f3).  Never imported by the product package.

What the reference does per sample (train.py:208-233, util.py:16-45,121-130):
``pil_loader`` -> ``T.RandomResizedCrop`` | ``T.Resize + T.CenterCrop`` ->
``T.RandomHorizontalFlip`` -> ``T.ToTensor`` -> ``AddGaussianNoise``.  On PIL images
torchvision's resize is ``PIL.Image.resize(size, BILINEAR)``, i.e. Pillow's
antialiased two-pass resampler (third party: Pillow ``src/libImaging/Resample.c``;
12.2.0 is the version in this image).  ``resample_u8`` restates that published algorithm
in numpy integer arithmetic; ``tests/golden/g11_input_pipeline.npz`` pins it bit for bit
against Pillow itself, and pins ``BucketSampler`` / ``AddGaussianNoise`` /
``crop_max_square`` against the reference's own ``util.py`` run in the build container.

torchvision is absent from the image: the crop-parameter rules (``get_params`` of
RandomResizedCrop, the size rule of Resize, the offsets of CenterCrop) are restated
from torchvision 0.10's published source and are **parity unpinned**.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2      # Resample.c: 8-bit pixels, 2 bits of head room


def _triangle(x):
    """Resample.c bilinear_filter."""
    x = np.abs(x)
    return np.where(x < 1.0, 1.0 - x, 0.0)


def resample_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the box [0, in_size).
    Returns (bounds (out,2) int32 = first tap / tap count, coeffs (out,ksize) int32)."""
    scale = float(in_size) / float(out_size)
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        x = np.arange(xmax, dtype=np.float64)
        w = _triangle(((x + xmin) - center + 0.5) * ss)
        ww = 0.0
        for v in w:                       # sequential sum, as the C loop
            ww += float(v)
        if ww != 0.0:
            w = w / ww
        kk[xx, :xmax] = (0.5 + w * float(1 << PRECISION_BITS)).astype(np.int64)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """One 8-bit pass of ImagingResampleHorizontal/Vertical_8bpc along ``axis``."""
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + img.shape[1:], np.uint8)
    for xx, (xmin, xmax) in enumerate(bounds):
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(xmax):
            acc += img[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis)


def resample_u8(img, out_h, out_w):
    """``Image.fromarray(img).resize((out_w, out_h), BILINEAR)`` for an (H, W, C) uint8 array:
    horizontal pass to 8 bits, then vertical pass (Resample.c ImagingResample)."""
    h, w = img.shape[:2]
    if w != out_w:
        img = _pass(img, *resample_coeffs(w, out_w), axis=1)
    if h != out_h:
        img = _pass(img, *resample_coeffs(h, out_h), axis=0)
    return img


# ------------------------------------------------------------------ transform parameters
def resize_rule(h, w, size):
    """torchvision 0.10 F.resize with an int size: the smaller edge becomes ``size``."""
    if w <= h:
        return int(size * h / w), size
    return size, int(size * w / h)


def center_crop_offsets(h, w, ch, cw):
    """torchvision F.center_crop (image at least as large as the crop)."""
    return int(round((h - ch) / 2.0)), int(round((w - cw) / 2.0))


def random_resized_crop_params(h, w, scale, ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """torchvision 0.10 RandomResizedCrop.get_params; consumes the global torch CPU generator
    exactly as it does (uniform area, uniform log-ratio, then two randints per try)."""
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


def train_transform(img, size, box, flip, noise, std):
    """train.py:214-233 with the default switches, the random draws made by the caller:
    crop ``box`` = (top, left, h, w) -> resize to size x size -> optional flip -> ToTensor
    -> + noise * std (util.py:121-130).  Returns (3, size, size) float32."""
    t, l, h, w = box
    out = resample_u8(img[t:t + h, l:l + w], size, size)
    if flip:
        out = out[:, ::-1]
    x = torch.from_numpy(np.ascontiguousarray(out)).permute(2, 0, 1).float().div(255)
    if noise is not None:
        x = x + noise * std
    return x


def valid_transform(img, size):
    """train.py:208-212: Resize(size) -> CenterCrop(size) -> ToTensor."""
    h, w = img.shape[:2]
    rh, rw = resize_rule(h, w, size)
    out = resample_u8(img, rh, rw)
    t, l = center_crop_offsets(rh, rw, size, size)
    out = out[t:t + size, l:l + size]
    return torch.from_numpy(np.ascontiguousarray(out)).permute(2, 0, 1).float().div(255)


def crop_max_square(img):
    """util.py:146-164 crop_center / crop_max_square with size=None on an (H, W, C) array."""
    h, w = img.shape[:2]
    c = min(h, w)
    return img[(h - c) // 2:(h + c) // 2, (w - c) // 2:(w + c) // 2]


# ------------------------------------------------------------------ sampler
def bucket_groups(lengths):
    """util.py:52-74 BucketSampler.__init__: samples grouped by their total target count
    (sum of the caption lengths), groups in decreasing count."""
    len_map = OrderedDict()
    for i, ls in enumerate(lengths):
        len_map.setdefault(sum(ls), []).append(i)
    return [idxs for _, idxs in reversed(sorted(len_map.items()))]


def bucket_epoch(groups, shuffle):
    """util.py:76-84 BucketSampler.__iter__: every group shuffled IN PLACE by ``shuffle``
    (np.random.shuffle in the reference; the next epoch starts from this epoch's permutation),
    groups concatenated."""
    order = []
    for idxs in groups:
        shuffle(idxs)
        order.extend(idxs)
    return order
