"""CPU: input pipeline.  The oracle (oracle/image_oracle.py) against Pillow's own resize and the
reference's util.py helpers (tests/golden/g11_input_pipeline.npz, made by tests/golden/make_golden.py), and the host
logic of sat_amd.data (sampler, staging, random draws).  No kernel runs here."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import image_oracle as IO


@pytest.fixture(scope="module")
def g11(golden_dir):
    return np.load(os.path.join(golden_dir, "g11_input_pipeline.npz"), allow_pickle=False)


def test_oracle_resample_equals_pillow(g11):
    for i, (h, w, oh, ow) in enumerate(g11["resize_cases"]):
        got = IO.resample_u8(g11["resize_in%d" % i], int(oh), int(ow))
        assert np.array_equal(got, g11["resize_out%d" % i]), "case %d (%dx%d -> %dx%d)" % (i, h, w, oh, ow)     # bytes: bit exact


def test_oracle_crop_then_resample_equals_pillow(g11):
    img = g11["crop_in"]
    for i, (t, l, h, w) in enumerate(g11["crop_boxes"]):
        assert np.array_equal(IO.resample_u8(img[t:t + h, l:l + w], 28, 28), g11["crop_out%d" % i])


def test_oracle_helpers_equal_reference_util(g11):
    for i in range(3):
        assert np.array_equal(IO.crop_max_square(g11["square_in%d" % i]), g11["square_out%d" % i])
    # AddGaussianNoise (util.py:121-130): tensor + randn(size) * std from the global CPU generator
    x = torch.from_numpy(g11["noise_in"])
    torch.manual_seed(9)
    assert np.array_equal((x + torch.randn(x.size()) * 0.01).numpy(), g11["noise_out"])
    groups = IO.bucket_groups(g11["bucket_lengths"].tolist())
    np.random.seed(77)
    assert IO.bucket_epoch(groups, np.random.shuffle) == g11["bucket_epoch0"].tolist()
    assert IO.bucket_epoch(groups, np.random.shuffle) == g11["bucket_epoch1"].tolist()


def test_oracle_live_against_installed_pillow():
    """extra sizes (incl. photographic 480x640 -> 224) against the Pillow in this image, when it is importable"""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    for h, w, oh, ow in ((480, 640, 224, 224), (333, 500, 224, 336), (64, 48, 224, 224), (9, 300, 8, 8)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(IO.resample_u8(img, oh, ow), np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR)))


# ------------------------------------------------------------------------------------------------ host logic of the product
def test_bucket_sampler_matches_reference_order(g11):
    import sat_amd  # noqa: F401
    from sat_amd.data import BucketSampler
    lengths = g11["bucket_lengths"].tolist()
    s = BucketSampler(lengths, 8)
    np.random.seed(77)
    assert list(iter(s)) == g11["bucket_epoch0"].tolist()
    assert list(iter(s)) == g11["bucket_epoch1"].tolist()
    assert len(s) == int(g11["bucket_len"])


def test_bucket_sampler_ranks_partition_the_epoch(g11):
    import sat_amd  # noqa: F401
    from sat_amd.data import BucketSampler
    lengths = g11["bucket_lengths"].tolist()
    W, B = 3, 4
    per_rank = []
    for r in range(W):
        s = BucketSampler(lengths, B, rank=r, world_size=W, seed=11)
        s.set_epoch(2)
        per_rank.append(list(iter(s)))
        assert len(per_rank[-1]) == len(s)
    assert len({len(p) for p in per_rank}) == 1                      # same number of batches everywhere
    allidx = [i for p in per_rank for i in p]
    assert set(allidx) == set(range(len(lengths)))                   # every sample seen
    assert len(allidx) - len(lengths) < W * B                        # padding below one global batch
    ref = BucketSampler(lengths, B, rank=0, world_size=W, seed=11)
    ref.set_epoch(2)
    order = ref.global_order()
    tot = [sum(l) for l in lengths]
    assert [tot[i] for i in order] == sorted(tot, reverse=True)      # longest first, as the reference
    for k in range(len(per_rank[0]) // B):                           # batch k of every rank comes from one run of the order
        run = order[k * W * B:(k + 1) * W * B]
        if len(run) == W * B:
            assert [i for r in range(W) for i in per_rank[r][k * B:(k + 1) * B]] == run
    again = BucketSampler(lengths, B, rank=1, world_size=W, seed=11)
    again.set_epoch(2)
    assert list(iter(again)) == per_rank[1]                          # reproducible per (seed, epoch)
    again.set_epoch(3)
    assert list(iter(again)) != per_rank[1]


def test_random_draws_follow_the_restated_rules():
    import sat_amd  # noqa: F401
    from sat_amd import data as D
    shapes = [(480, 640), (640, 427), (100, 100), (37, 200)]
    tf = D.BatchTransform(64, train=True, aug_scale=0.5, aug_hflip=0.5)
    torch.manual_seed(5)
    got = tf.draw(shapes)
    torch.manual_seed(5)
    for (h, w), d in zip(shapes, got):
        t, l, ch, cw = IO.random_resized_crop_params(h, w, (0.5, 1.0))
        flip = int(torch.rand(1).item() < 0.5)
        assert (d["crop_top"], d["crop_left"], d["crop_h"], d["crop_w"], d["flip"]) == (t, l, ch, cw, flip)
        if 3 / 4 <= w / h <= 4 / 3 or (h, w) == (480, 640):             # not the centre-crop fall-back of extreme aspects
            assert 0.5 * h * w * 0.98 <= ch * cw <= h * w and 3 / 4 * 0.95 <= cw / ch <= 4 / 3 * 1.05
    for h, w in shapes:                                              # valid / aug_scale == 1: Resize + CenterCrop
        d = D.center_desc(h, w, 64)
        rh, rw = IO.resize_rule(h, w, 64)
        assert (d["resized_h"], d["resized_w"]) == (rh, rw) and min(rh, rw) == 64
        assert (d["out_top"], d["out_left"]) == IO.center_crop_offsets(rh, rw, 64, 64)
    with pytest.raises(ValueError):
        D.BatchTransform(64, train=True, aug_scale=1.5)


def test_staged_batch_layout_and_dataset(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    import ctypes
    import sat_amd  # noqa: F401
    from sat_amd import _lib as L, data as D
    rng = np.random.default_rng(0)
    paths, pics = [], []
    for i, (h, w) in enumerate([(20, 31), (33, 17), (8, 8)]):
        im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        Image.fromarray(im).save(tmp_path / ("p%d.png" % i))
        paths.append("p%d.png" % i); pics.append(im)
    meta = {"vocab_stoi": {"<PAD>": 0, "<START>": 1, "<END>": 2, "<UNK>": 3, "a": 4},
            "train": {"img_paths": paths, "encoded_captions": [[[1, 4, 2, 0], [1, 4, 4, 2]]] * 3, "lengths": [[3, 4]] * 3}}
    (tmp_path / "d.json").write_text(json.dumps(meta))
    ds = D.CocoCaptionDataset(str(tmp_path / "d.json"), "train", root=str(tmp_path))
    assert len(ds) == 3 and ds.stoi("zzz") == 3 and ds.itos(4) == "a" and ds.itos(99) == "<UNK>"
    img, caps, lens = ds[1]
    assert np.array_equal(img, pics[1]) and caps.shape == (2, 4) and caps.dtype == torch.int64 and lens.tolist() == [3, 4]
    tf = D.BatchTransform(16, train=False)
    st = tf.stage([ds[i][0] for i in range(3)])
    buf = st.host.numpy()
    assert st.head % 256 == 0 and st.pixels_bytes == sum(p.size for p in pics)
    back = (L.ImageDesc * 3).from_buffer_copy(buf[:ctypes.sizeof(L.ImageDesc) * 3].tobytes())
    for i, p in enumerate(pics):
        assert (back[i].height, back[i].width) == p.shape[:2]
        assert np.array_equal(buf[st.head + back[i].offset: st.head + back[i].offset + p.size].reshape(p.shape), p)
    with pytest.raises(ValueError):
        tf.stage([np.zeros((4, 4), np.uint8)])
    with pytest.raises(L.SatHipError):                               # no GPU here: the transform refuses, no CPU path
        tf.run(st, torch.device("cpu"))
