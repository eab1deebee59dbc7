"""GPU: sat_image_batch_transform / sat_amd.data against the oracle (oracle/image_oracle.py, itself pinned bit for bit to
Pillow by g11) - byte and fp32 results are compared EXACTLY (integer arithmetic; the fp32 tail is one division, one
multiply and one add, each correctly rounded)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import image_oracle as IO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def D():
    import sat_amd  # noqa: F401
    from sat_amd import data
    return data


@pytest.fixture(scope="module")
def g11(golden_dir):
    return np.load(os.path.join(golden_dir, "g11_input_pipeline.npz"), allow_pickle=False)


def whole(h, w, oh, ow):
    return dict(height=h, width=w, crop_top=0, crop_left=0, crop_h=h, crop_w=w, resized_h=oh, resized_w=ow, out_top=0, out_left=0, flip=0)


def test_resample_bytes_equal_pillow_fixture(D, g11):
    for i, (h, w, oh, ow) in enumerate(g11["resize_cases"].tolist()):
        if oh != ow:
            continue                                     # the mirror transform makes square batches; rectangles go through the C call below
        tf = D.BatchTransform(oh, train=False)
        out, raw = tf.run(tf.stage([g11["resize_in%d" % i]], [whole(h, w, oh, ow)]), torch.device("cuda"), want_bytes=True)
        assert np.array_equal(raw[0].cpu().numpy(), g11["resize_out%d" % i]), "case %d" % i
        want = torch.from_numpy(g11["resize_out%d" % i]).permute(2, 0, 1).float().div(255)
        assert torch.equal(out[0].cpu(), want)


def test_rectangular_output_through_the_c_abi(D, g11):
    import ctypes as C
    from sat_amd import _lib as L
    lib = L.lib()
    for i, (h, w, oh, ow) in enumerate(g11["resize_cases"].tolist()):
        st = D.StagedBatch([g11["resize_in%d" % i]], [whole(h, w, oh, ow)])
        dev = st.host.cuda()
        need = lib.sat_image_batch_workspace_bytes(C.cast(st.desc, C.c_void_p), 1, oh, ow)
        assert need > 0
        ws = torch.empty(need, dtype=torch.uint8, device="cuda")
        raw = torch.empty(1, oh, ow, 3, dtype=torch.uint8, device="cuda")
        L.check(lib.sat_image_batch_transform(dev.data_ptr() + st.head, st.pixels_bytes, C.cast(st.desc, C.c_void_p), dev.data_ptr(), 1, oh, ow, None, 0.0,
                                              None, L.ptr(raw), L.ptr(ws), need, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "transform")
        assert np.array_equal(raw[0].cpu().numpy(), g11["resize_out%d" % i]), "case %d" % i


def test_crop_boxes_equal_pillow_fixture(D, g11):
    img = g11["crop_in"]
    boxes = g11["crop_boxes"].tolist()
    tf = D.BatchTransform(28, train=True)
    descs = [D.box_desc(img.shape[0], img.shape[1], b, 28) for b in boxes]
    _, raw = tf.run(tf.stage([img] * len(boxes), descs), torch.device("cuda"), want_bytes=True, noise=torch.zeros(len(boxes), 3, 28, 28, device="cuda"))
    for i in range(len(boxes)):
        assert np.array_equal(raw[i].cpu().numpy(), g11["crop_out%d" % i])


def test_ragged_train_batch_equals_oracle(D):
    """mixed sizes, up- and down-scaling, flips and noise in one batch; fp32 output compared bit for bit"""
    rng = np.random.default_rng(21)
    shapes = [(480, 640), (640, 427), (100, 100), (37, 200), (224, 224), (60, 45), (500, 333), (3, 3)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    S = 96
    tf = D.BatchTransform(S, train=True, aug_scale=0.3, aug_hflip=0.5, aug_noise_std=0.01)
    torch.manual_seed(8)
    descs = tf.draw(shapes)
    assert any(d["flip"] for d in descs) and not all(d["flip"] for d in descs)
    noise = torch.randn(len(imgs), 3, S, S)
    out = tf.run(tf.stage(imgs, descs), torch.device("cuda"), noise=noise.cuda()).cpu()
    for i, (im, d) in enumerate(zip(imgs, descs)):
        want = IO.train_transform(im, S, (d["crop_top"], d["crop_left"], d["crop_h"], d["crop_w"]), d["flip"], noise[i], 0.01)
        assert torch.equal(out[i], want), "picture %d: max|d| = %g" % (i, (out[i] - want).abs().max())


def test_valid_batch_equals_oracle(D):
    rng = np.random.default_rng(22)
    shapes = [(480, 640), (640, 480), (333, 500), (64, 64), (65, 200), (40, 30)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    tf = D.BatchTransform(64, train=False)
    out = tf(imgs).cpu()
    for i, im in enumerate(imgs):
        assert torch.equal(out[i], IO.valid_transform(im, 64)), "picture %d" % i
    same = D.BatchTransform(64, train=True, aug_scale=1.0, aug_hflip=0.0, aug_noise_std=0.0)(imgs).cpu()     # train.py:215-216
    assert torch.equal(same, out)


def test_full_size_batch_properties(D):
    """C2-sized batch (128 pictures of 480x640 -> 224): properties that need no CPU resampling of the whole batch"""
    g = torch.Generator().manual_seed(3)
    B, H, W, S = 128, 480, 640, 224
    base = torch.randint(0, 256, (4, H, W, 3), dtype=torch.uint8, generator=g).numpy()
    imgs = [base[i % 4] for i in range(B)]
    imgs[5] = np.full((H, W, 3), 77, np.uint8)                               # a flat picture stays flat
    imgs[6] = np.ascontiguousarray(base[0][:S, :S])                          # box of the output size: resampling is the identity
    tf = D.BatchTransform(S, train=True)
    descs = [D.box_desc(H, W, (7, 13, 400, 560), S, flip=((i // 4) % 2 == 1)) for i in range(B)]      # pictures 4-7, 12-15, ... mirrored
    descs[6] = D.box_desc(S, S, (0, 0, S, S), S)
    out, raw = tf.run(tf.stage(imgs, descs), torch.device("cuda"), noise=torch.zeros(B, 3, S, S, device="cuda"), want_bytes=True)
    assert torch.equal(raw[0], raw[8]) and torch.equal(raw[7], raw[15])      # same picture, same box, same flip -> same bytes
    assert torch.equal(raw[4], raw[0].flip(1)) and torch.equal(raw[7], raw[3].flip(1))        # the flip is an exact mirror
    assert bool((raw[5] == 77).all())
    assert np.array_equal(raw[6].cpu().numpy(), imgs[6])
    assert torch.equal(out.cpu(), raw.cpu().permute(0, 3, 1, 2).float().div(255))      # ToTensor on the host: correctly rounded division
    # the oracle on two pictures of the batch
    assert np.array_equal(raw[1].cpu().numpy(), IO.resample_u8(imgs[1][7:407, 13:573], S, S))
    assert np.array_equal(raw[127].cpu().numpy(), IO.resample_u8(imgs[127][7:407, 13:573], S, S)[:, ::-1])


def test_rejects_bad_descriptors(D):
    from sat_amd import _lib as L
    img = np.zeros((10, 12, 3), np.uint8)
    tf = D.BatchTransform(8, train=True)
    for bad in (dict(crop_h=11), dict(crop_left=5, crop_w=8), dict(resized_h=7), dict(out_left=1), dict(crop_h=0)):
        d = D.box_desc(10, 12, (0, 0, 10, 12), 8)
        d.update(bad)
        with pytest.raises(L.SatHipError):
            tf.run(tf.stage([img], [d]), torch.device("cuda"))
    big = np.zeros((8 * 70, 8, 3), np.uint8)                                              # > 64x shrink
    with pytest.raises(L.SatHipError):
        tf.run(tf.stage([big], [D.box_desc(560, 8, (0, 0, 560, 8), 8)]), torch.device("cuda"))


def test_device_loader_end_to_end(D, tmp_path):
    """JSON dataset on disk -> bucket sampler -> decode threads -> staged H2D -> kernels; every batch equals the oracle's
    per-sample valid transform, captions and lengths arrive as the default collate would stack them."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(4)
    n, R, T = 11, 2, 6
    paths, pics, caps, lens = [], [], [], []
    for i in range(n):
        h, w = int(rng.integers(20, 60)), int(rng.integers(20, 60))
        im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        Image.fromarray(im).save(tmp_path / ("p%d.png" % i))
        paths.append("p%d.png" % i); pics.append(im)
        ls = rng.integers(3, T + 1, size=R).tolist()
        lens.append(ls)
        caps.append([[1] + [4] * (l - 2) + [2] + [0] * (T - l) for l in ls])
    meta = {"vocab_stoi": {"<PAD>": 0, "<START>": 1, "<END>": 2, "<UNK>": 3, "a": 4},
            "val": {"img_paths": paths, "encoded_captions": caps, "lengths": lens}}
    (tmp_path / "d.json").write_text(json.dumps(meta))
    ds = D.CocoCaptionDataset(str(tmp_path / "d.json"), "val", root=str(tmp_path))
    sampler = D.BucketSampler(ds.lengths, 4, seed=1)
    order = sampler.global_order()
    loader = D.DeviceLoader(ds, 4, D.BatchTransform(16, train=False), sampler=sampler, workers=3)
    assert len(loader) == 3
    seen = 0
    for b, (img, c, l) in enumerate(loader):
        idxs = order[b * 4:(b + 1) * 4]
        assert img.is_cuda and img.shape == (len(idxs), 3, 16, 16) and c.shape == (len(idxs), R, T) and l.shape == (len(idxs), R)
        for j, i in enumerate(idxs):
            assert torch.equal(img[j].cpu(), IO.valid_transform(pics[i], 16))
            assert c[j].tolist() == caps[i] and l[j].tolist() == lens[i]
        seen += len(idxs)
    assert seen == n


def test_loader_batches_feed_the_train_step(D, tmp_path):
    """the README loop: DeviceLoader batches (img, caps, lengths on the device) go straight into SAT.training_step"""
    Image = pytest.importorskip("PIL.Image")
    import sat_amd  # noqa: F401
    from sat_amd import model as M
    from oracle import sat_oracle as O
    rng = np.random.default_rng(9)
    n, R, T, V = 8, 2, 7, 30
    paths, caps, lens = [], [], []
    for i in range(n):
        Image.fromarray(rng.integers(0, 256, (int(rng.integers(70, 110)), int(rng.integers(70, 110)), 3), dtype=np.uint8)).save(tmp_path / ("q%d.png" % i))
        paths.append("q%d.png" % i)
        ls = rng.integers(2, T, size=R).tolist()                     # targets per caption (words + <END>), at most T - 1 (preprocess.ipynb cell 17)
        lens.append(ls)
        caps.append([[V - 2] + rng.integers(1, V - 3, size=l - 1).tolist() + [V - 1] + [0] * (T - 1 - l) for l in ls])
    stoi = {"<PAD>": 0, "<UNK>": V - 3, "<START>": V - 2, "<END>": V - 1}
    (tmp_path / "d.json").write_text(json.dumps({"vocab_stoi": stoi, "train": {"img_paths": paths, "encoded_captions": caps, "lengths": lens}}))
    ds = D.CocoCaptionDataset(str(tmp_path / "d.json"), "train", root=str(tmp_path))
    loader = D.DeviceLoader(ds, 4, D.BatchTransform(64, train=True), sampler=D.BucketSampler(ds.lengths, 4, seed=3), workers=2)
    hp = O.default_hparams(encoder_arch="resnet18", encoder_dim=32, input_size=64, encoder_size=3, vocab_size=V, embed_dim=24, attention_dim=16,
                           decoder_dim=40, decoder_tf="always", weight_decay=0.0, decoder_lr=1e-3, embedding_lr=1e-2, encoder_lr=1e-5, opt="adam",
                           adam_b1=0.9, adam_b2=0.999, momentum=0.9, nesterov=False, scheduler=None)
    hp.vocab_stoi = stoi; hp.vocab_itos = {v: k for k, v in stoi.items()}
    torch.manual_seed(1)
    model = M.SAT(**vars(hp)).cuda().train()
    opt = model.configure_optimizers()
    seen = 0
    for batch in loader:
        assert all(t.is_cuda for t in batch)
        opt.zero_grad(set_to_none=True)
        out = model.training_step(batch, 0)
        out["loss"].backward()
        opt.step()
        assert bool(torch.isfinite(out["loss"]))
        seen += batch[0].shape[0]
    assert seen == n
