"""predict_class mirror (predict.py API) on the GPU vs the oracle / reference fixtures."""
import argparse

import numpy as np
import pytest
import torch

from oracle import patterns, postprocess as opost

pytestmark = pytest.mark.gpu


def _pc(size, conf, iou, float16=False, model=None):
    from objectdetection_amd.predict import predict_class
    a = argparse.Namespace(device='cuda', float16=float16, input_size=size, confidence_threshold=conf,
                           iou_threshold=iou)
    return predict_class(None, args=a, model=model)


def _model(size, ncls):
    from objectdetection_amd.model.yolov7 import yolov7
    net = yolov7(patterns.default_args(input_size=size, output_class=ncls))
    patterns.fill_state_dict(net)
    return net


def test_decode_nms_iou_api_match_reference_fixtures(golden):
    g = golden('nms.npz')
    for tag in ('spread', 'dense', 'edge', 'empty'):
        size, cthr, ithr, h, w = g[f'{tag}.cfg']
        pc = _pc(int(size), cthr, ithr, model=_model(64, 1))
        rows = pc._decode(g[f'{tag}.pred'].copy(), (int(h), int(w), 3))
        if len(g[f'{tag}.kept']) == 0:
            assert rows is None
        else:
            assert np.array_equal(rows, g[f'{tag}.rows'])
    pc = _pc(640, 0.5, 0.5, model=_model(64, 1))
    a = g['iou.a']
    assert np.array_equal(pc.iou(a[0:1], a), g['iou.val'], equal_nan=True)
    # NaN corners: np.maximum / np.minimum of predict.py:35-38 hand the NaN on to the IoU
    bad = a.copy()
    bad[1, 0] = np.nan
    bad[2, 1] = np.nan
    with np.errstate(all='ignore'):
        assert np.array_equal(pc.iou(bad[0:1], bad), opost.iou_xywh(bad[0], bad), equal_nan=True)
        assert np.array_equal(pc.iou(bad[1:2], bad), opost.iou_xywh(bad[1], bad), equal_nan=True)
    # stand-alone _nms on pre-filtered rows
    cand, idx = opost.filter_indices(g['spread.pred'], 0.5)
    keep = opost.nms_indices(cand, 0.5)
    assert np.array_equal(pc._nms(cand), cand[keep])


def test_predict_end_to_end_f32():
    size, ncls = 96, 3
    net = _model(size, ncls)
    pc = _pc(size, 0.45, 0.5, float16=False, model=net)
    img = (patterns.image_batch(1, size)[0].permute(1, 2, 0).numpy() * 255).astype(np.uint8)
    got = pc.predict(img)
    # identical pre-NMS tensor: the GPU's own deployed prediction, post-processed by the oracle
    x = torch.from_numpy((img.astype(np.float32) / 255)[None].transpose(0, 3, 1, 2).copy()).cuda()
    from objectdetection_amd import engine
    engine.set_compute_dtype('f32')
    pred = pc.model(x).cpu().numpy()[0]
    engine.set_compute_dtype(None)
    rows, kept = opost.predict_decode(pred, img.shape, size, 0.45, 0.5)
    assert rows is not None and len(rows) > 3
    assert np.array_equal(got, rows)


def test_predict_batch_bf16_runs_and_is_consistent():
    size, ncls = 64, 2
    pc = _pc(size, 0.3, 0.5, float16=True, model=_model(size, ncls))
    x = patterns.image_batch(3, size).cuda()
    out = pc.predict_batch(x)
    assert len(out) == 3
    single = pc.predict_batch(x[1:2])[0]
    assert (single is None and out[1] is None) or np.array_equal(single, out[1])


def test_detect_pipeline_matches_sequential():
    """The multi-stream pipeline (alternating plans) returns exactly what the one-stream path returns."""
    import torch
    from oracle import patterns
    from objectdetection_amd import engine, postprocess
    from objectdetection_amd.model.yolov7 import yolov7
    from objectdetection_amd.predict import detect_pipeline
    args = patterns.default_args(input_size=128, output_class=3)
    torch.manual_seed(5)
    model = yolov7(args).cuda().eval()
    engine.set_compute_dtype('bf16')
    try:
        g = torch.Generator().manual_seed(1)
        batches = [torch.rand(3, 3, 128, 128, generator=g).cuda() for _ in range(5)]
        want = []
        for x in batches:
            r = postprocess.postprocess_heads(model.head_tensors(x), 128, args.anchor, 0.5, 0.5)
            torch.cuda.synchronize()
            want.append((r.count.cpu(), r.idx.cpu(), r.rows.cpu()))
        # with and without the second half of the launches on a stream of its own (beside the next batch's first half)
        for halves in (True, False):
            pipe = detect_pipeline(model, 0.5, 0.5, overlap_halves=halves)
            assert (pipe.back_streams is not None) == halves
            plan = model._plan(batches[0], (0, 0))
            assert 0 < plan.split_index < len(plan.launches)
            got = [pipe.submit(x) for x in batches]
            pipe.synchronize()
            for (cnt, idx, rows), r in zip(want, got):
                assert torch.equal(cnt, r.count.cpu())
                assert int(cnt.sum()) > 0
                for b, n in enumerate(cnt.tolist()):
                    assert torch.equal(idx[b, :n], r.idx.cpu()[b, :n])
                    assert torch.equal(rows[b, :n], r.rows.cpu()[b, :n])
    finally:
        engine.set_compute_dtype(None)


@pytest.mark.parametrize('h,w,size', [(480, 640, 128), (100, 37, 64), (64, 64, 64), (20, 30, 64), (641, 1283, 640)])
def test_device_preprocess_matches_restated_cv2_resize(h, w, size):
    """odamd_preprocess_u8 (resize + /255 + HWC->CHW) equals the numpy restatement of OpenCV's 8-bit
    bilinear resize exactly, and that restatement stays within one grey level of exact bilinear
    interpolation (cv2 itself is not installed here: parity with it is unpinned)."""
    import ctypes as C
    import torch
    from objectdetection_amd._lib import lib, check
    from oracle import postprocess as opost
    rng = np.random.default_rng(h * 7 + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    want = opost.resize_u8_linear(img, size)
    src = torch.from_numpy(img).cuda()
    out = torch.empty((1, 3, size, size), dtype=torch.float32, device='cuda')
    check(lib.odamd_preprocess_u8(src.data_ptr(), h, w, out.data_ptr(), size, None))
    torch.cuda.synchronize()
    got = out.cpu().numpy()[0].transpose(1, 2, 0)
    np.testing.assert_array_equal(got, want.astype(np.float32) / np.float32(255))
    # exact bilinear (float64, half-pixel centres, clamped)
    fy = np.clip((np.arange(size) + 0.5) * h / size - 0.5, 0, h - 1)
    fx = np.clip((np.arange(size) + 0.5) * w / size - 0.5, 0, w - 1)
    y0, x0 = np.floor(fy).astype(int), np.floor(fx).astype(int)
    y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
    wy, wx = (fy - y0)[:, None, None], (fx - x0)[None, :, None]
    im = img.astype(np.float64)
    ref = (im[y0][:, x0] * (1 - wy) * (1 - wx) + im[y0][:, x1] * (1 - wy) * wx
           + im[y1][:, x0] * wy * (1 - wx) + im[y1][:, x1] * wy * wx)
    assert np.abs(want.astype(np.float64) - ref).max() <= 1.0


def test_batched_preprocess_and_host_staging():
    """odamd_preprocess_u8_batch == odamd_preprocess_u8 per image, bit for bit; PostResult.to_host (one gather, page-locked
    staging) hands back exactly the rows and indices of the device result, as views or as copies."""
    import torch
    from objectdetection_amd import postprocess as post
    from objectdetection_amd._lib import lib, check
    rng = np.random.default_rng(3)
    b, h, w, size = 5, 70, 90, 64
    imgs = torch.from_numpy(rng.integers(0, 256, (b, h, w, 3), dtype=np.uint8)).cuda()
    one = torch.empty((b, 3, size, size), dtype=torch.float32, device='cuda')
    for i in range(b):
        check(lib.odamd_preprocess_u8(imgs[i].data_ptr(), h, w, one[i].data_ptr(), size, None))
    allb = torch.empty_like(one)
    check(lib.odamd_preprocess_u8_batch(imgs.data_ptr(), b, h, w, allb.data_ptr(), size, None))
    torch.cuda.synchronize()
    assert torch.equal(one, allb)
    # synthetic deployed predictions: image 2 keeps nothing
    r, k = 500, 9
    pred = torch.from_numpy(rng.random((b, r, k), dtype=np.float32))
    pred[..., 0:2] = pred[..., 0:2] * 300 + 50
    pred[..., 2:4] = pred[..., 2:4] * 40 + 5
    pred[2, :, 4] = 0.0
    res = post.postprocess(pred.cuda(), 0.5, 0.5)
    cnt = res.count.cpu().tolist()
    assert cnt[2] == 0 and sum(cnt) > 0
    stager = post.HostStager(2)
    for copy in (False, True):
        host = res.to_host(stager, copy=copy)
        for i, (rows, idx) in enumerate(host):
            assert (rows is None) == (cnt[i] == 0) and len(idx) == cnt[i]
            if cnt[i]:
                assert np.array_equal(rows, res.rows[i, :cnt[i]].cpu().numpy())
                assert np.array_equal(idx, res.idx[i, :cnt[i]].cpu().numpy())
    legacy = res.per_image()
    assert all(np.array_equal(a[1], b_[1]) for a, b_ in zip(legacy, host)) and legacy[0][1].dtype == np.int64


def test_checkpoint_round_trip_train_to_predict(tmp_path):
    """run.py-style training writes a checkpoint; predict_class loads it (pickled module, as
    predict.py:29-30 does) and predicts on an image of another size (device-side resize)."""
    import argparse
    import torch
    from oracle import patterns
    from objectdetection_amd.train_class import train_class
    from objectdetection_amd.predict import predict_class
    path = str(tmp_path / 'last.pt')
    a = patterns.default_args(input_size=128, output_class=3)
    a.__dict__.update(device='cuda', batch=4, epoch=1, lr_start=0.001, lr_end_ratio=0.1, lr_end_epoch=10,
                      warmup_ratio=0.01, regularization='L2', r_value=0.0005, ema=True, distributed=False, local_rank=0,
                      weight_path='', weight_again=True, print_info=False, synthetic_images=8, save_path=path, save_epoch=1)
    torch.manual_seed(2)
    train_class(a).train()
    pa = argparse.Namespace(device='cuda', float16=True, input_size=128, confidence_threshold=0.3, iou_threshold=0.5)
    pc = predict_class(path, pa)
    img = np.random.default_rng(4).integers(0, 256, (300, 400, 3), dtype=np.uint8)
    out = pc.predict(img)
    assert out is None or (out.ndim == 2 and out.shape[1] == 8 and np.isfinite(out).all())
    again = pc.predict(img)
    assert (out is None) == (again is None) and (out is None or np.array_equal(out, again))


def test_reference_checkpoint_predicts_on_the_kernels(golden, tmp_path):
    """predict_class(model_path) on a checkpoint file the REFERENCE wrote (pickled model.yolov7.yolov7 with a
    pruned width list): the module comes back as this build's class and its eval output equals the
    reference's own (tests/golden/ref_ckpt.npz); the weights-only form written next to a checkpoint loads to the
    same result."""
    import os
    from objectdetection_amd import engine
    from objectdetection_amd.model.layer import deploy
    from objectdetection_amd.model.yolov7 import yolov7
    from objectdetection_amd.predict import predict_class
    from objectdetection_amd.train_class import train_class
    g = golden('ref_ckpt.npz')
    path = os.path.join(os.path.dirname(__file__), 'golden', 'ref_ckpt.pt')
    a = argparse.Namespace(device='cuda', float16=False, input_size=64, confidence_threshold=0.3, iou_threshold=0.5)
    pc = predict_class(path, args=a)
    assert type(pc.model.model) is yolov7
    x = patterns.image_batch(2, 64).cuda()
    engine.set_compute_dtype('f32')
    try:
        got = pc.model(x).cpu().numpy()
    finally:
        engine.set_compute_dtype(None)
    ref = g['deploy']
    for sl in (slice(0, 4), slice(4, None)):
        np.testing.assert_allclose(got[..., sl], ref[..., sl], rtol=1e-4, atol=1e-4 * np.abs(ref[..., sl]).max())
    # the same model through the weights-only form
    out = str(tmp_path / 'again.pt')
    train_class.save_checkpoint({'model': pc.model.model, 'epoch_finished': 1}, out)
    pc2 = predict_class(os.path.splitext(out)[0] + '.weights.pt', args=a)
    engine.set_compute_dtype('f32')
    try:
        again = pc2.model(x).cpu().numpy()
    finally:
        engine.set_compute_dtype(None)
    assert np.array_equal(again, got)
    img = (patterns.image_batch(1, 64)[0].permute(1, 2, 0).numpy() * 255).astype(np.uint8)
    r1, r2 = pc.predict(img), pc2.predict(img)
    assert (r1 is None and r2 is None) or np.array_equal(r1, r2)
