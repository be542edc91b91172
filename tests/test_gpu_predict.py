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
    """Two-stream pipeline (alternating plans) returns exactly what the one-stream path returns."""
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
        pipe = detect_pipeline(model, 0.5, 0.5)
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
