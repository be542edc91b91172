"""GPU parity of decode / filter / sort / NMS (libodamd.so through the C ABI) vs the oracle."""
import numpy as np
import pytest
import torch

from oracle import network, patterns, postprocess as opost

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def pp():
    from objectdetection_amd import postprocess
    return postprocess


def _synthetic_pred(seed, n, ncls, size, dense=False):
    rng = np.random.default_rng(seed)
    p = np.zeros((n, 5 + ncls), np.float32)
    if dense:
        c = rng.uniform(size * 0.1, size * 0.9, (max(n // 150, 4), 2))
        p[:, 0:2] = c[rng.integers(0, len(c), n)] + rng.normal(0, size / 80, (n, 2))
        p[:, 2:4] = rng.uniform(size / 20, size / 8, (n, 2))
    else:
        p[:, 0:2] = rng.uniform(0, size, (n, 2))
        p[:, 2:4] = rng.uniform(4, size / 5, (n, 2))
    p[:, 4] = rng.uniform(0, 1, n)
    p[:, 5:] = rng.uniform(0, 1, (n, ncls))
    return p


def test_decode_matches_oracle_and_golden(pp, golden):
    g = golden('decode.npz')
    for tag, size, ncls, batch in (('d64', 64, 3, 2), ('d160', 160, 80, 1)):
        args = patterns.default_args(input_size=size, output_class=ncls)
        heads = [patterns.wave(f'{tag}.h{i}', (batch, 3, h, h, 5 + ncls), 3.0) for i, h in enumerate(args.output_size)]
        dev = [h.cuda() for h in heads]
        for deploy, key in ((False, 'decoded'), (True, 'deployed')):
            got = pp.decode(dev, size, args.anchor, deploy).cpu().numpy()
            ref = network.decode(heads, size, args.anchor)
            ref = network.deploy_sigmoid(ref) if deploy else ref
            # decoded boxes within 1e-5 relative (float32 exp differs in the last bit between CPU and GPU)
            np.testing.assert_allclose(got, ref.numpy(), rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(got, g[f'{tag}.{key}'], rtol=1e-5, atol=1e-6)


def test_decode_backward_matches_autograd(pp):
    size, ncls, batch = 64, 3, 2
    args = patterns.default_args(input_size=size, output_class=ncls)
    heads = [patterns.wave(f'db.h{i}', (batch, 3, h, h, 5 + ncls), 2.0).requires_grad_(True)
             for i, h in enumerate(args.output_size)]
    out = network.decode(heads, size, args.anchor)
    gout = patterns.wave('db.g', out.shape, 1.0)
    out.backward(gout)
    got = pp.decode_backward([h.detach().cuda() for h in heads], size, args.anchor, gout.cuda())
    for h, gg in zip(heads, got):
        np.testing.assert_allclose(gg.cpu().numpy(), h.grad.numpy(), rtol=1e-4, atol=1e-6)


def test_postprocess_golden_bit_exact(pp, golden):
    g = golden('nms.npz')
    for tag in ('spread', 'dense', 'dense_lo', 'edge', 'empty', 'single'):
        size, cthr, ithr, h, w = g[f'{tag}.cfg']
        pred = torch.from_numpy(g[f'{tag}.pred']).cuda()
        scale = torch.tensor([[np.float32(w / size), np.float32(h / size)]], dtype=torch.float32).cuda()
        rows, kept = pp.postprocess(pred, cthr, ithr, scale).per_image()[0]
        assert np.array_equal(kept, g[f'{tag}.kept']), tag
        if len(kept) == 0:
            assert rows is None
        else:
            assert np.array_equal(rows, g[f'{tag}.rows']), tag


@pytest.mark.parametrize('n,ncls,dense,cthr,ithr', [(5000, 80, False, 0.25, 0.5), (6000, 1, True, 0.1, 0.45),
                                                    (4097, 5, True, 0.001, 0.6), (63, 2, False, 0.0, 0.5)])
def test_postprocess_matches_oracle_bit_exact(pp, n, ncls, dense, cthr, ithr):
    batch = 3
    preds = np.stack([_synthetic_pred(100 + b, n, ncls, 640, dense) for b in range(batch)])
    res = pp.postprocess(torch.from_numpy(preds).cuda(), cthr, ithr).per_image()
    for b in range(batch):
        rows, kept = opost.predict_decode(preds[b], (640, 640, 3), 640, cthr, ithr)
        assert np.array_equal(res[b][1], kept), f'image {b}'
        assert np.array_equal(res[b][0], rows)


def test_candidate_counts_far_below_the_padded_row_count(pp):
    """Images of one batch whose candidates fill 1 %, 25 %, 50 % and nearly all of the 20 000 rows (padded to
    32 768 sort keys): the sort works on the power-of-two prefix that holds an image's candidates (one 4 096-key
    chunk, 8 192, 16 384 and all 32 768 keys here), and the order has to be the full sort's."""
    n, ncls = 20000, 3
    preds = np.stack([_synthetic_pred(300 + b, n, ncls, 640) for b in range(4)])
    for b, keep_frac in enumerate((0.012, 0.25, 0.5, 0.97)):
        # confidence * best class score >= 0.25 for about keep_frac of the rows
        preds[b, :, 5:] = 1.0
        preds[b, :, 4] = np.where(np.random.default_rng(b).uniform(0, 1, n) < keep_frac, 0.3 + 0.7 * preds[b, :, 4], 0.2 * preds[b, :, 4])
    res = pp.postprocess(torch.from_numpy(preds).cuda(), 0.25, 0.5)
    cand = res.cand.cpu().numpy()
    assert cand[0] <= 4096 < cand[1] <= 8192 < cand[2] <= 16384 < cand[3], cand
    out = res.per_image()
    for b in range(4):
        rows, kept = opost.predict_decode(preds[b], (640, 640, 3), 640, 0.25, 0.5)
        assert np.array_equal(out[b][1], kept), f'image {b}'
        assert np.array_equal(out[b][0], rows)


@pytest.mark.parametrize('dense,cthr,ithr', [(True, 0.0, 0.5), (False, 0.0, 0.3)])
def test_greedy_pass_over_several_compacted_chunks(pp, dense, cthr, ithr):
    """64 images of 17 000 candidates each (32 768 sort keys; a batch of 64 gets mask chunks of 4 096 rows): the greedy
    pass runs five times over a candidate list that is compacted in between (decided rows and suppressed candidates
    dropped, the rest moved up).  Three distinct images are repeated through the batch; kept indices and rows of every
    copy equal the oracle's, and the candidate count reported is the one before compaction."""
    n, ncls, batch = 17000, 1, 64
    distinct = [_synthetic_pred(700 + b, n, ncls, 640, dense) for b in range(3)]
    want = [opost.predict_decode(p, (640, 640, 3), 640, cthr, ithr) for p in distinct]
    preds = np.stack([distinct[b % 3] for b in range(batch)])
    res = pp.postprocess(torch.from_numpy(preds).cuda(), cthr, ithr)
    cand = res.cand.cpu().numpy()
    out = res.per_image()
    for b in range(batch):
        rows, kept = want[b % 3]
        assert cand[b] == len(opost.filter_indices(distinct[b % 3], cthr)[1])
        assert np.array_equal(out[b][1], kept), f'image {b}'
        assert np.array_equal(out[b][0], rows)
    assert 0 < len(want[0][1]) < n


@pytest.mark.parametrize('ithr', [0.3, 0.5, 0.7, 0.02, 0.98, 0.0, 1.0, 1.5])
def test_nms_mixed_scales_and_irregular_boxes(pp, ithr):
    """Boxes from 2 to 400 pixels (log-uniform, all shapes) in overlapping clusters, with zero, negative,
    huge, infinite and NaN extents sprinkled in: the size windows of the mask kernel must drop only pairs
    that cannot reach the threshold, and irregular boxes keep the reference arithmetic (a NaN IoU
    suppresses, predict.py:90).  Kept indices against the restated loop, bit for bit."""
    rng = np.random.default_rng(77)
    n, ncls = 3500, 3
    rows = np.zeros((2, n, 5 + ncls), np.float32)
    for b in range(2):
        centres = rng.uniform(40, 600, (60, 2))
        c = centres[rng.integers(0, 60, n)] + rng.normal(0, 6, (n, 2))
        wh = np.exp(rng.uniform(np.log(2.0), np.log(400.0), (n, 2)))
        # families of near-identical boxes, so that suppression actually happens at every scale
        fam = rng.integers(0, 400, n)
        wh = wh[fam] * rng.uniform(0.9, 1.1, (n, 2))
        c = c[fam] + rng.normal(0, 1.0, (n, 2)) * wh * 0.05
        rows[b, :, 0:2] = c - wh / 2
        rows[b, :, 2:4] = wh
        rows[b, :, 4] = rng.uniform(0.05, 1, n)
        rows[b, :, 5:] = rng.uniform(0.05, 1, (n, ncls))
        bad = rng.choice(n, 40, replace=False)
        rows[b, bad[0:8], 2] = 0.0
        rows[b, bad[8:16], 3] = -5.0
        rows[b, bad[16:22], 2:4] = 0.0
        rows[b, bad[22:26], 2] = np.inf
        rows[b, bad[26:30], 0] = np.nan
        rows[b, bad[30:34], 2:4] = 1.0e20
        rows[b, bad[34:40], 2:4] = 1.0e-25
        # NaN scores of either sign (np.argsort puts all of them last, so the reversed order starts with them)
        nan_neg = np.array([0xFFC00000], np.uint32).view(np.float32)[0]
        odd = rng.choice(n, 6, replace=False)
        rows[b, odd[0:2], 4] = np.nan
        rows[b, odd[2:4], 4] = nan_neg
        rows[b, odd[4:6], 6] = np.nan
    with np.errstate(all='ignore'):
        res = pp.nms_rows(torch.from_numpy(rows).cuda(), ithr).per_image()
        for b in range(2):
            want = opost.nms_indices(rows[b], ithr)
            if 0.1 < ithr < 0.9:
                assert 200 < len(want) < n - 200
            assert np.array_equal(res[b][1], want), f'image {b}'


def test_postprocess_heads_equals_materialised(pp):
    size, ncls, batch = 160, 7, 4
    args = patterns.default_args(input_size=size, output_class=ncls)
    heads = [patterns.wave(f'ph.h{i}', (batch, 3, h, h, 5 + ncls), 2.5).cuda() for i, h in enumerate(args.output_size)]
    full = pp.decode(heads, size, args.anchor, True)
    a = pp.postprocess(full, 0.3, 0.5).per_image()
    b = pp.postprocess_heads(heads, size, args.anchor, 0.3, 0.5).per_image()
    for (ra, ka), (rb, kb) in zip(a, b):
        assert len(ka) > 0
        assert np.array_equal(ka, kb)
        assert np.array_equal(ra, rb)


@pytest.mark.parametrize('r,size', [(25200, 640), (100800, 1280)])
def test_full_size_properties(pp, r, size):
    """25 200 rows/image (640^2 grid) and 100 800 rows/image (1280^2, the dense-candidate stress
    configuration: the suppression mask is processed in row chunks), conf 0.001: properties that pin
    greedy NMS at any size."""
    ncls, thr = 80, 0.5
    pred = _synthetic_pred(7, r, ncls, size, dense=False)
    pred[:, 2:4] *= 0.25
    res = pp.postprocess(torch.from_numpy(pred[None]).cuda(), 0.001, thr)
    rows, kept = res.per_image()[0]
    cand, idx = opost.filter_indices(pred, 0.001)
    assert int(res.cand.cpu()[0]) == len(idx)
    order = opost.score_order(cand)
    pos = {int(idx[o]): i for i, o in enumerate(order)}
    # kept is a subsequence of the score order
    ranks = np.array([pos[int(k)] for k in kept])
    assert np.all(np.diff(ranks) > 0)
    boxes = {int(i): cand[j, 0:4] for j, i in enumerate(idx)}
    kb = np.stack([boxes[int(k)] for k in kept])
    # (1) kept boxes do not suppress each other; (2) every dropped candidate is suppressed by an earlier kept box
    sample = np.random.default_rng(0).choice(len(kept), 200, replace=False)
    for s in sample:
        iou = opost.iou_xywh(kb[s], kb[:s])
        assert np.all(iou < np.float32(thr))
    kept_set = set(int(k) for k in kept)
    dropped = [int(idx[o]) for o in order if int(idx[o]) not in kept_set]
    for d in np.random.default_rng(1).choice(dropped, 200, replace=False):
        earlier = kb[ranks < pos[int(d)]]
        iou = opost.iou_xywh(boxes[int(d)], earlier)
        assert np.any(~(iou < np.float32(thr)))
    # idempotence: running the kept rows through NMS again keeps all of them
    again = pp.postprocess(torch.from_numpy(pred[kept][None]).cuda(), 0.001, thr).per_image()[0][1]
    assert np.array_equal(np.sort(again), np.arange(len(kept)))


def test_iou_mask_matches_numpy(pp):
    import ctypes as C
    from objectdetection_amd._lib import lib, check
    n = 256
    boxes = _synthetic_pred(3, n, 1, 200, dense=True)[:, 0:4].copy()
    boxes[5, 2:4] = 0
    boxes[9] = boxes[5]
    dev = torch.from_numpy(boxes).cuda()
    mask = torch.zeros((n, n // 64), dtype=torch.int64, device='cuda')
    check(lib.odamd_iou_mask(C.c_void_p(dev.data_ptr()), n, 0.4, C.c_void_p(mask.data_ptr()), C.c_void_p(0)))
    torch.cuda.synchronize()
    got = mask.cpu().numpy().view(np.uint64)
    for i in range(n):
        iou = opost.iou_xywh(boxes[i], boxes)
        sup = ~(iou < np.float32(0.4))
        for j in range(i + 1, n):
            assert bool((got[i, j // 64] >> np.uint64(j % 64)) & np.uint64(1)) == bool(sup[j]), (i, j)
