"""The dense-candidate configuration as a pipeline: 1280 x 1280 input, 80 classes, confidence 0.001,
random-init weights under seed 999 (the bench's initialisation).  The network (f32 and bf16) against the CPU
oracle, then the fused decode / filter / sort / NMS of the raw heads against the numpy restatement of
predict.py:67-92 run on the GPU's OWN deployed tensor: kept indices and rows bit for bit."""
import numpy as np
import pytest
import torch

from oracle import network, patterns, postprocess as opost

pytestmark = pytest.mark.gpu


def rel_err(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12)


@pytest.fixture(scope='module')
def dense():
    import objectdetection_amd.engine as engine
    from objectdetection_amd.model.yolov7 import yolov7
    args = patterns.default_args(input_size=1280, output_class=80)
    assert args.output_size == (160, 80, 40)
    torch.manual_seed(999)
    net = yolov7(args)
    x = torch.rand(2, 3, 1280, 1280, generator=torch.Generator().manual_seed(999))
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        ref = network.yolov7_forward(sd, x, args).numpy()
    return engine, args, net.cuda().eval(), x, ref


@pytest.mark.parametrize('dtype,tol', [('f32', 1e-4), ('bf16', 2e-2)])
def test_dense_network_vs_oracle(dense, dtype, tol):
    engine, args, net, x, ref = dense
    engine.set_compute_dtype(dtype)
    try:
        got = net(x.cuda()).cpu().numpy()
    finally:
        engine.set_compute_dtype(None)
    assert got.shape == (2, 100800, 85)
    for sl in (slice(0, 4), slice(4, None)):
        err = rel_err(got[..., sl], ref[..., sl])
        assert err < tol, (dtype, sl, err)
        np.testing.assert_allclose(got[..., sl], ref[..., sl], rtol=tol, atol=tol * float(np.abs(ref[..., sl]).max()))


def test_dense_postprocess_of_real_heads_bit_exact(dense):
    """conf 0.001 / iou 0.5 on the network's own output: ~94 k candidates and ~22 k kept boxes per image."""
    from objectdetection_amd import postprocess
    from objectdetection_amd.model.layer import deploy
    engine, args, net, x, _ = dense
    engine.set_compute_dtype('bf16')
    try:
        xg = x.cuda()
        heads = net.head_tensors(xg)
        res = postprocess.postprocess_heads(heads, 1280, args.anchor, 0.001, 0.5)
        cand = res.cand.cpu().numpy()
        per = res.per_image()
        deployed = deploy(net)(xg).cpu().numpy()       # the materialised (B,R,K) tensor of the same forward
    finally:
        engine.set_compute_dtype(None)
    assert cand.min() > 60000                          # the dense regime
    rows, kept = opost.predict_decode(deployed[0], (1280, 1280, 3), 1280, 0.001, 0.5)
    assert len(kept) > 10000
    assert np.array_equal(per[0][1], kept)
    assert np.array_equal(per[0][0].view(np.uint32), rows.view(np.uint32))
    # second image: candidate filter and count (the greedy loop itself is checked on image 0)
    _, idx = opost.filter_indices(deployed[1], 0.001)
    assert cand[1] == len(idx)
