"""The CPU oracle must reproduce the fixtures recorded from the reference."""
import argparse

import numpy as np
import torch

from oracle import network, patterns, postprocess, training


def _sd_for(tag, g, extra_prefix=''):
    """Re-create the pattern-filled state_dict of a recorded block from fixture key names."""
    sd = {}
    for key in g.files:
        if key.startswith(tag + '.grad.'):
            n = key[len(tag) + 6:]
            sd[n] = patterns.wave(n, g[key].shape, 1.0)  # placeholder, refilled below
        elif key.startswith(tag + '.buf.'):
            n = key[len(tag) + 5:]
            sd[n] = torch.zeros(g[key].shape)
    m = _Holder(sd)
    patterns.fill_state_dict(m)
    return {k: v.clone() for k, v in m.state_dict().items()}


class _Holder(torch.nn.Module):
    """Minimal module whose state_dict has exactly the given names (dots kept)."""

    def __init__(self, sd):
        super().__init__()
        self._names = list(sd)
        self._t = {k: v.clone() for k, v in sd.items()}

    def state_dict(self):
        return self._t


def _run_block(tag, sd, x, training):
    net = network.Net({('.' + k): v for k, v in sd.items()}, training)
    if tag.startswith('cbs'):
        return net.cbs('', x, 2 if tag.endswith('s2') else 1)
    return getattr(net, tag.split('_n')[0] if tag.startswith('elan_n') else tag)('', x)


INPUTS = {
    'cbs_k1': ('x16', (2, 16, 12, 12), 1.0, 0.0), 'cbs_k3': ('x16', (2, 16, 12, 12), 1.0, 0.0),
    'cbs_k3s2': ('x16', (2, 16, 12, 12), 1.0, 0.0), 'cbs_in3': ('x3', (2, 3, 16, 16), 0.5, 0.5),
    'elan': ('x16', (2, 16, 12, 12), 1.0, 0.0), 'elan_n2': ('x16', (2, 16, 12, 12), 1.0, 0.0),
    'elan_h': ('x32', (2, 32, 8, 8), 1.0, 0.0), 'mp': ('x16', (2, 16, 12, 12), 1.0, 0.0),
    'sppcspc': ('xs', (2, 32, 20, 20), 1.0, 0.0),
}


def test_blocks_match_reference(golden):
    g = golden('blocks.npz')
    for tag, (xn, shape, sc, off) in INPUTS.items():
        x = patterns.wave(xn, shape, sc, off)
        sd = _sd_for(tag, g)
        y = _run_block(tag, {k: v.clone() for k, v in sd.items()}, x, False)
        np.testing.assert_allclose(y.numpy(), g[f'{tag}.eval'], rtol=1e-5, atol=1e-6, err_msg=tag)
        # train mode: output, input grad, weight grads, running stats
        sdt = {k: (v.clone().requires_grad_(True) if k.endswith(('weight', 'bias')) else v.clone())
               for k, v in sd.items()}
        xin = x.clone().requires_grad_(True)
        y = _run_block(tag, sdt, xin, True)
        (y * patterns.wave(tag + '.gy', y.shape, 1.0)).sum().backward()
        np.testing.assert_allclose(y.detach().numpy(), g[f'{tag}.train'], rtol=1e-5, atol=1e-6, err_msg=tag)
        np.testing.assert_allclose(xin.grad.numpy(), g[f'{tag}.dx'], rtol=1e-4, atol=1e-5, err_msg=tag)
        for k, v in sdt.items():
            if v.requires_grad:
                ref = g[f'{tag}.grad.{k}']   # sums over all pixels: float32 summation order differs
                np.testing.assert_allclose(v.grad.numpy(), ref, rtol=1e-4, atol=2e-4 * np.abs(ref).max(),
                                           err_msg=f'{tag} {k}')
            elif v.dtype.is_floating_point:
                np.testing.assert_allclose(v.numpy(), g[f'{tag}.buf.{k}'], rtol=1e-5, atol=1e-6,
                                           err_msg=f'{tag} {k}')


def test_head_matches_reference(golden):
    g = golden('blocks.npz')
    m = _Holder({'output.weight': torch.zeros(24, 16, 1, 1), 'output.bias': torch.zeros(24)})
    patterns.fill_state_dict(m)
    net = network.Net({('.' + k): v for k, v in m.state_dict().items()})
    y = net.head('', patterns.wave('x16', (2, 16, 12, 12), 1.0), 3, 8)
    np.testing.assert_allclose(y.numpy(), g['head.eval'], rtol=1e-5, atol=1e-6)


def _net_sd(args):
    """Pattern-filled state dict with the build's own name/shape table (no reference needed)."""
    from objectdetection_amd.model.yolov7 import yolov7
    m = yolov7(args)
    patterns.fill_state_dict(m)
    return {k: v.clone() for k, v in m.state_dict().items()}


def test_whole_net_matches_reference(golden):
    g = golden('net.npz')
    for tag, size, ncls, batch in (('s64', 64, 3, 2), ('s96', 96, 1, 1)):
        args = patterns.default_args(input_size=size, output_class=ncls)
        sd = _net_sd(args)
        x = patterns.image_batch(batch, size)
        with torch.no_grad():
            y = network.yolov7_forward(sd, x, args)
            yd = network.yolov7_forward(sd, x, args, deploy=True)
        np.testing.assert_allclose(y.numpy(), g[f'{tag}.eval'], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(yd.numpy(), g[f'{tag}.deploy'], rtol=2e-5, atol=2e-5)


def test_train_step_matches_reference(golden):
    g = golden('net.npz')
    tag, size, ncls, batch = 's64', 64, 3, 2
    args = patterns.default_args(input_size=size, output_class=ncls)
    sd = {k: (v.clone().requires_grad_(True) if k.endswith(('weight', 'bias')) else v.clone())
          for k, v in _net_sd(args).items()}
    x = patterns.image_batch(batch, size)
    pred = network.yolov7_forward(sd, x, args, training=True)
    np.testing.assert_allclose(pred.detach().numpy(), g[f'{tag}.train'], rtol=1e-4, atol=1e-4)
    screen = [torch.from_numpy(g[f'{tag}.screen{b}']) for b in range(batch)]
    lab = torch.from_numpy(g[f'{tag}.label'])
    vals = training.detection_loss(pred, screen, lab, 0.01, ncls)
    np.testing.assert_allclose([v.item() for v in vals], g[f'{tag}.loss'], rtol=1e-5)
    vals[0].backward()
    for key in g.files:
        if key.startswith(f'{tag}.grad.'):
            n = key[len(tag) + 6:]
            ref = g[key]
            np.testing.assert_allclose(sd[n].grad.numpy(), ref, rtol=2e-3, atol=2e-4 * np.abs(ref).max(), err_msg=n)
        elif key.startswith(f'{tag}.buf.'):
            n = key[len(tag) + 5:]
            np.testing.assert_allclose(sd[n].numpy(), g[key], rtol=1e-5, atol=1e-6, err_msg=n)


def test_decode_matches_reference(golden):
    g = golden('decode.npz')
    for tag, size, ncls, batch in (('d64', 64, 3, 2), ('d160', 160, 80, 1)):
        args = patterns.default_args(input_size=size, output_class=ncls)
        heads = [patterns.wave(f'{tag}.h{i}', (batch, 3, h, h, 5 + ncls), 3.0) for i, h in enumerate(args.output_size)]
        y = network.decode(heads, size, args.anchor)
        np.testing.assert_allclose(y.numpy(), g[f'{tag}.decoded'], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(network.deploy_sigmoid(y).numpy(), g[f'{tag}.deployed'], rtol=1e-6, atol=1e-6)


def test_postprocess_kept_indices_bit_exact(golden):
    g = golden('nms.npz')
    for tag in ('spread', 'dense', 'dense_lo', 'edge', 'empty', 'single'):
        size, cthr, ithr, h, w = g[f'{tag}.cfg']
        rows, kept = postprocess.predict_decode(g[f'{tag}.pred'], (int(h), int(w), 3), int(size), cthr, ithr)
        assert np.array_equal(kept, g[f'{tag}.kept']), tag
        if len(kept) == 0:
            assert rows is None
        else:
            assert np.array_equal(rows, g[f'{tag}.rows']), tag
    a = g['iou.a']
    got = postprocess.iou_xywh(a[0], a)
    assert np.array_equal(got, g['iou.val'], equal_nan=True)


def test_loss_matches_reference(golden):
    g = golden('training.npz')
    for tag, ncls in (('c1', 1), ('c4', 4)):
        b, r, k = 3, 60, 5 + ncls
        pred = patterns.wave(f'loss.{tag}.pred', (b, r, k), 2.0)
        pred[:, :, 0:2] = pred[:, :, 0:2] * 20 + 60
        pred[:, :, 2:4] = pred[:, :, 2:4].abs() * 15 + 5
        screen = [torch.tensor([1, 5, 17]), torch.tensor([], dtype=torch.int64), torch.tensor([0, 2, 30, 59])]
        lab = patterns.wave(f'loss.{tag}.lab', (7, k), 0.45, 0.5)
        lab[:, 0:2] = lab[:, 0:2] * 40 + 40
        lab[:, 2:4] = lab[:, 2:4] * 30 + 6
        p = pred.clone().requires_grad_(True)
        vals = training.detection_loss(p, screen, lab, 0.01, ncls)
        vals[0].backward()
        np.testing.assert_allclose([v.item() for v in vals], g[f'{tag}.loss'], rtol=1e-6)
        np.testing.assert_allclose(p.grad.numpy(), g[f'{tag}.dpred'], rtol=1e-5, atol=1e-7)
        p2 = pred.clone().requires_grad_(True)
        vals = training.detection_loss(p2, [torch.tensor([], dtype=torch.int64)] * b, lab[:0], 0.01, ncls)
        vals[0].backward()
        np.testing.assert_allclose([float(v) for v in vals], g[f'{tag}.loss_nopos'], rtol=1e-6)
        np.testing.assert_allclose(p2.grad.numpy(), g[f'{tag}.dpred_nopos'], rtol=1e-5, atol=1e-7)


def test_lr_schedule_matches_reference(golden):
    g = golden('training.npz')
    for tag, (step_epoch, lr_end_epoch, warm) in (('a', (10, 100, 0.01)), ('b', (400, 3, 0.05))):
        s = training.LrSchedule(0.001, 0.1, lr_end_epoch, warm, step_epoch)
        seq = [s.step() for _ in range(30)]
        assert np.array_equal(np.array(seq), g[f'lr.{tag}'])


def test_adam_ema_match_reference(golden):
    g = golden('training.npz')
    shapes = [(7, 5), (33,), (4, 3, 3, 3)]
    ps = [patterns.wave(f'adam.p{i}', s, 0.5) for i, s in enumerate(shapes)]
    ms = [torch.zeros(s) for s in shapes]
    vs = [torch.zeros(s) for s in shapes]
    ema = {f'p{i}': p.clone() for i, p in enumerate(ps)}
    ema['running'] = torch.zeros(5)
    running = torch.zeros(5)
    for step in range(3):
        for i in range(3):
            training.adam_step(ps[i], patterns.wave(f'adam.g{i}.{step}', shapes[i], 0.3), ms[i], vs[i], step + 1,
                               0.001 * (step + 1))
        running += 0.5
        training.ema_update(ema, {**{f'p{i}': p for i, p in enumerate(ps)}, 'running': running}, step + 1)
    for i in range(3):
        np.testing.assert_allclose(ps[i].numpy(), g[f'adam.p{i}'], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(ms[i].numpy(), g[f'adam.m{i}'], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(vs[i].numpy(), g[f'adam.v{i}'], rtol=1e-6, atol=1e-12)
        np.testing.assert_allclose(ema[f'p{i}'].numpy(), g[f'ema.p{i}'], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(ema['running'].numpy(), g['ema.running'], rtol=1e-6)
