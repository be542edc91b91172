"""Generate the golden fixtures in this directory by running the REFERENCE.

Run in the build container only (needs /root/reference on disk):

    python tests/golden/make_golden.py

The reference's predict.py / train_class.py import cv2, wandb and torchvision
for image I/O, logging and validation only; none of the arithmetic captured here
touches them, so empty stand-in modules are registered for those three names
before the import.  Weights and inputs come from oracle.patterns (closed form),
so tests regenerate identical inputs on the GPU box without the reference.
The fixtures are data only: inputs and the reference's outputs.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
# The reference's `model` directory has no __init__.py (a namespace package), and this repository ships a
# regular package of the same name (the import-path alias of the MI355X classes) that would win over it:
# import the reference first, with only its own tree on the path.
sys.path.insert(0, REF)
for name in ('cv2', 'wandb', 'torchvision'):
    sys.modules.setdefault(name, types.ModuleType(name))
_ARGV = list(sys.argv)
sys.argv = ['make_golden']

import model.layer as ref_layer  # noqa: E402
from model.yolov7 import yolov7 as ref_yolov7  # noqa: E402
import predict as ref_predict  # noqa: E402
import train_class as ref_train  # noqa: E402

assert ref_layer.__file__.startswith(REF), ref_layer.__file__
sys.path.insert(1, ROOT)
from oracle import patterns  # noqa: E402

torch.set_num_threads(4)
f32 = np.float32


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB')


# --------------------------------------------------------------------------- blocks
def block_case(tag, module, x, out):
    """Forward (eval + train) and backward through one reference block."""
    patterns.fill_state_dict(module)
    module.eval()
    with torch.no_grad():
        out[f'{tag}.eval'] = module(x.clone()).numpy()
    module.train()
    xin = x.clone().requires_grad_(True)
    y = module(xin)
    gy = patterns.wave(tag + '.gy', y.shape, 1.0)
    (y * gy).sum().backward()
    out[f'{tag}.train'] = y.detach().numpy()
    out[f'{tag}.dx'] = xin.grad.numpy()
    for n, p in module.named_parameters():
        out[f'{tag}.grad.{n}'] = p.grad.numpy()
    for n, b in module.named_buffers():
        if b.dtype.is_floating_point:
            out[f'{tag}.buf.{n}'] = b.detach().numpy()


def gen_blocks():
    out = {}
    x16 = patterns.wave('x16', (2, 16, 12, 12), 1.0)
    x32 = patterns.wave('x32', (2, 32, 8, 8), 1.0)
    block_case('cbs_k1', ref_layer.cbs(16, 24, 1, 1), x16, out)
    block_case('cbs_k3', ref_layer.cbs(16, 16, 3, 1), x16, out)
    block_case('cbs_k3s2', ref_layer.cbs(16, 32, 3, 2), x16, out)
    block_case('cbs_in3', ref_layer.cbs(3, 16, 3, 1), patterns.wave('x3', (2, 3, 16, 16), 0.5, 0.5), out)
    block_case('elan', ref_layer.elan(16, 32, 1), x16, out)
    block_case('elan_n2', ref_layer.elan(16, 32, 2), x16, out)
    block_case('elan_h', ref_layer.elan_h(32, 16), x32, out)
    block_case('mp', ref_layer.mp(16, 32), x16, out)
    block_case('sppcspc', ref_layer.sppcspc(32, 16), patterns.wave('xs', (2, 32, 20, 20), 1.0), out)
    hd = ref_layer.head(16, 12, 3)
    patterns.fill_state_dict(hd)
    with torch.no_grad():
        out['head.eval'] = hd(x16).numpy()
    save('blocks.npz', **out)


# --------------------------------------------------------------------------- whole net
def gen_net():
    out = {}
    for tag, size, ncls, batch in (('s64', 64, 3, 2), ('s96', 96, 1, 1)):
        args = patterns.default_args(input_size=size, output_class=ncls)
        net = ref_yolov7(args)
        patterns.fill_state_dict(net)
        x = patterns.image_batch(batch, size)
        net.eval()
        with torch.no_grad():
            out[f'{tag}.eval'] = net(x.clone()).numpy()
            out[f'{tag}.deploy'] = ref_layer.deploy(net)(x.clone()).numpy()
        if tag != 's64':
            continue
        # one training step: forward (batch stats), loss, backward
        net.train()
        pred = net(x.clone())
        out[f'{tag}.train'] = pred.detach().numpy()
        r = pred.shape[1]
        screen = [torch.tensor(sorted({(7 * i + 3 * b) % r for i in range(9)}), dtype=torch.int64)
                  for b in range(batch)]
        npos = sum(len(s) for s in screen)
        lab = patterns.wave('label', (npos, 5 + ncls), 0.4, 0.5)
        lab[:, 0:2] = lab[:, 0:2] * size
        lab[:, 2:4] = lab[:, 2:4] * size * 0.5 + 4
        largs = argparse.Namespace(device='cpu', input_size=size, output_size=args.output_size,
                                   label_smooth=0.01, output_class=ncls)
        loss_fn = ref_train.loss_class(largs)
        loss, fl, cl, kl = loss_fn(pred, screen, lab)
        loss.backward()
        out[f'{tag}.loss'] = np.array([loss.item(), fl.item(), cl.item(), kl.item()], f32)
        out[f'{tag}.label'] = lab.numpy()
        for b, s in enumerate(screen):
            out[f'{tag}.screen{b}'] = s.numpy()
        for n, p in net.named_parameters():
            g = p.grad
            if n in ('cbs0.conv.weight', 'cbs3.bn.weight', 'elan4.cbs5.conv.weight', 'sppcspc11.cbs9.conv.weight',
                     'elan_h19.cbs7.bn.bias', 'head0.output.weight', 'head2.output.bias', 'mp20.cbs3.conv.weight'):
                out[f'{tag}.grad.{n}'] = g.numpy()
            out[f'{tag}.gnorm.{n}'] = np.array([g.double().norm().item(), g.double().sum().item()])
        for n, b in net.named_buffers():
            if n in ('cbs0.bn.running_mean', 'cbs0.bn.running_var', 'elan_h25.cbs7.bn.running_var'):
                out[f'{tag}.buf.{n}'] = b.numpy()
    save('net.npz', **out)


def gen_net128():
    """Training step at 128x128, batch 4 (deepest maps 4x4 -> 64 values per BN channel): a better
    conditioned whole-network check than the 64x64 case; only small tensors are stored."""
    out = {}
    tag, size, ncls, batch = 's128', 128, 5, 4
    args = patterns.default_args(input_size=size, output_class=ncls)
    net = ref_yolov7(args)
    patterns.fill_state_dict(net)
    x = patterns.image_batch(batch, size)
    net.train()
    pred = net(x.clone())
    out[f'{tag}.train_sample'] = pred.detach().numpy()[:, ::7]
    r = pred.shape[1]
    screen = [torch.tensor(sorted({(11 * i + 5 * b) % r for i in range(12)}), dtype=torch.int64) for b in range(batch)]
    npos = sum(len(s) for s in screen)
    lab = patterns.wave('label128', (npos, 5 + ncls), 0.4, 0.5)
    lab[:, 0:2] = lab[:, 0:2] * size
    lab[:, 2:4] = lab[:, 2:4] * size * 0.5 + 4
    largs = argparse.Namespace(device='cpu', input_size=size, output_size=args.output_size, label_smooth=0.01,
                               output_class=ncls)
    loss, fl, cl, kl = ref_train.loss_class(largs)(pred, screen, lab)
    loss.backward()
    out[f'{tag}.loss'] = np.array([loss.item(), fl.item(), cl.item(), kl.item()], f32)
    out[f'{tag}.label'] = lab.numpy()
    for b, s in enumerate(screen):
        out[f'{tag}.screen{b}'] = s.numpy()
    for n, p in net.named_parameters():
        g = p.grad
        if n in ('cbs0.conv.weight', 'cbs3.bn.weight', 'elan_h19.cbs7.bn.bias', 'head0.output.weight',
                 'head2.output.bias', 'mp20.cbs3.conv.weight', 'elan4.cbs1.conv.weight'):
            out[f'{tag}.grad.{n}'] = g.numpy()
        out[f'{tag}.gnorm.{n}'] = np.array([g.double().norm().item(), g.double().sum().item()])
    for n, b in net.named_buffers():
        if n in ('cbs0.bn.running_mean', 'cbs0.bn.running_var', 'elan_h25.cbs7.bn.running_var'):
            out[f'{tag}.buf.{n}'] = b.numpy()
    save('net128.npz', **out)


# --------------------------------------------------------------------------- decode
def gen_decode():
    out = {}
    for tag, size, ncls, batch in (('d64', 64, 3, 2), ('d160', 160, 80, 1)):
        args = patterns.default_args(input_size=size, output_class=ncls)
        dec = ref_layer.decode(args)
        heads = [patterns.wave(f'{tag}.h{i}', (batch, 3, h, h, 5 + ncls), 3.0) for i, h in enumerate(args.output_size)]
        with torch.no_grad():
            y = dec([h.clone() for h in heads])
            out[f'{tag}.decoded'] = y.numpy()

            class _Id(torch.nn.Module):
                def forward(self, x):
                    return x
            out[f'{tag}.deployed'] = ref_layer.deploy(_Id())(y.clone()).numpy()
    save('decode.npz', **out)


# --------------------------------------------------------------------------- NMS
def make_pred(seed, n, ncls, size, kind):
    """Synthetic deployed prediction rows (cx,cy,w,h,conf,cls...) in [0,size]."""
    rng = np.random.default_rng(seed)
    p = np.zeros((n, 5 + ncls), f32)
    if kind == 'spread':
        p[:, 0:2] = rng.uniform(0, size, (n, 2))
        p[:, 2:4] = rng.uniform(4, size / 4, (n, 2))
    elif kind == 'dense':      # heavy overlap: few centres, jitter
        c = rng.uniform(size * 0.2, size * 0.8, (8, 2))
        p[:, 0:2] = c[rng.integers(0, 8, n)] + rng.normal(0, 6, (n, 2))
        p[:, 2:4] = rng.uniform(30, 60, (n, 2))
    elif kind == 'edge':       # many boxes hanging over the top/left border, some zero-area
        p[:, 0:2] = rng.uniform(-10, size * 0.3, (n, 2))
        p[:, 2:4] = rng.uniform(0, 40, (n, 2))
        p[::7, 2] = 0
        p[::11, 3] = 0
        p[::77, 2:4] = 0
    p[:, 4] = rng.uniform(0, 1, n)
    p[:, 5:] = rng.uniform(0, 1, (n, ncls))
    return p.astype(f32)


def gen_nms():
    out = {}
    pc = ref_predict.predict_class.__new__(ref_predict.predict_class)
    cases = [('spread', 1, 1500, 3, 640, 'spread', 0.5, 0.5, (480, 720, 3)),
             ('dense', 2, 1200, 1, 640, 'dense', 0.3, 0.5, (640, 640, 3)),
             ('dense_lo', 3, 900, 80, 640, 'dense', 0.001, 0.3, (1080, 1920, 3)),
             ('edge', 4, 1000, 2, 320, 'edge', 0.2, 0.45, (200, 300, 3)),
             ('empty', 5, 64, 2, 320, 'spread', 1.5, 0.5, (320, 320, 3)),
             ('single', 6, 1, 2, 320, 'spread', 0.0, 0.5, (320, 320, 3))]
    for tag, seed, n, ncls, size, kind, cthr, ithr, shape in cases:
        pred = make_pred(seed, n, ncls, size, kind)
        if tag == 'single':
            pred[0, 0:4] = (100, 100, 20, 20)
            pred[0, 4] = 0.9
        pc.input_size = size
        pc.confidence_threshold = cthr
        pc.iou_threshold = ithr
        res = pc._decode(pred.copy(), shape)
        out[f'{tag}.pred'] = pred
        out[f'{tag}.cfg'] = np.array([size, cthr, ithr, shape[0], shape[1]], np.float64)
        if res is None:
            out[f'{tag}.rows'] = np.zeros((0, 5 + ncls), f32)
            out[f'{tag}.kept'] = np.zeros((0,), np.int64)
            continue
        # recover kept indices: columns 4: pass through the reference untouched
        lut = {pred[i, 4:].tobytes(): i for i in range(n)}
        assert len(lut) == n, 'tag columns must be unique'
        kept = np.array([lut[r[4:].tobytes()] for r in res], np.int64)
        out[f'{tag}.rows'] = res.astype(f32)
        out[f'{tag}.kept'] = kept
        print(f'  nms {tag}: {n} rows -> {len(kept)} kept')
    # bare iou vector
    a = make_pred(9, 64, 1, 100, 'dense')[:, 0:4]
    out['iou.a'] = a
    out['iou.val'] = ref_predict.predict_class.iou(a[0:1], a)
    save('nms.npz', **out)


# --------------------------------------------------------------------------- training arithmetic
def gen_training():
    out = {}
    # loss on synthetic predictions, with and without positives, C=1 and C=4
    for tag, ncls in (('c1', 1), ('c4', 4)):
        b, r, k = 3, 60, 5 + ncls
        pred = patterns.wave(f'loss.{tag}.pred', (b, r, k), 2.0)
        pred[:, :, 0:2] = pred[:, :, 0:2] * 20 + 60
        pred[:, :, 2:4] = pred[:, :, 2:4].abs() * 15 + 5
        screen = [torch.tensor([1, 5, 17]), torch.tensor([], dtype=torch.int64), torch.tensor([0, 2, 30, 59])]
        lab = patterns.wave(f'loss.{tag}.lab', (7, k), 0.45, 0.5)
        lab[:, 0:2] = lab[:, 0:2] * 40 + 40
        lab[:, 2:4] = lab[:, 2:4] * 30 + 6
        largs = argparse.Namespace(device='cpu', input_size=128, output_size=(16, 8, 4), label_smooth=0.01,
                                   output_class=ncls)
        fn = ref_train.loss_class(largs)
        p = pred.clone().requires_grad_(True)
        vals = fn(p, screen, lab)
        vals[0].backward()
        out[f'{tag}.loss'] = np.array([v.item() for v in vals], f32)
        out[f'{tag}.dpred'] = p.grad.numpy()
        p2 = pred.clone().requires_grad_(True)
        vals = fn(p2, [torch.tensor([], dtype=torch.int64)] * b, lab[:0])
        vals[0].backward()
        out[f'{tag}.loss_nopos'] = np.array([float(v) for v in vals], f32)
        out[f'{tag}.dpred_nopos'] = p2.grad.numpy()
    # lr schedule: first 30 values, two settings
    for tag, (step_epoch, lr_end_epoch, warm) in (('a', (10, 100, 0.01)), ('b', (400, 3, 0.05))):
        a = argparse.Namespace(lr_start=0.001, lr_end_ratio=0.1, lr_end_epoch=lr_end_epoch, warmup_ratio=warm)
        sched = ref_train.lr_adjust(a, step_epoch, 0)

        class _Opt:
            param_groups = [{'lr': 0.0}]
        opt = _Opt()
        seq = []
        for _ in range(30):
            sched(opt)
            seq.append(opt.param_groups[0]['lr'])
        out[f'lr.{tag}'] = np.array(seq, np.float64)
    # three Adam steps + EMA on a small parameter set
    params = [torch.nn.Parameter(patterns.wave(f'adam.p{i}', s, 0.5)) for i, s in enumerate([(7, 5), (33,), (4, 3, 3, 3)])]
    opt = torch.optim.Adam(params, lr=0.001, betas=(0.937, 0.999), weight_decay=0.0005)

    class _M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ps = torch.nn.ParameterList(params)
            self.register_buffer('running', torch.zeros(5))
            self.register_buffer('count', torch.zeros((), dtype=torch.int64))
    m = _M()
    ema = ref_train.model_ema(m)
    for step in range(3):
        for i, p in enumerate(params):
            p.grad = patterns.wave(f'adam.g{i}.{step}', p.shape, 0.3)
        for g in opt.param_groups:
            g['lr'] = 0.001 * (step + 1)
        opt.step()
        m.running += 0.5
        ema.update(m)
    for i, p in enumerate(params):
        out[f'adam.p{i}'] = p.detach().numpy()
        out[f'adam.m{i}'] = opt.state[p]['exp_avg'].numpy()
        out[f'adam.v{i}'] = opt.state[p]['exp_avg_sq'].numpy()
        out[f'ema.p{i}'] = ema.ema_model.ps[i].detach().numpy()
    out['ema.running'] = ema.ema_model.running.numpy()
    save('training.npz', **out)


def gen_pruned():
    """yolov7(args, config=...) as the pruning step of train_class.py:367-386 rebuilds it: irregular cbs
    widths.  Stores the width list, every state_dict key with its shape, and the eval output."""
    args = patterns.default_args(input_size=64, output_class=3)
    std = ref_yolov7(args)
    widths = [m.conv.out_channels for m in std.modules() if isinstance(m, ref_layer.cbs)]
    cfg = [max(5, w - w // 5 - (i % 3)) for i, w in enumerate(widths)]
    net = ref_yolov7(args, config=cfg)
    patterns.fill_state_dict(net)
    net.eval()
    x = patterns.image_batch(2, 64)
    with torch.no_grad():
        y = net(x).numpy()
    sd = net.state_dict()
    keys = list(sd.keys())
    shapes = np.zeros((len(keys), 4), np.int64)
    for i, k in enumerate(keys):
        shp = tuple(sd[k].shape)
        shapes[i, :len(shp)] = shp
    save('pruned.npz', widths=np.asarray(widths), cfg=np.asarray(cfg), keys=np.asarray(keys), shapes=shapes, eval=y)


def gen_ckpt():
    """A checkpoint file exactly as the reference writes it (train_class.py:262-276: torch.save of a dict whose
    'model' entry is the pickled module object, class paths model.yolov7.yolov7 / model.layer.*), for a small
    pruned configuration so that the file stays small, plus the reference's own eval output for it."""
    args = patterns.default_args(input_size=64, output_class=3)
    std = ref_yolov7(args)
    widths = [m.conv.out_channels for m in std.modules() if isinstance(m, ref_layer.cbs)]
    cfg = [max(6, w // 6 + (i % 4)) for i, w in enumerate(widths)]
    net = ref_yolov7(args, config=cfg)
    patterns.fill_state_dict(net)
    net.eval()
    x = patterns.image_batch(2, 64)
    with torch.no_grad():
        y = net(x).numpy()
        yd = ref_layer.deploy(net)(x).numpy()
    model_dict = {'model': net, 'epoch_finished': 3, 'optimizer_state_dict': None, 'ema_update': 0, 'standard': 0.5,
                  'train_loss': 1.0, 'val_loss': 1.0, 'val_m_ap': 0.5}
    path = os.path.join(HERE, 'ref_ckpt.pt')
    torch.save(model_dict, path)
    print(f'ref_ckpt.pt: {os.path.getsize(path) / 1024:.1f} KiB')
    save('ref_ckpt.npz', cfg=np.asarray(cfg), eval=y, deploy=yd)


if __name__ == '__main__':
    if len(_ARGV) > 1 and _ARGV[1] == 'ckpt':
        gen_ckpt()
        sys.exit(0)
    if len(_ARGV) > 1 and _ARGV[1] == 'net128':
        gen_net128()
        sys.exit(0)
    if len(_ARGV) > 1 and _ARGV[1] == 'pruned':
        gen_pruned()
        sys.exit(0)
    gen_blocks()
    gen_net128()
    gen_net()
    gen_decode()
    gen_nms()
    gen_training()
    gen_pruned()
    gen_ckpt()
