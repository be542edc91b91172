"""GPU parity of the training-step kernels (BN train fwd/bwd, dgrad, wgrad, pools, loss, Adam+EMA)
against the fixtures recorded from the reference and the CPU oracle.  float32, tolerance 1e-4
(max-norm relative; weight gradients are sums over all pixels, so 2e-4 of their max)."""
import numpy as np
import pytest
import torch

from oracle import patterns, training as otrain

pytestmark = pytest.mark.gpu


def rel_err(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12)


@pytest.fixture(scope='module')
def te():
    import objectdetection_amd.train_engine as t
    return t


def _block_cases():
    from objectdetection_amd.model import layer
    return {'cbs_k1': (lambda: layer.cbs(16, 24, 1, 1), 'x16', (2, 16, 12, 12)),
            'cbs_k3': (lambda: layer.cbs(16, 16, 3, 1), 'x16', (2, 16, 12, 12)),
            'cbs_k3s2': (lambda: layer.cbs(16, 32, 3, 2), 'x16', (2, 16, 12, 12)),
            'elan': (lambda: layer.elan(16, 32, 1), 'x16', (2, 16, 12, 12)),
            'elan_n2': (lambda: layer.elan(16, 32, 2), 'x16', (2, 16, 12, 12)),
            'elan_h': (lambda: layer.elan_h(32, 16), 'x32', (2, 32, 8, 8)),
            'mp': (lambda: layer.mp(16, 32), 'x16', (2, 16, 12, 12)),
            'sppcspc': (lambda: layer.sppcspc(32, 16), 'xs', (2, 32, 20, 20))}


@pytest.mark.parametrize('tag', ['cbs_k1', 'cbs_k3', 'cbs_k3s2', 'elan', 'elan_n2', 'elan_h', 'mp', 'sppcspc'])
def test_block_train_step_matches_reference(te, golden, tag):
    g = golden('blocks.npz')
    make, xn, shape = _block_cases()[tag]
    mod = make()
    patterns.fill_state_dict(mod)
    mod = mod.cuda().train()
    state = te.FlatState(mod)
    plan = te.TrainPlan(mod, state, shape[0], shape[2], block_input=shape[1:])
    x = patterns.wave(xn, shape, 1.0).cuda()
    y = plan.forward(x)
    assert rel_err(y.cpu().numpy(), g[f'{tag}.train']) < 1e-4
    plan.gout_nchw.copy_(patterns.wave(tag + '.gy', tuple(y.shape), 1.0).cuda())
    state.grad.zero_()
    plan.backward(x)
    torch.cuda.synchronize()
    assert rel_err(plan.dx_nchw.cpu().numpy(), g[f'{tag}.dx']) < 1e-4
    for n, p in mod.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), g[f'{tag}.grad.{n}']) < 2e-4, n
    for n, b in mod.named_buffers():
        if b.dtype.is_floating_point:
            assert rel_err(b.cpu().numpy(), g[f'{tag}.buf.{n}']) < 1e-5, n


@pytest.mark.parametrize('tag', ['elan', 'elan_h'])
def test_sibling_fusion_of_the_training_plan(te, golden, tag):
    """The two 1x1 cbs blocks at the head of elan / elan_h run as one op of the summed width in the training plan
    (their tensors are adjacent in the flat state).  Both forms meet the reference fixture, and the fused plan has one
    convolution, one BatchNorm and one weight gradient fewer in each direction."""
    g = golden('blocks.npz')
    make, xn, shape = _block_cases()[tag]
    counts = {}
    for fuse in (True, False):
        te.TrainPlan.FUSE_SIBLINGS = fuse
        try:
            mod = make()
            patterns.fill_state_dict(mod)
            mod = mod.cuda().train()
            state = te.FlatState(mod)
            assert len(state.sibling_pairs) == 1
            plan = te.TrainPlan(mod, state, shape[0], shape[2], block_input=shape[1:])
        finally:
            te.TrainPlan.FUSE_SIBLINGS = True
        counts[fuse] = (sum(1 for f in plan.fwd if f[2] == 'conv'), sum(1 for f in plan.bwd if f[2] == 'wgrad'))
        x = patterns.wave(xn, shape, 1.0).cuda()
        y = plan.forward(x)
        assert rel_err(y.cpu().numpy(), g[f'{tag}.train']) < 1e-4
        plan.gout_nchw.copy_(patterns.wave(tag + '.gy', tuple(y.shape), 1.0).cuda())
        state.grad.zero_()
        plan.backward(x)
        torch.cuda.synchronize()
        assert rel_err(plan.dx_nchw.cpu().numpy(), g[f'{tag}.dx']) < 1e-4
        for n, p in mod.named_parameters():
            assert rel_err(p.grad.cpu().numpy(), g[f'{tag}.grad.{n}']) < 2e-4, (fuse, n)
        for n, b in mod.named_buffers():
            if b.dtype.is_floating_point:
                assert rel_err(b.cpu().numpy(), g[f'{tag}.buf.{n}']) < 1e-5, (fuse, n)
    assert counts[True][0] == counts[False][0] - 1 and counts[True][1] == counts[False][1] - 1, counts


def test_loss_matches_reference_fixture(te, golden):
    import ctypes as C
    from objectdetection_amd._lib import lib, check
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
        for s_list, labels, key in ((screen, lab, ''), ([torch.tensor([], dtype=torch.int64)] * b, lab[:0], '_nopos')):
            p = pred.cuda().contiguous()
            dp = torch.empty_like(p)
            out = torch.zeros(4, device='cuda')
            ws = torch.empty(int(lib.odamd_loss_workspace_bytes()), dtype=torch.uint8, device='cuda')
            idx = torch.cat([s.to(torch.int32) for s in s_list]).cuda() if len(labels) else None
            img = torch.cat([torch.full((len(s),), i, dtype=torch.int32) for i, s in enumerate(s_list)]).cuda() \
                if len(labels) else None
            lb = labels.cuda().contiguous() if len(labels) else None
            check(lib.odamd_detection_loss(p.data_ptr(), dp.data_ptr(), idx.data_ptr() if idx is not None else None,
                                           img.data_ptr() if img is not None else None,
                                           lb.data_ptr() if lb is not None else None, b, r, k, len(labels), 0.01,
                                           out.data_ptr(), ws.data_ptr(), ws.numel(), None))
            torch.cuda.synchronize()
            np.testing.assert_allclose(out.cpu().numpy(), g[f'{tag}.loss{key}'], rtol=2e-5)
            ref = g[f'{tag}.dpred{key}']
            assert rel_err(dp.cpu().numpy(), ref) < 1e-4


def test_adam_ema_matches_reference_fixture(te, golden):
    import ctypes as C
    from objectdetection_amd._lib import lib, check
    g = golden('training.npz')
    shapes = [(7, 5), (33,), (4, 3, 3, 3)]
    sizes = [int(np.prod(s)) for s in shapes]
    flat = torch.cat([patterns.wave(f'adam.p{i}', s, 0.5).reshape(-1) for i, s in enumerate(shapes)]).cuda()
    m = torch.zeros_like(flat)
    v = torch.zeros_like(flat)
    ema = flat.clone()
    run, ema_run = torch.zeros(8, device='cuda'), torch.zeros(8, device='cuda')
    for step in range(3):
        grad = torch.cat([patterns.wave(f'adam.g{i}.{step}', s, 0.3).reshape(-1) for i, s in enumerate(shapes)]).cuda()
        d = otrain.ema_decay(step + 1)
        check(lib.odamd_adam_ema_step(flat.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), ema.data_ptr(),
                                      flat.numel(), 0.001 * (step + 1), 0.937, 0.999, 1e-8, 0.0005, step + 1, d, 1.0,
                                      None))
        run[:5] += 0.5
        check(lib.odamd_ema_update(ema_run.data_ptr(), run.data_ptr(), 8, d, None))
    torch.cuda.synchronize()
    o = 0
    for i, n in enumerate(sizes):
        np.testing.assert_allclose(flat[o:o + n].cpu().numpy(), g[f'adam.p{i}'].reshape(-1), rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(m[o:o + n].cpu().numpy(), g[f'adam.m{i}'].reshape(-1), rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(v[o:o + n].cpu().numpy(), g[f'adam.v{i}'].reshape(-1), rtol=1e-5, atol=1e-12)
        np.testing.assert_allclose(ema[o:o + n].cpu().numpy(), g[f'ema.p{i}'].reshape(-1), rtol=1e-5, atol=1e-9)
        o += n
    np.testing.assert_allclose(ema_run[:5].cpu().numpy(), g['ema.running'], rtol=1e-5)


def _oracle_f64_step(tag, size, ncls, batch, g, net):
    """fp64 CPU oracle of the same training step: prediction and every parameter gradient."""
    from oracle import network
    args = patterns.default_args(input_size=size, output_class=ncls)
    sd = {k: (v.detach().cpu().double().clone().requires_grad_(True) if k.endswith(('weight', 'bias'))
              else (v.detach().cpu().double() if v.dtype.is_floating_point else v.detach().cpu().clone()))
          for k, v in net.state_dict().items()}
    x = patterns.image_batch(batch, size).double()
    pred = network.yolov7_forward(sd, x, args, training=True)
    screen = [torch.from_numpy(g[f'{tag}.screen{b}']) for b in range(batch)]
    vals = otrain.detection_loss(pred, screen, torch.from_numpy(g[f'{tag}.label']).double(), 0.01, ncls)
    vals[0].backward()
    return pred.detach().numpy(), {k: v.grad.numpy() for k, v in sd.items() if getattr(v, 'requires_grad', False)}


NOISE_FACTOR = 5      # GPU error vs fp64 <= NOISE_FACTOR x the reference fixture's own error vs fp64 (+ 1e-4);
                      # measured ratios on MI355X: forward 2.2 / 1.1, gradients 1.8, 1.5, 0.7, 4.3, 1.1, 0.4, 0.7, all parameters 1.1


@pytest.mark.parametrize('tag,size,ncls,batch,fname', [('s128', 128, 5, 4, 'net128.npz')])
def test_whole_network_train_step_matches_reference(te, golden, tag, size, ncls, batch, fname):
    """Forward (train-mode BN), loss and backward of the whole network.

    Through ~80 train-mode BatchNorms the float32 reference itself is only reproducible to about
    1e-4 (forward) and 1e-3..1e-2 (early-layer gradients): its fixture differs from an fp64
    evaluation of the same step by that much, and so does a 1-thread rerun of torch.  The bar here
    is therefore "as close to the fp64 truth as the reference is": every GPU error against the fp64
    oracle must stay within NOISE_FACTOR (5) x the reference fixture's own error (+1e-4; both are single draws of
    rounding noise that the network amplifies).  The per-block tests above pin the kernels at 1e-4
    against the reference directly.  (The 64x64 fixture is not used here: its deepest maps are 2x2,
    so BatchNorm runs over 8 values per channel and the step is too ill-conditioned to compare.)"""
    from objectdetection_amd.model.yolov7 import yolov7
    g = golden(fname)
    args = patterns.default_args(input_size=size, output_class=ncls)
    net = yolov7(args)
    patterns.fill_state_dict(net)
    p64, g64 = _oracle_f64_step(tag, size, ncls, batch, g, net)
    net = net.cuda().train()
    state = te.FlatState(net)
    plan = te.TrainPlan(net, state, batch, size)
    x = patterns.image_batch(batch, size).cuda()
    pred = plan.forward(x).cpu().numpy()
    if f'{tag}.train' in g.files:
        ref = g[f'{tag}.train']
    else:
        pred, p64, ref = pred[:, ::7], p64[:, ::7], g[f'{tag}.train_sample']
    fwd_noise = 0.0
    for sl, what in ((slice(0, 4), 'boxes'), (slice(4, None), 'logits')):
        e_gpu, e_ref = rel_err(pred[..., sl], p64[..., sl]), rel_err(ref[..., sl], p64[..., sl])
        fwd_noise = max(fwd_noise, e_gpu, e_ref)
        print(tag, 'forward', what, 'gpu-vs-f64', e_gpu, 'reference-vs-f64', e_ref)
        assert e_gpu <= NOISE_FACTOR * e_ref + 1e-4, what
    screen = [torch.from_numpy(g[f'{tag}.screen{b}']) for b in range(batch)]
    lab = torch.from_numpy(g[f'{tag}.label'])
    loss = plan.loss(screen, lab, 0.01).cpu().numpy()
    print(tag, 'loss', loss, g[f'{tag}.loss'])
    np.testing.assert_allclose(loss, g[f'{tag}.loss'], rtol=1e-4 + 2 * fwd_noise)
    state.grad.zero_()
    plan.backward(x)
    torch.cuda.synchronize()
    params = dict(net.named_parameters())
    for key in g.files:
        if key.startswith(f'{tag}.grad.'):
            n = key[len(tag) + 6:]
            e_gpu, e_ref = rel_err(params[n].grad.cpu().numpy(), g64[n]), rel_err(g[key], g64[n])
            print(tag, 'grad', n, 'gpu-vs-f64', e_gpu, 'reference-vs-f64', e_ref)
            assert e_gpu <= NOISE_FACTOR * e_ref + 1e-4, n
        elif key.startswith(f'{tag}.buf.'):
            n = key[len(tag) + 5:]
            assert rel_err(dict(net.named_buffers())[n].cpu().numpy(), g[key]) < 1e-4, n
    # every parameter: gradient norm against fp64, within NOISE_FACTOR x the worst reference deviation
    worst_ref = max(abs(g[k][0] - np.linalg.norm(g64[k[len(tag) + 7:]].ravel())) /
                    (np.linalg.norm(g64[k[len(tag) + 7:]].ravel()) + 1e-30) for k in g.files if k.startswith(f'{tag}.gnorm.'))
    worst_gpu = max(rel_err(p.grad.cpu().numpy(), g64[n]) for n, p in params.items())
    print(tag, 'all parameters: worst gpu gradient error vs f64', worst_gpu, ' worst reference norm error', worst_ref)
    assert worst_gpu <= NOISE_FACTOR * worst_ref + 1e-4


@pytest.mark.parametrize('size,batch', [(128, 4), (256, 4)])
def test_whole_network_train_step_bf16(te, size, batch):
    """bf16 mixed-precision step of the whole network against the fp64 oracle of the same step (random
    initialisation, synthetic labels).

    In train mode this network amplifies perturbations strongly (57 batch-statistics BatchNorms in series): in
    fp64 arithmetic a relative input noise of 1e-3 moves the logits by 7 %, and merely STORING weights, raw
    convolution outputs and activations as bf16 (oracle act_bf16, all arithmetic still fp64) moves them by
    ~25 % rms.  That deviation is a property of the precision policy, not of a kernel, and two correct bf16
    implementations differ from each other by as much (a different rounding of one element in 4000 is enough).
    The bar is therefore relative to that emulation: the kernels must be as close to fp64 as the emulated
    storage precision is - forward rms, loss, and the direction of the whole gradient.  Kernel-level bf16
    accuracy (2e-2 / 5e-2) is pinned block by block in test_block_train_step_bf16."""
    from objectdetection_amd.model.yolov7 import yolov7
    from objectdetection_amd.engine import BF16
    from objectdetection_amd.train_class import synthetic_dataset
    from oracle import network
    ncls = 5
    args = patterns.default_args(input_size=size, output_class=ncls)
    torch.manual_seed(128)
    net = yolov7(args)
    ds = synthetic_dataset(args, batch, seed=7)
    image, screen, expend, _ = ds.collate_fn([ds[i] for i in range(batch)])

    def oracle(emulate):
        sd = {k: (v.detach().double().clone().requires_grad_(True) if k.endswith(('weight', 'bias'))
                  else (v.detach().double() if v.dtype.is_floating_point else v.detach().clone()))
              for k, v in net.state_dict().items()}
        p = network.yolov7_forward(sd, image.double(), args, training=True, act_bf16=emulate)
        vals = otrain.detection_loss(p, screen, expend.double(), args.label_smooth, ncls)
        vals[0].backward()
        grads = np.concatenate([sd[n].grad.numpy().ravel() for n, _ in net.named_parameters()])
        return p.detach().numpy(), np.array([float(v.detach()) for v in vals]), grads

    p64, loss64, g64 = oracle(False)
    pq, lossq, gq = oracle(True)
    net = net.cuda().train()
    state = te.FlatState(net)
    plan = te.TrainPlan(net, state, batch, size, dtype=BF16)
    x = image.cuda()
    pred = plan.forward(x).cpu().numpy()
    loss = plan.loss(screen, expend, args.label_smooth).cpu().numpy()
    state.grad.zero_()
    plan.backward(x)
    torch.cuda.synchronize()
    ga = np.concatenate([p.grad.double().cpu().numpy().ravel() for _, p in net.named_parameters()])

    def rms(a, b, sl):
        return float(np.sqrt(np.mean((a[..., sl] - b[..., sl]) ** 2)) / np.sqrt(np.mean(b[..., sl] ** 2)))

    def cos(a, b):
        return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))

    lg, bx = slice(4, None), slice(0, 4)
    m = dict(gpu_logits=rms(pred, p64, lg), emu_logits=rms(pq, p64, lg), gpu_boxes=rms(pred, p64, bx), emu_boxes=rms(pq, p64, bx),
             gpu_cos=cos(ga, g64), emu_cos=cos(gq, g64), gpu_vs_emu_cos=cos(ga, gq),
             gpu_norm=float(np.linalg.norm(ga) / np.linalg.norm(g64)), emu_norm=float(np.linalg.norm(gq) / np.linalg.norm(g64)))
    print(f'bf16 whole network {size}x{size} b{batch}:', {k: round(v, 4) for k, v in m.items()}, 'loss', loss, loss64, lossq)
    # Measured on MI355X (128 x 128 / 256 x 256): logits rms 0.276 / 0.259 against the emulation's 0.262 / 0.251, boxes
    # 0.033 / 0.026 against 0.027 / 0.026, gradient cosine 0.569 / 0.590 against 0.642 / 0.633, gradient norm ratio
    # 0.961 / 0.972 against 1.057 / 0.914, loss terms within 1.2e-3 / 5.5e-3 / 3e-3 / 1.4e-2 of fp64.  The bars are
    # those values with about 30 % of margin (any change of a rounding moves them by a few per cent).
    np.testing.assert_allclose(loss, loss64, rtol=2e-2)
    assert m['gpu_logits'] <= 1.25 * m['emu_logits'], m
    assert m['gpu_boxes'] <= 1.5 * m['emu_boxes'] + 0.003, m
    assert m['gpu_cos'] >= m['emu_cos'] - 0.10, m
    assert abs(m['gpu_norm'] - 1) <= 0.06, m


def test_whole_network_train_step_640_f32(te):
    """One float32 training step at the full input size (640 x 640, batch 4, 5 classes, random initialisation under a
    seed, synthetic labels): prediction, the loss 4-tuple and every parameter gradient against the fp64 CPU oracle of
    the same step.  The bar is the one of the 128 x 128 fixture test: the kernels' error against fp64 may be at most
    NOISE_FACTOR x the error of the oracle evaluated in float32 (plain torch CPU arithmetic, i.e. what the reference
    computes) against fp64, + 1e-4."""
    from objectdetection_amd.model.yolov7 import yolov7
    from objectdetection_amd.train_class import synthetic_dataset
    from oracle import network
    size, batch, ncls = 640, 4, 5
    args = patterns.default_args(input_size=size, output_class=ncls)
    torch.manual_seed(640)
    net = yolov7(args)
    ds = synthetic_dataset(args, batch, seed=11)
    image, screen, expend, _ = ds.collate_fn([ds[i] for i in range(batch)])

    def oracle(dt):
        sd = {k: (v.detach().to(dt).clone().requires_grad_(True) if k.endswith(('weight', 'bias'))
                  else (v.detach().to(dt) if v.dtype.is_floating_point else v.detach().clone()))
              for k, v in net.state_dict().items()}
        p = network.yolov7_forward(sd, image.to(dt), args, training=True)
        vals = otrain.detection_loss(p, screen, expend.to(dt), args.label_smooth, ncls)
        vals[0].backward()
        return (p.detach().double().numpy(), np.array([float(v.detach()) for v in vals]),
                {n: sd[n].grad.double().numpy() for n, _ in net.named_parameters()})

    p64, loss64, g64 = oracle(torch.float64)
    p32, loss32, g32 = oracle(torch.float32)
    net = net.cuda().train()
    state = te.FlatState(net)
    plan = te.TrainPlan(net, state, batch, size)
    x = image.cuda()
    pred = plan.forward(x).cpu().numpy()
    loss = plan.loss(screen, expend, args.label_smooth).cpu().numpy()
    state.grad.zero_()
    plan.backward(x)
    torch.cuda.synchronize()
    for sl, what in ((slice(0, 4), 'boxes'), (slice(4, None), 'logits')):
        e_gpu, e_ref = rel_err(pred[..., sl], p64[..., sl]), rel_err(p32[..., sl], p64[..., sl])
        print('640 forward', what, 'gpu-vs-f64', e_gpu, 'f32-oracle-vs-f64', e_ref)
        assert e_gpu <= NOISE_FACTOR * e_ref + 1e-4, what
    print('640 loss', loss, loss64, loss32)
    np.testing.assert_allclose(loss, loss64, rtol=1e-4 + NOISE_FACTOR * float(np.abs(loss32 / loss64 - 1).max()))
    worst_gpu = worst_ref = 0.0
    for n, p in net.named_parameters():
        e_gpu, e_ref = rel_err(p.grad.cpu().numpy(), g64[n]), rel_err(g32[n], g64[n])
        worst_gpu, worst_ref = max(worst_gpu, e_gpu), max(worst_ref, e_ref)
    ga = np.concatenate([p.grad.double().cpu().numpy().ravel() for _, p in net.named_parameters()])
    gf = np.concatenate([g64[n].ravel() for n, _ in net.named_parameters()])
    gr = np.concatenate([g32[n].ravel() for n, _ in net.named_parameters()])
    n_gpu, n_ref = abs(np.linalg.norm(ga) / np.linalg.norm(gf) - 1), abs(np.linalg.norm(gr) / np.linalg.norm(gf) - 1)
    print('640 gradients: worst per-parameter error gpu', worst_gpu, 'f32 oracle', worst_ref, ' norm error gpu', n_gpu,
          'f32 oracle', n_ref)
    assert worst_gpu <= NOISE_FACTOR * worst_ref + 1e-4
    assert n_gpu <= NOISE_FACTOR * n_ref + 1e-4


@pytest.mark.parametrize('cin,cout,k,s,h,w,b', [
    (3, 16, 3, 1, 32, 32, 2), (32, 32, 3, 1, 20, 20, 3), (64, 64, 3, 1, 24, 40, 2), (32, 64, 3, 2, 40, 40, 2),
    (128, 256, 3, 2, 20, 20, 2), (64, 32, 1, 1, 40, 40, 2), (128, 128, 1, 1, 20, 20, 4), (256, 30, 1, 1, 20, 20, 2),
    (40, 72, 3, 1, 10, 14, 2), (16, 32, 3, 2, 64, 64, 1), (512, 256, 1, 1, 4, 4, 3), (256, 128, 3, 1, 8, 8, 2)])
def test_conv_wgrad_shapes(cin, cout, k, s, h, w, b):
    """Weight gradient launch alone, every tile configuration, against float64 autograd; twice into the
    same dw (accumulation) and bitwise repeatable."""
    from objectdetection_amd._lib import lib, check
    g = torch.Generator().manual_seed(cin * 1000 + cout)
    cpi, cpo = -(-cin // 8) * 8, -(-cout // 8) * 8
    ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
    x = torch.randn(b, h, w, cpi, generator=g)
    dy = torch.randn(b, ho, wo, cpo, generator=g)
    xd = x[..., :cin].permute(0, 3, 1, 2).double()
    dyd = dy[..., :cout].permute(0, 3, 1, 2).double()
    wgt = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(xd, wgt, stride=s, padding=k // 2).backward(dyd)
    need = int(lib.odamd_conv_wgrad_workspace_bytes(b, h, w, cin, cout, k, s, 0))
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device='cuda')
    xg, dyg = x.cuda(), dy.cuda()
    outs = []
    for rep in range(2):
        dw = torch.zeros(cout, cin, k, k, device='cuda')
        for _ in range(rep + 1):
            check(lib.odamd_conv_wgrad(xg.data_ptr(), cpi, dyg.data_ptr(), cpo, dw.data_ptr(), b, h, w, cin, cout, k, s, 0,
                                       ws.data_ptr(), need, None))
        torch.cuda.synchronize()
        outs.append(dw.cpu())
    ref = wgt.grad
    scale = float(ref.abs().max())
    assert float((outs[0].double() - ref).abs().max()) <= 2e-5 * scale
    assert float((outs[1].double() - 2 * ref).abs().max()) <= 4e-5 * scale
    dw = torch.zeros(cout, cin, k, k, device='cuda')
    check(lib.odamd_conv_wgrad(xg.data_ptr(), cpi, dyg.data_ptr(), cpo, dw.data_ptr(), b, h, w, cin, cout, k, s, 0,
                               ws.data_ptr(), need, None))
    assert torch.equal(dw.cpu(), outs[0])
    assert lib.odamd_conv_wgrad(xg.data_ptr(), cpi, dyg.data_ptr(), cpo, dw.data_ptr(), b, h, w, cin, cout, k, s, 0,
                                ws.data_ptr(), need - 4, None) == -3


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('cin,cout,k,s,h,w,b', [
    (3, 16, 3, 1, 32, 32, 2), (32, 32, 3, 1, 20, 20, 3), (64, 64, 3, 1, 24, 40, 2), (32, 64, 3, 2, 40, 40, 2),
    (128, 256, 3, 2, 20, 20, 2), (64, 32, 1, 1, 40, 40, 2), (128, 128, 1, 1, 20, 20, 4), (256, 30, 1, 1, 20, 20, 2),
    (40, 72, 3, 1, 10, 14, 2), (16, 32, 3, 2, 64, 64, 1), (512, 256, 1, 1, 4, 4, 3), (256, 128, 3, 1, 8, 8, 2)])
def test_conv_wgrad_dtypes(cin, cout, k, s, h, w, b, dtype):
    """Weight gradient on float32 and bf16 operands (values exactly representable in bf16, so both
    modes have the same float64 reference and only the float32 accumulation differs)."""
    from objectdetection_amd._lib import lib, check
    g = torch.Generator().manual_seed(cin * 1000 + cout + 7)
    cpi, cpo = -(-cin // 8) * 8, -(-cout // 8) * 8
    ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
    x = torch.randn(b, h, w, cpi, generator=g).bfloat16().float()
    dy = torch.randn(b, ho, wo, cpo, generator=g).bfloat16().float()
    wgt = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x[..., :cin].permute(0, 3, 1, 2).double(), wgt, stride=s, padding=k // 2).backward(
        dy[..., :cout].permute(0, 3, 1, 2).double())
    dt = 0 if dtype == 'f32' else 1
    need = int(lib.odamd_conv_wgrad_workspace_bytes(b, h, w, cin, cout, k, s, dt))
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device='cuda')
    xg = x.cuda() if dt == 0 else x.cuda().bfloat16()
    dyg = dy.cuda() if dt == 0 else dy.cuda().bfloat16()
    dw = torch.zeros(cout, cin, k, k, device='cuda')
    for _ in range(2):
        check(lib.odamd_conv_wgrad(xg.data_ptr(), cpi, dyg.data_ptr(), cpo, dw.data_ptr(), b, h, w, cin, cout, k, s, dt,
                                   ws.data_ptr(), need, None))
    torch.cuda.synchronize()
    ref = 2 * wgt.grad
    assert float((dw.cpu().double() - ref).abs().max()) <= 4e-5 * float(ref.abs().max())


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('cin,cout,k,s,h,w,b', [
    (64, 64, 1, 1, 80, 80, 8), (128, 128, 1, 1, 37, 53, 3), (32, 32, 3, 1, 160, 160, 2), (16, 32, 3, 2, 128, 128, 4),
    (3, 16, 3, 1, 96, 96, 4), (64, 64, 3, 2, 80, 80, 4), (256, 256, 1, 1, 40, 40, 8)])
def test_conv_wgrad_many_patches(cin, cout, k, s, h, w, b, dtype):
    """Shapes where a workgroup walks several patches (next patch's loads in flight during the matrix work), the
    flat pixel list of a 1x1 layer ends in a ragged patch, and the split count is chosen by the cost estimate."""
    test_conv_wgrad_dtypes(cin, cout, k, s, h, w, b, dtype)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('kind,kw,shape', [
    ('elan', dict(in_=32, out_=64, n=2), (2, 32, 16, 16)),
    ('mp', dict(in_=32, out_=64), (2, 32, 16, 16)),
    ('sppcspc', dict(in_=64, out_=32), (2, 64, 20, 20)),
    ('elan_h', dict(in_=256, out_=128), (4, 256, 20, 20))])
def test_backward_write_modes_and_weight_layouts_agree(te, kind, kw, shape, dtype):
    """The first data-gradient launch that writes a gradient range stores instead of adding into a zero-filled
    buffer, and bf16 launches read fragment-ordered weight copies on the packed-weight kernels where one applies:
    both are choices of the plan, not of the arithmetic.  Store-first against add-everywhere is bit-identical;
    packed against plain kernels agrees to bf16 rounding of the stored tensors."""
    from objectdetection_amd.model import layer
    from objectdetection_amd.engine import BF16, F32
    torch.manual_seed(31)
    mod = getattr(layer, kind)(**kw).cuda().train()
    g = torch.Generator().manual_seed(32)
    x = torch.randn(shape, generator=g).cuda()
    results = []
    for store_first, packed in ((True, True), (False, True), (True, False)):
        te.TrainPlan.STORE_FIRST, te.TrainPlan.PACKED_WEIGHTS = store_first, packed
        try:
            state = te.FlatState(mod)
            plan = te.TrainPlan(mod, state, shape[0], shape[2], dtype=BF16 if dtype == 'bf16' else F32, block_input=shape[1:])
        finally:
            te.TrainPlan.STORE_FIRST, te.TrainPlan.PACKED_WEIGHTS = True, True
        y = plan.forward(x).clone()
        if not results:
            gy = torch.randn(y.shape, generator=g).cuda()
        plan.gout_nchw.copy_(gy)
        state.grad.zero_()
        plan.backward(x)
        torch.cuda.synchronize()
        results.append([y.cpu(), plan.dx_nchw.cpu().clone(), state.grad.cpu().clone()])
    base, added, plain = results
    # no kernel of the step adds with atomics (the stride-1 max pools of sppcspc gather in a fixed order): bitwise equal
    for name, a, b2 in zip(('y', 'dx', 'grad'), base, added):
        assert torch.equal(a, b2), (name, int((a != b2).sum()), float((a - b2).abs().max()), float(a.abs().max()))
    for a, b2 in zip(base, plain):
        scale = float(a.abs().max()) + 1e-30
        assert float((a - b2).abs().max()) <= (2e-2 if dtype == 'bf16' else 0.0) * scale


def _oracle_block_train(kind, mod, x, gy, act_bf16=False):
    """float32 CPU autograd of one block in train mode: output, dx, parameter gradients.  act_bf16: block
    activations rounded to bf16 (what the bf16 path pools), gradients straight through."""
    from oracle import network
    sd = {}
    for key, v in mod.state_dict().items():
        t = v.detach().cpu().clone()
        if t.dtype.is_floating_point and key.endswith(('weight', 'bias')):
            t.requires_grad_(True)
        sd['.' + key] = t
    net = network.Net(sd, training=True, act_bf16=act_bf16)
    xr = x.clone().requires_grad_(True)
    y = net.cbs('', xr, mod.conv.stride[0]) if kind == 'cbs' else getattr(net, kind)('', xr)
    y.backward(gy)
    return y.detach(), xr.grad, {k[1:]: v.grad for k, v in sd.items() if v.requires_grad}


@pytest.mark.parametrize('dtype', ['bf16', 'f32'])
@pytest.mark.parametrize('h,w,c', [(20, 20, 16), (13, 17, 8), (32, 32, 24), (40, 40, 8)])
def test_spp_backward_fused_with_ties(h, w, c, dtype):
    """odamd_spp_backward (one launch for the k = 5, 9, 13 pools of model/layer.py:215-217) on coarsely quantised
    inputs with ties everywhere: each window's gradient goes to its first maximum in row-major order, like torch; the
    three contributions are summed in float32 (and rounded to bf16 once).  Planes that fit the LDS take the fixed-order
    gather (no atomics: two runs are bitwise equal); 40 x 40 takes the three separate launches."""
    from objectdetection_amd._lib import lib, check
    if dtype == 'bf16' and h * w > 1024:
        pytest.skip('bf16 planes beyond 1024 pixels are not supported by the pool backward kernels')
    g = torch.Generator().manual_seed(9)
    b = 3
    tdt, code, es = (torch.bfloat16, 1, 2) if dtype == 'bf16' else (torch.float32, 0, 4)
    x = (torch.randn(b, c, h, w, generator=g) * 2).round() / 2
    xr = x.clone().requires_grad_(True)
    gys, total = [], 0
    for k in (5, 9, 13):
        y = torch.nn.functional.max_pool2d(xr, k, 1, k // 2)
        gy = (torch.randn(y.shape, generator=g) * 4).round() / 4
        total = total + (y * gy).sum()
        gys.append(gy)
    total.backward()
    xg = x.permute(0, 2, 3, 1).contiguous().to(tdt).cuda()
    gyg = torch.cat([t.permute(0, 2, 3, 1) for t in gys], dim=3).contiguous().to(tdt).cuda()      # three slices of one buffer
    base = ((torch.randn(b, h, w, c, generator=g) * 8).round() / 8).to(tdt).cuda()
    p0 = gyg.data_ptr()
    runs = []
    for _ in range(2):
        dx = base.clone()
        check(lib.odamd_spp_backward(xg.data_ptr(), c, p0, p0 + es * c, p0 + 2 * es * c, 3 * c, dx.data_ptr(), c, b, h, w, c, code, None))
        torch.cuda.synchronize()
        runs.append(dx.float().cpu())
    want = (base.float().cpu() + xr.grad.permute(0, 2, 3, 1)).to(tdt).float()
    one_launch = h * w * (8 * 19 if dtype == 'bf16' else 4 * 25) <= 160 * 1024 and w <= 255
    if one_launch or dtype == 'f32':     # exact sums of quantised values in float32, one rounding
        assert torch.equal(runs[0], want)
    else:                                # three launches: dx is rounded to bf16 after each of them
        assert float((runs[0] - want).abs().max()) <= 2.0 ** -6 * float(want.abs().max())
    if one_launch:
        assert torch.equal(runs[0], runs[1])


@pytest.mark.parametrize('kind,kw,shape', [
    ('cbs', dict(in_=16, out_=32, kernel_size=1, stride=1), (2, 16, 16, 16)),
    ('cbs', dict(in_=32, out_=32, kernel_size=3, stride=1), (2, 32, 20, 20)),
    ('cbs', dict(in_=16, out_=32, kernel_size=3, stride=2), (2, 16, 24, 24)),
    ('elan', dict(in_=32, out_=64, n=2), (2, 32, 16, 16)),
    ('elan_h', dict(in_=64, out_=32), (2, 64, 12, 12)),
    ('mp', dict(in_=32, out_=64), (2, 32, 16, 16)),
    ('sppcspc', dict(in_=64, out_=32), (2, 64, 20, 20))])
def test_block_train_step_bf16(te, kind, kw, shape):
    """bf16 mixed-precision training step of every block type (bf16 activations, gradients and MFMA
    operands; float32 accumulation, BN statistics and parameter gradients) against float32 CPU
    autograd; tolerance 2e-2 of each tensor's max."""
    from objectdetection_amd.model import layer
    from objectdetection_amd.engine import BF16
    torch.manual_seed(21)
    mod = getattr(layer, kind)(**kw)
    g = torch.Generator().manual_seed(22)
    with torch.no_grad():
        for m in mod.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    x = torch.randn(shape, generator=g)
    mod.train()
    probe = _oracle_block_train(kind, mod, x, None) if False else None
    with torch.no_grad():
        yshape = mod.cpu().eval()(x.cuda()).shape if False else None
    # output shape from the oracle itself
    sd0 = {('.' + k2): v.detach().clone() for k2, v in mod.state_dict().items()}
    from oracle import network
    with torch.no_grad():
        y0 = (network.Net(sd0, training=True).cbs('', x, mod.conv.stride[0]) if kind == 'cbs'
              else getattr(network.Net(sd0, training=True), kind)('', x))
    gy = torch.randn(y0.shape, generator=g)
    y_ref, dx_ref, grads = _oracle_block_train(kind, mod, x, gy)
    _, dx_q, grads_q = _oracle_block_train(kind, mod, x, gy, act_bf16=True)
    mod = mod.cuda().train()
    state = te.FlatState(mod)
    plan = te.TrainPlan(mod, state, shape[0], shape[2], dtype=BF16, block_input=shape[1:])
    y = plan.forward(x.cuda())
    assert rel_err(y.cpu().numpy(), y_ref.numpy()) < 2e-2
    plan.gout_nchw.copy_(gy.cuda())
    state.grad.zero_()
    plan.backward(x.cuda())
    torch.cuda.synchronize()
    pairs = {'dx': (plan.dx_nchw.cpu(), dx_ref)}
    for n, p in mod.named_parameters():
        pairs[n] = (p.grad.cpu(), grads[n])
    if kind in ('mp', 'sppcspc'):
        # Max-pool backward sends a window's gradient to its arg-max.  bf16 activations turn float32
        # near-ties into ties or flip them, so against the float32 oracle a few windows route their gradient
        # to a neighbouring pixel (a property of pooling low-precision activations, not of the kernels:
        # test_pool_backward_kernels_with_ties pins those exactly).  The comparison is therefore tie-aware: the
        # oracle pools the same bf16-rounded activations the kernels pool (act_bf16), which leaves only the
        # windows whose two largest values differ by the last bf16 bit; the bar is every tensor's direction
        # and size.  Against the plain float32 oracle the direction alone is checked, loosely.
        refs = dict(grads_q, dx=dx_q)
        cos_q, cos_f, nrm = {}, {}, {}
        for k2, (a, r) in pairs.items():
            rq = refs[k2]
            cos_q[k2] = float(torch.nn.functional.cosine_similarity(a.flatten().double(), rq.flatten().double(), dim=0))
            cos_f[k2] = float(torch.nn.functional.cosine_similarity(a.flatten().double(), r.flatten().double(), dim=0))
            nrm[k2] = abs(float(a.double().norm() / rq.double().norm()) - 1.0)
        print(kind, 'bf16 pool block: min cosine vs bf16-activation oracle', min(cos_q.values()), 'vs f32 oracle',
              min(cos_f.values()), 'max norm deviation', max(nrm.values()))
        bad = {k2: (cos_q[k2], nrm[k2]) for k2 in pairs if not (cos_q[k2] > 0.99 and nrm[k2] < 0.08)}
        assert not bad, bad
        assert min(cos_f.values()) > 0.9, cos_f
    else:
        # conv-only blocks: 2e-2 for the data gradient; parameter gradients are sums of such terms over
        # all pixels behind up to six bf16 layers: 5e-2
        errs = {k2: rel_err(a.numpy(), r.numpy()) for k2, (a, r) in pairs.items()}
        bad = {k2: v for k2, v in errs.items() if not v < (2e-2 if k2 == 'dx' else 5e-2)}
        assert not bad, (bad, max(errs.values()))


@pytest.mark.parametrize('accum', [0, 1])
@pytest.mark.parametrize('cdy,cdx,hr,wr,b', [(32, 16, 12, 12, 2), (64, 32, 10, 20, 2), (32, 32, 7, 13, 3), (64, 8, 5, 9, 1)])
def test_stride2_data_gradient_direct_kernel(cdy, cdx, hr, wr, b, accum):
    """Data gradient of a stride-2 3x3 convolution with few channels (bf16): the kernel that reads the four shifted dY
    blocks straight from global memory against float64 conv_transpose2d, stored and added to an existing gradient, on
    grids that are not whole 4 x 8 blocks; and against the generic kernel (forced by passing a unit scale)."""
    import ctypes as C
    from objectdetection_amd import _lib
    from objectdetection_amd._lib import lib, check
    g = torch.Generator().manual_seed(cdy * 100 + cdx + hr)
    dy = torch.randn(b, hr, wr, cdy, generator=g).bfloat16()
    w = (torch.randn(cdy, cdx, 3, 3, generator=g) * 0.2).bfloat16()          # forward weight (cout = cdy, cin = cdx)
    base = torch.randn(b, 2 * hr, 2 * wr, cdx, generator=g).bfloat16()
    ref = torch.nn.functional.conv_transpose2d(dy.double().permute(0, 3, 1, 2), w.double(), stride=2, padding=1,
                                               output_padding=1).permute(0, 2, 3, 1)
    if accum:
        ref = ref + base.double()
    # data-gradient weights: [dX channel][2 - ky][2 - kx][dY channel]
    wb = w.flip(2, 3).permute(1, 2, 3, 0).contiguous().cuda()
    dyg = dy.cuda()
    ones, zeros = torch.ones(64, device='cuda'), torch.zeros(64, device='cuda')
    outs = []
    for generic in (False, True):
        dx = base.clone().cuda() if accum else torch.full_like(base, 7.0).cuda()
        d = _lib.ConvDesc()
        d.x, d.w, d.y = dyg.data_ptr(), wb.data_ptr(), dx.data_ptr()
        d.scale, d.shift = (ones.data_ptr(), zeros.data_ptr()) if generic else (None, None)
        d.batch, d.h, d.w_in, d.cin, d.cout = b, hr, wr, cdy, cdx
        d.x_ld, d.y_ld, d.ksize, d.stride, d.act = cdy, cdx, 3, 1, 0
        d.dtype, d.in_pool2, d.out_nchw_f32, d.in_upzero2, d.accumulate = 1, 0, 0, 1, accum
        check(lib.odamd_conv_forward(C.byref(d), None))
        torch.cuda.synchronize()
        outs.append(dx.float().cpu().double())
    scale = float(ref.abs().max())
    for got in outs:
        assert float((got - ref).abs().max()) <= 2.0 ** -7 * scale
    assert float((outs[0] - outs[1]).abs().max()) <= 2.0 ** -7 * scale


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_pool_backward_kernels_with_ties(dtype):
    """MaxPool2d(2,2) and stride-1 MaxPool2d(k, pad k/2) backward alone, on coarsely quantised inputs
    (many exact ties): gradient goes to the first maximum in row-major window order, like torch."""
    from objectdetection_amd._lib import lib, check
    dt, tdt = (0, torch.float32) if dtype == 'f32' else (1, torch.bfloat16)
    g = torch.Generator().manual_seed(5)
    b, h, w, c = 2, 20, 20, 16
    x = (torch.randn(b, c, h, w, generator=g) * 2).round() / 2          # few levels: ties everywhere
    for k in (2, 5, 9, 13):
        xr = x.clone().requires_grad_(True)
        y = torch.nn.functional.max_pool2d(xr, 2, 2) if k == 2 else torch.nn.functional.max_pool2d(xr, k, 1, k // 2)
        gy = (torch.randn(y.shape, generator=g) * 4).round() / 4
        y.backward(gy)
        xg = x.permute(0, 2, 3, 1).contiguous().to(tdt).cuda()
        gyg = gy.permute(0, 2, 3, 1).contiguous().to(tdt).cuda()
        dx = torch.zeros_like(xg)
        if k == 2:
            check(lib.odamd_maxpool2_backward(xg.data_ptr(), c, gyg.data_ptr(), c, dx.data_ptr(), c, b, h, w, c, dt, None))
        else:
            check(lib.odamd_maxpool_s1_backward(xg.data_ptr(), c, gyg.data_ptr(), c, dx.data_ptr(), c, b, h, w, c, k, dt, None))
        torch.cuda.synchronize()
        got = dx.float().cpu().permute(0, 3, 1, 2)
        ref = xr.grad
        tol = 1e-6 if dt == 0 else 1e-2          # bf16 rounds the accumulated sums
        assert float((got - ref).abs().max()) <= tol * float(ref.abs().max()), k


@pytest.mark.parametrize('tag', ['cbs_k3', 'elan', 'mp', 'sppcspc'])
def test_standalone_block_trains_through_autograd(golden, tag):
    """Any block is an ordinary autograd module (model/layer.py:12-16): module(x) in train mode, backward of a
    scalar of the output, then x.grad, every param.grad and the BN buffers against the reference fixtures."""
    g = golden('blocks.npz')
    make, xn, shape = _block_cases()[tag]
    mod = make()
    patterns.fill_state_dict(mod)
    mod = mod.cuda().train()
    x = patterns.wave(xn, shape, 1.0).cuda().requires_grad_(True)
    y = mod(x)
    assert y.requires_grad and rel_err(y.detach().cpu().numpy(), g[f'{tag}.train']) < 1e-4
    gy = patterns.wave(tag + '.gy', tuple(y.shape), 1.0).cuda()
    (y * gy).sum().backward()
    torch.cuda.synchronize()
    assert rel_err(x.grad.cpu().numpy(), g[f'{tag}.dx']) < 1e-4
    for n, p in mod.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), g[f'{tag}.grad.{n}']) < 2e-4, n
    for n, b in mod.named_buffers():
        if b.dtype.is_floating_point:
            assert rel_err(b.cpu().numpy(), g[f'{tag}.buf.{n}']) < 1e-5, n
        else:
            assert int(b) == 1, n                      # num_batches_tracked advanced once
    # eval mode afterwards runs the inference plan on the updated running statistics
    mod.eval()
    with torch.no_grad():
        assert torch.isfinite(mod(x.detach())).all()


@pytest.mark.parametrize('kind,kw,shape', [
    ('elan', dict(in_=24, n=1, config=[13, 25, 24, 52, 37]), (2, 24, 12, 12)),
    ('elan_h', dict(in_=40, config=[21, 19, 11, 13, 9, 10, 30]), (2, 40, 8, 8)),
    ('mp', dict(in_=24, config=[13, 17, 21]), (2, 24, 12, 12)),
    ('sppcspc', dict(in_=40, config=[21, 19, 22, 18, 26, 23, 35]), (2, 40, 20, 20))])
def test_pruned_block_train_step(te, kind, kw, shape):
    """Training step of blocks built from a pruned width list (train_class.py:367-386 rebuilds the network from one
    in order to train it): irregular widths put zero-padded gaps into the concat buffers.  Forward, dx, every
    parameter gradient and the BN buffers against float32 CPU autograd, 1e-4 (weight gradients 2e-4)."""
    from objectdetection_amd.model import layer
    torch.manual_seed(41)
    mod = getattr(layer, kind)(**kw)
    g = torch.Generator().manual_seed(42)
    with torch.no_grad():
        for m in mod.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    x = torch.randn(shape, generator=g)
    mod.train()
    from oracle import network
    sd0 = {('.' + k2): v.detach().clone() for k2, v in mod.state_dict().items()}
    with torch.no_grad():
        y0 = getattr(network.Net(sd0, training=True), kind)('', x)
    gy = torch.randn(y0.shape, generator=g)
    sd_after = {}
    y_ref, dx_ref, grads = _oracle_block_train(kind, mod, x, gy)
    # running buffers after one step, from a second oracle pass on a copy of the state
    sd1 = {('.' + k2): v.detach().clone() for k2, v in mod.state_dict().items()}
    with torch.no_grad():
        getattr(network.Net(sd1, training=True), kind)('', x)
    mod = mod.cuda().train()
    state = te.FlatState(mod)
    plan = te.TrainPlan(mod, state, shape[0], shape[2], block_input=shape[1:])
    y = plan.forward(x.cuda())
    assert rel_err(y.cpu().numpy(), y_ref.numpy()) < 1e-4
    plan.gout_nchw.copy_(gy.cuda())
    state.grad.zero_()
    plan.backward(x.cuda())
    torch.cuda.synchronize()
    assert rel_err(plan.dx_nchw.cpu().numpy(), dx_ref.numpy()) < 1e-4
    for n, p in mod.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), grads[n].numpy()) < 2e-4, n
    for n, b in mod.named_buffers():
        if b.dtype.is_floating_point:
            assert rel_err(b.cpu().numpy(), sd1['.' + n].numpy()) < 1e-5, n


def test_pruned_network_train_step(te, golden):
    """One training step of a whole pruned network (the width list of tests/golden/pruned.npz): losses against
    the CPU oracle, gradients finite and non-zero for every parameter, in f32 and bf16."""
    from objectdetection_amd.train_class import train_class, synthetic_dataset
    from oracle import network
    import argparse
    cfg = golden('pruned.npz')['cfg'].tolist()
    for amp, tol in ((False, 2e-3), (True, 5e-2)):
        a = patterns.default_args(input_size=128, output_class=3)
        a.__dict__.update(device='cuda', batch=4, epoch=1, lr_start=0.001, lr_end_ratio=0.1, lr_end_epoch=10, warmup_ratio=0.01,
                          regularization='L2', r_value=0.0005, ema=True, distributed=False, local_rank=0, weight_path='',
                          weight_again=True, print_info=False, synthetic_images=8, save_path=None, amp=amp)
        torch.manual_seed(9)
        tc = train_class(a)
        from objectdetection_amd.model.yolov7 import yolov7
        tc.model = yolov7(a, config=cfg).cuda()
        tc.model.train_dtype = 'bf16' if amp else 'f32'
        tc.model_dict['model'] = tc.model
        tc.optimizer, tc.optimizer_adjust = tc.optimizer_load()
        ds = synthetic_dataset(a, 8)
        image, screen, expend, _ = ds.collate_fn([ds[i] for i in range(4)])
        sd = {k: v.detach().cpu().clone() for k, v in tc.model.state_dict().items()}
        with torch.no_grad():
            ref_pred = network.yolov7_forward(sd, image, a, training=True)
            ref = [float(v) for v in otrain.detection_loss(ref_pred, screen, expend, a.label_smooth, a.output_class)]
        tc.model.train()
        losses = tc.train_step(image, screen, expend).cpu().numpy()
        np.testing.assert_allclose(losses, ref, rtol=tol)
        st = tc.model.train_state()
        assert torch.isfinite(st.grad).all() and torch.isfinite(st.param).all()
        for n, p in tc.model.named_parameters():
            assert float(p.grad.abs().sum()) > 0, n
        again = tc.train_step(image, screen, expend).cpu().numpy()
        assert np.isfinite(again).all()
