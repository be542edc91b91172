"""GPU parity of the convolution blocks and the whole YOLOv7 forward (HIP, via the C ABI)
against the fp32 CPU oracle and the fixtures recorded from the reference.

Tolerances (stated by the task): float32 path 1e-4, bf16 path 2e-2, both as
max|got - ref| / max|ref| over a tensor."""
import numpy as np
import pytest
import torch

from oracle import network, patterns

pytestmark = pytest.mark.gpu
TOL = {'f32': 1e-4, 'bf16': 2e-2}


def rel_err(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12)


def assert_elementwise(got, ref, tol):
    """Every element on its own: |got - ref| <= tol * |ref| + tol * max|ref| (the max-norm check alone
    would let small-magnitude outputs be arbitrarily wrong)."""
    ref = np.asarray(ref)
    np.testing.assert_allclose(np.asarray(got), ref, rtol=tol, atol=tol * float(np.abs(ref).max()))


@pytest.fixture(scope='module')
def od():
    import objectdetection_amd.engine as engine
    from objectdetection_amd.model import layer, yolov7
    return engine, layer, yolov7


def _oracle_block(kind, mod, x):
    sd = {('.' + k): v.detach().cpu().float() for k, v in mod.state_dict().items()}
    net = network.Net(sd)
    if kind == 'cbs':
        return net.cbs('', x, mod.conv.stride[0])
    return getattr(net, kind)('', x)


BLOCK_CASES = [
    ('cbs', dict(in_=16, out_=24, kernel_size=1, stride=1), (2, 16, 12, 12)),
    ('cbs', dict(in_=32, out_=32, kernel_size=3, stride=1), (2, 32, 40, 40)),
    ('cbs', dict(in_=16, out_=32, kernel_size=3, stride=2), (3, 16, 24, 24)),
    ('cbs', dict(in_=64, out_=160, kernel_size=3, stride=2), (1, 64, 20, 20)),
    ('cbs', dict(in_=128, out_=64, kernel_size=1, stride=1), (2, 128, 33, 17)),
    ('cbs', dict(in_=48, out_=80, kernel_size=3, stride=1), (1, 48, 20, 20)),
    # deep layers on small maps: the LDS-DMA kernel (fragment-ordered weights), whole and ragged channel
    # blocks, ragged tiles, 1x1 over a pixel count that is no multiple of the tile
    ('cbs', dict(in_=128, out_=128, kernel_size=3, stride=1), (2, 128, 20, 20)),
    ('cbs', dict(in_=64, out_=64, kernel_size=3, stride=1), (2, 64, 40, 40)),
    ('cbs', dict(in_=80, out_=100, kernel_size=3, stride=1), (3, 80, 13, 27)),
    ('cbs', dict(in_=256, out_=128, kernel_size=3, stride=1), (1, 256, 20, 20)),
    ('cbs', dict(in_=512, out_=256, kernel_size=1, stride=1), (4, 512, 20, 20)),
    ('cbs', dict(in_=256, out_=136, kernel_size=1, stride=1), (3, 256, 19, 21)),
    ('cbs', dict(in_=1024, out_=256, kernel_size=1, stride=1), (2, 1024, 20, 20)),
    # 3x3 stride 2 with >= 64 channels on maps up to 160 x 160 (mp.cbs3): the LDS-DMA kernel with its even / odd column
    # planes; ragged tiles, odd output sizes, one and two channel blocks, a ragged channel block
    ('cbs', dict(in_=64, out_=64, kernel_size=3, stride=2), (2, 64, 80, 80)),
    ('cbs', dict(in_=128, out_=128, kernel_size=3, stride=2), (3, 128, 40, 40)),
    ('cbs', dict(in_=256, out_=256, kernel_size=3, stride=2), (1, 256, 40, 40)),
    ('cbs', dict(in_=64, out_=96, kernel_size=3, stride=2), (2, 64, 37, 53)),
    ('cbs', dict(in_=80, out_=200, kernel_size=3, stride=2), (1, 80, 26, 30)),
    # the same kernels on maps beyond 80 x 80 (stride 1, from 128 input channels) / beyond 160 x 160 inputs (stride 2, from
    # 128 output channels): the ELAN chains and the stem of the widest preset
    ('cbs', dict(in_=128, out_=128, kernel_size=3, stride=1), (1, 128, 100, 100)),
    ('cbs', dict(in_=160, out_=64, kernel_size=3, stride=1), (1, 160, 90, 97)),
    ('cbs', dict(in_=64, out_=128, kernel_size=3, stride=2), (1, 64, 200, 180)),
    # 256 / 512 / 1024 input channels, output channels in multiples of 128, pixel count a multiple of 64: the persistent
    # 1x1 kernel with register-resident weights (1..3 channel blocks, one and two channel groups per wave, fewer tiles
    # than workgroups, odd tile counts, several rounds per workgroup)
    ('cbs', dict(in_=256, out_=128, kernel_size=1, stride=1), (3, 256, 40, 40)),
    ('cbs', dict(in_=512, out_=512, kernel_size=1, stride=1), (5, 512, 32, 32)),
    ('cbs', dict(in_=512, out_=128, kernel_size=1, stride=1), (4, 512, 40, 40)),
    ('cbs', dict(in_=1024, out_=256, kernel_size=1, stride=1), (4, 1024, 32, 32)),
    ('cbs', dict(in_=256, out_=384, kernel_size=1, stride=1), (2, 256, 64, 64)),
    ('cbs', dict(in_=256, out_=256, kernel_size=1, stride=1), (16, 256, 80, 80)),
    # >= 32768 pixels: the persistent streaming 1x1 kernel (ragged last tile, 1..2 channel blocks, partial chunk)
    ('cbs', dict(in_=64, out_=32, kernel_size=1, stride=1), (2, 64, 128, 128)),
    ('cbs', dict(in_=128, out_=128, kernel_size=1, stride=1), (2, 128, 128, 130)),
    ('cbs', dict(in_=256, out_=256, kernel_size=1, stride=1), (1, 256, 192, 192)),
    ('cbs', dict(in_=32, out_=64, kernel_size=1, stride=1), (3, 32, 120, 120)),
    ('cbs', dict(in_=96, out_=40, kernel_size=1, stride=1), (1, 96, 200, 170)),
    # >= 2^22 (stride 1) / 2^20 (stride 2) output pixels, 32 input channels: the persistent 3x3 kernel (tile widths 32 / 8 / ragged 32, stride 1
    # and 2 with its even / odd column planes, ragged bottom tiles, item ranges that do not divide by 8)
    ('cbs', dict(in_=32, out_=32, kernel_size=3, stride=1), (41, 32, 320, 320)),
    ('cbs', dict(in_=32, out_=32, kernel_size=3, stride=1), (20, 32, 500, 424)),
    ('cbs', dict(in_=32, out_=32, kernel_size=3, stride=1), (35, 32, 348, 348)),
    ('cbs', dict(in_=32, out_=64, kernel_size=3, stride=2), (11, 32, 640, 640)),
    ('cbs', dict(in_=32, out_=64, kernel_size=3, stride=2), (6, 32, 870, 810)),
    ('elan', dict(in_=16, out_=32, n=1), (2, 16, 12, 12)),
    ('elan', dict(in_=32, out_=64, n=2), (1, 32, 16, 16)),
    ('elan_h', dict(in_=64, out_=32), (2, 64, 8, 8)),
    ('mp', dict(in_=32, out_=64), (2, 32, 16, 16)),
    # whole 8 x 16 patches, 32 / 64 / 128 input channels: both 1x1 convolutions of the block in one launch (bf16 plans);
    # the last shape has no whole patches and takes the two separate launches
    ('mp', dict(in_=128, out_=128), (2, 128, 40, 48)),
    ('mp', dict(in_=64, out_=128), (3, 64, 24, 32)),
    ('mp', dict(in_=128, out_=256), (2, 128, 16, 32)),
    ('mp', dict(in_=32, out_=64), (1, 32, 20, 20)),
    ('sppcspc', dict(in_=64, out_=32), (2, 64, 20, 20)),      # plane fits LDS: cascaded 5x5 pools
    ('sppcspc', dict(in_=32, out_=32), (1, 32, 48, 48)),      # larger plane: direct 13x13 walk
]


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('kind,kw,shape', BLOCK_CASES)
def test_block_forward(od, kind, kw, shape, dtype):
    engine, layer, _ = od
    engine.set_compute_dtype(dtype)
    try:
        mod = getattr(layer, kind)(**kw)
        patterns.fill_state_dict(mod)
        # white-noise input: a smooth sine input against sine weights cancels so strongly that the
        # output is ~100x smaller than its terms, which measures bf16 cancellation, not the kernel
        x = torch.from_numpy(np.random.default_rng(42).standard_normal(shape).astype(np.float32))
        ref = _oracle_block(kind, mod, x).numpy()
        got = mod.cuda().eval()(x.cuda()).cpu().numpy()
        assert got.shape == ref.shape
        assert rel_err(got, ref) < TOL[dtype], (kind, kw, rel_err(got, ref))
        assert_elementwise(got, ref, TOL[dtype])
    finally:
        engine.set_compute_dtype(None)


@pytest.mark.parametrize('in_,out_,shape', [(128, 128, (3, 128, 32, 48)), (64, 128, (2, 64, 16, 16)), (128, 256, (1, 128, 24, 32)),
                                            # 256 input channels: pairs of 8 x 8 patches, channels in chunks (an odd patch count,
                                            # a 40-wide map, enough pairs for several rounds per workgroup)
                                            (256, 256, (1, 256, 24, 24)), (256, 256, (3, 256, 40, 40)), (256, 256, (40, 256, 80, 80)),
                                            (128, 256, (5, 128, 40, 40))])
def test_fused_mp_pair_equals_separate_launches(od, in_, out_, shape):
    """odamd_conv_mp_forward (both 1x1 convolutions of an mp block in one launch) against the two odamd_conv_forward
    launches it replaces, on the same weights and input: bit for bit."""
    engine, layer, _ = od
    x = torch.from_numpy(np.random.default_rng(7).standard_normal(shape).astype(np.float32)).cuda()
    outs = []
    engine.set_compute_dtype('bf16')
    try:
        for fuse in (True, False):
            engine.FUSE_MP = fuse
            mod = layer.mp(in_, out_)
            patterns.fill_state_dict(mod)
            mod = mod.cuda().eval()
            outs.append(mod(x).cpu().numpy())
            plans = engine.module_plans(mod)
            kinds = [name for plan in plans.values() for _, _, name, _ in plan.launches]
            assert ('conv_mp' in kinds) == fuse, kinds
    finally:
        engine.FUSE_MP = True
        engine.set_compute_dtype(None)
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize('kind,kw,shape', [
    ('elan', dict(in_=64, out_=128, n=1), (2, 64, 48, 48)),        # 32 -> 32 -> 32, whole tiles
    ('elan', dict(in_=64, out_=128, n=1), (3, 64, 37, 53)),        # ragged tiles on both axes, one tile row taller than the run
    ('elan_h', dict(in_=128, out_=64), (2, 128, 40, 56)),          # 64 -> 32 -> 32 and 32 -> 32 -> 32
    ('elan_h', dict(in_=128, out_=64), (1, 128, 90, 21)),
    ('elan', dict(in_=64, out_=128, n=1), (70, 64, 80, 80)),       # enough tiles for vertical runs of several tiles
])
def test_fused_chain_pair_equals_separate_launches(od, kind, kw, shape):
    """odamd_conv_chain_pair_forward (two chained 3x3 convolutions in one launch, the tile between them in LDS) against
    the two odamd_conv_forward launches it replaces, on the same weights and input.  Both paths round the first
    convolution's result to bf16 before the second reads it; the float32 sums inside a convolution are taken in a
    different order by the different kernels, so single bf16 results may differ by one unit in the last place."""
    engine, layer, _ = od
    x = torch.from_numpy(np.random.default_rng(11).standard_normal(shape).astype(np.float32)).cuda()
    outs = []
    engine.set_compute_dtype('bf16')
    try:
        for fuse in (True, False):
            engine.FUSE_CHAIN = fuse
            mod = getattr(layer, kind)(**kw)
            patterns.fill_state_dict(mod)
            mod = mod.cuda().eval()
            outs.append(mod(x).float().cpu().numpy())
            plans = engine.module_plans(mod)
            kinds = [name for plan in plans.values() for _, _, name, _ in plan.launches]
            assert ('conv_chain_pair' in kinds) == fuse, kinds
            if fuse:      # the concat slices the pair wrote, against the f32 oracle of the block's input
                ref = _oracle_block(kind, mod.cpu(), x.cpu()).numpy()
                assert rel_err(outs[0], ref) < TOL['bf16']
                assert_elementwise(outs[0], ref, TOL['bf16'])
    finally:
        engine.FUSE_CHAIN = True
        engine.set_compute_dtype(None)
    a, b = outs
    scale = np.abs(b).max()
    assert np.abs(a - b).max() <= 2e-2 * scale, np.abs(a - b).max() / scale
    assert (a != b).mean() < 0.05, (a != b).mean()


@pytest.mark.parametrize('cin,shape', [(64, (3, 37, 53)), (32, (2, 16, 16)), (64, (5, 64, 48))])
def test_chain_pair_entry_point_against_two_launches(od, cin, shape):
    """odamd_conv_chain_pair_forward through the C ABI on slices of one wide NHWC buffer (the concat addressing of the
    ELAN blocks), 32 and 64 input channels, against odamd_conv_forward twice on the same buffers."""
    import ctypes as C
    from objectdetection_amd import _lib
    engine = od[0]
    B, H, W = shape
    ld = cin + 64 + 32
    rng = np.random.default_rng(3)
    w1 = torch.from_numpy(rng.standard_normal((32, 3, 3, cin)).astype(np.float32) * 0.08).cuda().bfloat16()
    w2 = torch.from_numpy(rng.standard_normal((32, 3, 3, 32)).astype(np.float32) * 0.1).cuda().bfloat16()
    sc = [torch.from_numpy(rng.uniform(0.5, 1.5, 32).astype(np.float32)).cuda() for _ in range(2)]
    sh = [torch.from_numpy(rng.uniform(-0.5, 0.5, 32).astype(np.float32)).cuda() for _ in range(2)]
    x = torch.from_numpy(rng.standard_normal((B, H, W, ld)).astype(np.float32)).cuda().bfloat16()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def desc(buf, c0, y0, w, i, ci, packed):
        d = _lib.ConvDesc()
        d.x, d.y = buf.data_ptr() + c0 * 2, buf.data_ptr() + y0 * 2
        d.w, d.scale, d.shift = w.data_ptr(), sc[i].data_ptr(), sh[i].data_ptr()
        d.batch, d.h, d.w_in, d.cin, d.cout, d.x_ld, d.y_ld = B, H, W, ci, 32, ld, ld
        d.ksize, d.stride, d.act, d.dtype, d.w_packed = 3, 1, 1, engine.BF16, packed
        return d
    got, ref = x.clone(), x.clone()
    f1, f2 = engine._pack_fragments(w1, 1, cin * 2), engine._pack_fragments(w2, 1, 64)
    da, db = desc(got, 0, cin, f1, 0, cin, 1), desc(got, cin, cin + 32, f2, 1, 32, 1)
    assert _lib.lib.odamd_conv_chain_pair_supported(C.byref(da), C.byref(db)) == 1
    _lib.check(_lib.lib.odamd_conv_chain_pair_forward(C.byref(da), C.byref(db), st), 'pair')
    for d in (desc(ref, 0, cin, w1, 0, cin, 0), desc(ref, cin, cin + 32, w2, 1, 32, 0)):
        _lib.check(_lib.lib.odamd_conv_forward(C.byref(d), st), 'conv')
    torch.cuda.synchronize()
    a, b = got.float().cpu().numpy(), ref.float().cpu().numpy()
    assert np.array_equal(a[..., :cin], b[..., :cin]) and np.array_equal(a[..., cin + 64:], b[..., cin + 64:])   # untouched
    for lo in (cin, cin + 32):
        u, v = a[..., lo:lo + 32], b[..., lo:lo + 32]
        assert np.abs(u - v).max() <= 2e-2 * np.abs(v).max(), (lo, np.abs(u - v).max())
        assert (u != v).mean() < 0.02, (lo, (u != v).mean())
    # a second convolution that does not read the first one's output is refused
    db.x = got.data_ptr()
    assert _lib.lib.odamd_conv_chain_pair_supported(C.byref(da), C.byref(db)) == 0
    assert _lib.lib.odamd_conv_chain_pair_forward(C.byref(da), C.byref(db), st) != 0


def test_block_golden_f32(od, golden):
    """Same blocks and inputs as the fixtures recorded from the reference."""
    engine, layer, _ = od
    g = golden('blocks.npz')
    cases = {'cbs_k1': (layer.cbs(16, 24, 1, 1), 'x16', (2, 16, 12, 12)),
             'cbs_k3': (layer.cbs(16, 16, 3, 1), 'x16', (2, 16, 12, 12)),
             'cbs_k3s2': (layer.cbs(16, 32, 3, 2), 'x16', (2, 16, 12, 12)),
             'elan': (layer.elan(16, 32, 1), 'x16', (2, 16, 12, 12)),
             'elan_n2': (layer.elan(16, 32, 2), 'x16', (2, 16, 12, 12)),
             'elan_h': (layer.elan_h(32, 16), 'x32', (2, 32, 8, 8)),
             'mp': (layer.mp(16, 32), 'x16', (2, 16, 12, 12)),
             'sppcspc': (layer.sppcspc(32, 16), 'xs', (2, 32, 20, 20))}
    for tag, (mod, xn, shape) in cases.items():
        patterns.fill_state_dict(mod)
        got = mod.cuda().eval()(patterns.wave(xn, shape, 1.0).cuda()).cpu().numpy()
        assert rel_err(got, g[f'{tag}.eval']) < TOL['f32'], (tag, rel_err(got, g[f'{tag}.eval']))
        assert_elementwise(got, g[f'{tag}.eval'], TOL['f32'])
    hd = layer.head(16, 12, 3)
    patterns.fill_state_dict(hd)
    got = hd.cuda().eval()(patterns.wave('x16', (2, 16, 12, 12), 1.0).cuda()).cpu().numpy()
    assert got.shape == g['head.eval'].shape
    assert rel_err(got, g['head.eval']) < TOL['f32']
    assert_elementwise(got, g['head.eval'], TOL['f32'])


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_whole_network_vs_reference_fixture(od, golden, dtype):
    engine, layer, ymod = od
    g = golden('net.npz')
    engine.set_compute_dtype(dtype)
    try:
        for tag, size, ncls, batch in (('s64', 64, 3, 2), ('s96', 96, 1, 1)):
            args = patterns.default_args(input_size=size, output_class=ncls)
            net = ymod.yolov7(args)
            patterns.fill_state_dict(net)
            net = net.cuda().eval()
            x = patterns.image_batch(batch, size).cuda()
            got = net(x).cpu().numpy()
            ref = g[f'{tag}.eval']
            assert got.shape == ref.shape
            # boxes (pixels) and logits have different scales: check them separately
            assert rel_err(got[..., :4], ref[..., :4]) < TOL[dtype], rel_err(got[..., :4], ref[..., :4])
            assert rel_err(got[..., 4:], ref[..., 4:]) < TOL[dtype], rel_err(got[..., 4:], ref[..., 4:])
            assert_elementwise(got[..., :4], ref[..., :4], TOL[dtype])
            assert_elementwise(got[..., 4:], ref[..., 4:], TOL[dtype])
            dep = layer.deploy(net)(x).cpu().numpy()
            assert rel_err(dep[..., 4:], g[f'{tag}.deploy'][..., 4:]) < TOL[dtype]
            assert_elementwise(dep[..., 4:], g[f'{tag}.deploy'][..., 4:], TOL[dtype])
    finally:
        engine.set_compute_dtype(None)


def test_network_640_bf16_vs_f32_oracle(od):
    """Full-size input (640^2, 80 classes, batch 4: every map has a multiple of 64 pixels, so the deep 1x1 layers take the
    register-resident-weights kernel on slices of the concat buffers): bf16 path against the CPU fp32 oracle."""
    engine, layer, ymod = od
    args = patterns.default_args(input_size=640, output_class=80)
    net = ymod.yolov7(args)
    patterns.fill_state_dict(net)
    x = patterns.image_batch(4, 640)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    with torch.no_grad():
        ref = network.yolov7_forward(sd, x, args).numpy()
    net = net.cuda().eval()
    for dtype in ('f32', 'bf16'):
        engine.set_compute_dtype(dtype)
        try:
            got = net(x.cuda()).cpu().numpy()
        finally:
            engine.set_compute_dtype(None)
        assert got.shape == (4, 25200, 85)
        assert rel_err(got[..., :4], ref[..., :4]) < TOL[dtype], (dtype, rel_err(got[..., :4], ref[..., :4]))
        assert rel_err(got[..., 4:], ref[..., 4:]) < TOL[dtype], (dtype, rel_err(got[..., 4:], ref[..., 4:]))
        assert_elementwise(got[..., :4], ref[..., :4], TOL[dtype])
        assert_elementwise(got[..., 4:], ref[..., 4:], TOL[dtype])


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('cout,h,w', [(16, 64, 64), (24, 37, 70), (40, 16, 130), (32, 50, 77), (64, 33, 130)])
def test_first_layer_entry_point(cout, h, w, dtype):
    """odamd_conv_first_forward alone (weights [cout][ky][kx][3]) against torch conv2d on random data."""
    from objectdetection_amd._lib import lib, check
    g = torch.Generator().manual_seed(cout + h)
    x = torch.rand(2, 3, h, w, generator=g)
    wt = torch.randn(cout, 3, 3, 3, generator=g) * 0.3
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.2
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), padding=1) * scale.double()[None, :, None, None] \
        + shift.double()[None, :, None, None]
    ref = torch.nn.functional.silu(ref).permute(0, 2, 3, 1)
    dt = 0 if dtype == 'f32' else 1
    ld = -(-cout // 16) * 16
    y = torch.zeros(2, h, w, ld, dtype=torch.float32 if dt == 0 else torch.bfloat16, device='cuda')
    xg, wg = x.cuda(), wt.permute(0, 2, 3, 1).contiguous().cuda()
    sg, hg = scale.cuda(), shift.cuda()
    check(lib.odamd_conv_first_forward(xg.data_ptr(), wg.data_ptr(), sg.data_ptr(), hg.data_ptr(), y.data_ptr(), 2, h, w,
                                       cout, ld, 1, dt, None))
    torch.cuda.synchronize()
    got = y[..., :cout].float().cpu().double()
    assert rel_err(got.numpy(), ref.numpy()) < (1e-5 if dt == 0 else 1.5e-2)
    assert float(y[..., cout:].float().abs().max()) == 0.0 if ld > cout else True


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('preset,size', [('s', 128), ('m', 64), ('l', 64)])
def test_every_layer_against_oracle_taps(od, dtype, preset, size):
    """Every cbs output inside the whole network (random weights and BN statistics) against the oracle's
    activation of the same name: a wrong layer cannot hide behind the heads.  All three width / depth
    presets of model/yolov7.py:9-12 (up to 2048 input channels in a 1x1 layer for 'l')."""
    engine, layer, ymod = od
    args = patterns.default_args(input_size=size, output_class=4)
    args.model_type = preset
    torch.manual_seed(11)
    net = ymod.yolov7(args)
    g = torch.Generator().manual_seed(12)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.bias.shape, generator=g) * 0.5 + 0.25)
    x = torch.rand(2, 3, size, size, generator=g)
    taps = {}
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    with torch.no_grad():
        network.yolov7_forward(sd, x, args, taps=taps)
    net = net.cuda().eval()
    names = {m: n for n, m in net.named_modules()}
    engine.set_compute_dtype(dtype)
    try:
        xg = x.cuda()
        net(xg)
        plan = net._plan(xg)
    finally:
        engine.set_compute_dtype(None)
    checked, worst = 0, 0.0
    for op in plan.b.ops:
        if op['kind'] not in ('conv', 'first', 'first2'):
            continue
        y = op['y']
        buf = plan.bufs[y.buf][..., y.c0:y.c0 + y.cphys].float().cpu()
        c0 = 0
        for m in op.get('mods', [op['mod']]):
            ref = taps[names[m]].permute(0, 2, 3, 1)
            got = buf[..., c0:c0 + ref.shape[3]]
            c0 += ref.shape[3]
            err = rel_err(got.numpy(), ref.numpy())
            worst = max(worst, err)
            assert err < TOL[dtype], (names[m], err)
            checked += 1
    assert checked >= {'s': 55, 'm': 60, 'l': 70}[preset], checked


@pytest.mark.parametrize('size', [64, 96, 160])
def test_fused_first_two_layers(od, size):
    """bf16 plans run cbs0 + cbs1 (model/yolov7.py:114-115) as one launch; against the two separate
    launches (same bf16 operands, different accumulation order inside the matrix cores) the result may
    differ by a bf16 rounding of the intermediate here and there, never more; sizes with whole and
    ragged output tiles, image borders included."""
    engine, layer, ymod = od
    args = patterns.default_args(input_size=size, output_class=3)
    torch.manual_seed(5)
    net = ymod.yolov7(args)
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.bias.shape, generator=g) * 0.5 + 0.25)
    net = net.cuda().eval()
    x = torch.rand(3, 3, size, size, generator=g).cuda()
    outs, kinds = [], []
    engine.set_compute_dtype('bf16')
    try:
        for fuse in (True, False):
            engine.FUSE_FIRST2 = fuse
            net._plans.clear()          # plans are cached per (shape, dtype): lower again
            net(x)
            plan = net._plan(x)
            kinds.append([op['kind'] for op in plan.b.ops[:2]])
            op = next(o for o in plan.b.ops if o['kind'] in ('first2', 'conv'))
            y = op['y']
            outs.append(plan.bufs[y.buf][..., y.c0:y.c0 + 32].clone())
    finally:
        engine.FUSE_FIRST2 = True
        engine.set_compute_dtype(None)
    assert kinds[0][0] == 'first2' and kinds[1] == ['first', 'conv'], kinds
    a, b = outs[0].float(), outs[1].float()
    assert (a - b).abs().max() <= 2.0 ** -7 * b.abs().max()
    assert (a != b).float().mean() < 0.02


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_pruned_config_network(od, dtype):
    """A pruned network (train_class.py:367-386 rebuilds yolov7 from a list of irregular cbs widths):
    every layer against the oracle.  Irregular widths exercise the zero-padded channel slices."""
    engine, layer, ymod = od
    args = patterns.default_args(input_size=64, output_class=3)
    torch.manual_seed(31)
    std = ymod.yolov7(args)
    widths = [m.conv.out_channels for m in std.modules() if isinstance(m, layer.cbs)]
    cfg = [max(5, w - w // 5 - (i % 3)) for i, w in enumerate(widths)]
    net = ymod.yolov7(args, config=cfg)
    got_widths = [m.conv.out_channels for m in net.modules() if isinstance(m, layer.cbs)]
    assert got_widths == cfg
    g = torch.Generator().manual_seed(32)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.bias.shape, generator=g) * 0.5 + 0.25)
    x = torch.rand(2, 3, 64, 64, generator=g)
    taps = {}
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    with torch.no_grad():
        ref = network.yolov7_forward(sd, x, args, taps=taps).numpy()
    net = net.cuda().eval()
    names = {m: n for n, m in net.named_modules()}
    engine.set_compute_dtype(dtype)
    try:
        got = net(x.cuda()).cpu().numpy()
        plan = net._plan(x.cuda())
    finally:
        engine.set_compute_dtype(None)
    assert rel_err(got[..., :4], ref[..., :4]) < TOL[dtype]
    assert rel_err(got[..., 4:], ref[..., 4:]) < TOL[dtype]
    checked = 0
    for op in plan.b.ops:
        if op['kind'] not in ('conv', 'first', 'first2'):
            continue
        y = op['y']
        buf = plan.bufs[y.buf][..., y.c0:y.c0 + y.cphys].float().cpu()
        mods = op.get('mods', [op['mod']])
        if len(mods) > 1 or len(y.segments) > 1:
            continue
        r = taps[names[mods[0]]].permute(0, 2, 3, 1)
        assert rel_err(buf[..., :r.shape[3]].numpy(), r.numpy()) < TOL[dtype], names[mods[0]]
        checked += 1
    assert checked >= 50


def test_pruned_network_vs_reference_fixture(od, golden):
    engine, layer, ymod = od
    g = golden('pruned.npz')
    args = patterns.default_args(input_size=64, output_class=3)
    net = ymod.yolov7(args, config=g['cfg'].tolist())
    patterns.fill_state_dict(net)
    got = net.cuda().eval()(patterns.image_batch(2, 64).cuda()).cpu().numpy()
    assert rel_err(got[..., :4], g['eval'][..., :4]) < TOL['f32']
    assert rel_err(got[..., 4:], g['eval'][..., 4:]) < TOL['f32']
    assert_elementwise(got[..., :4], g['eval'][..., :4], TOL['f32'])
    assert_elementwise(got[..., 4:], g['eval'][..., 4:], TOL['f32'])


@pytest.mark.parametrize('c,ld,dt', [(3, 8, 'f32'), (3, 8, 'bf16'), (8, 8, 'bf16'), (1, 8, 'f32'), (5, 16, 'bf16'), (3, 4, 'f32')])
def test_nchw_to_nhwc_layouts(c, ld, dt):
    """NCHW float32 planes -> NHWC rows of `ld` channels: the per-pixel kernel for at most eight planes into
    eight-channel pixels (padding channels written as zeros) and the tiled kernel for everything else."""
    from objectdetection_amd._lib import lib, check
    from objectdetection_amd.engine import F32, BF16
    b, h, w = 2, 37, 41
    x = torch.randn(b, c, h, w, generator=torch.Generator().manual_seed(c * 10 + ld)).cuda()
    tdt = torch.float32 if dt == 'f32' else torch.bfloat16
    y = torch.zeros(b, h, w, ld, dtype=tdt, device='cuda')
    check(lib.odamd_nchw_to_nhwc(x.data_ptr(), y.data_ptr(), b, c, h, w, c, ld, F32 if dt == 'f32' else BF16,
                                 torch.cuda.current_stream().cuda_stream), 'nchw_to_nhwc')
    torch.cuda.synchronize()
    ref = torch.zeros(b, h, w, ld, dtype=tdt, device='cuda')
    ref[..., :c] = x.permute(0, 2, 3, 1).to(tdt)
    assert torch.equal(y, ref)
