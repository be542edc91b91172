"""Static execution plans for the YOLOv7 blocks on libodamd.so.

A module tree (model/layer.py, model/yolov7.py mirrors) is lowered ONCE per (input shape,
compute dtype) into a flat list of kernel launches over pre-allocated NHWC buffers:

* activations are NHWC, bf16 or f32; every `concat` of the reference (model/layer.py:25-27)
  becomes producers writing at channel offsets of one buffer, so no concat kernel exists;
* BatchNorm (eval) is folded into a per-channel (scale, shift) pair applied with SiLU in the
  convolution epilogue (model/layer.py:12-16);
* MaxPool2d(2,2) in front of a 1x1 conv (mp, model/layer.py:140-141) is applied while the
  convolution stages its input;
* the three heads store float32 NCHW, which the reference's raw reshape (model/layer.py:235)
  reinterprets as (B, A, H, W, K); decode (+ deploy sigmoid) reads that layout directly.

The plan only enqueues work on torch's current HIP stream, so a whole forward can be captured
in a HIP graph.  There is no fallback: every op is a call into the C ABI.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import lib, check

F32, BF16 = 0, 1
FUSE_MP = True                # both 1x1 convolutions of an mp block in one launch (bf16 plans)
FUSE_FIRST2 = True     # bf16 plans run cbs0 + cbs1 as one launch when the widths are 16 and 32
SPLIT_MAP_DIV = 32     # a pipeline's second stream takes a plan's launches from the first map of input size / SPLIT_MAP_DIV on
FUSE_CHAIN = True      # two chained 3x3 convolutions with 32 channels between and after them in one launch (bf16 plans)
SIDE_STREAMS = True    # side branches of a plan (heads, lateral 1x1 convolutions) run on a second HIP stream
_TORCH_DT = {0: torch.float32, 1: torch.bfloat16}
CH_ALIGN = 16            # channel granule of buffer slices in bf16 plans (f32 plans use 8)

_compute_dtype = None    # None: follow the parameters' dtype

# Kernels write parameters and BatchNorm running buffers through raw pointers (the optimizer step, the
# train-mode forward), which torch's per-tensor `_version` counters do not see.  Every such write bumps
# this generation; cached inference plans compare it (together with the `_version`s) before they run and
# re-fold their weights when it moved.
_weights_generation = [0]


def touch_weights():
    """Tell cached inference plans that parameters or BN buffers were modified behind torch's back."""
    _weights_generation[0] += 1


def set_compute_dtype(name):
    """Force the arithmetic type of all plans: 'bf16', 'f32' or None (follow parameter dtype:
    float32 parameters -> f32, half/bfloat16 parameters -> bf16)."""
    global _compute_dtype
    _compute_dtype = {None: None, 'bf16': BF16, 'f32': F32, 'fp32': F32}[name]


def _round_up(v, m):
    return (v + m - 1) // m * m


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Act:
    """Symbolic activation: channels [c0, c0+cphys) of NHWC buffer `buf`.

    `segments` lists (physical offset from c0, logical channel count) runs; gaps between runs
    are zero padding that keeps every run 16-channel aligned (pruned, irregular widths).
    """

    def __init__(self, buf, c0, segments, h, w, align=CH_ALIGN):
        self.buf, self.c0, self.segments, self.h, self.w, self.align = buf, c0, segments, h, w, align

    @property
    def c(self):
        return sum(n for _, n in self.segments)

    @property
    def cphys(self):
        off, n = self.segments[-1]
        return _round_up(off + n, self.align)


class PlanBuilder:
    def __init__(self, batch, dtype, device):
        self.batch, self.dtype, self.device = batch, dtype, device
        self.align = CH_ALIGN if dtype == BF16 else 8     # one MFMA k-step: 16 bf16 or 8 f32 channels
        self.buffers = []      # (cphys, h, w)
        self.ops = []          # dicts, resolved to launches by Plan
        self.convs = []        # ops that own packed weights
        self._side = 0         # > 0: ops emitted now belong to a side branch (see side_branch)

    def side_branch(self):
        """Context manager: the ops emitted inside do not feed the ops that follow them directly (a branch whose
        result is only read further down, e.g. a head or the lateral 1x1 of the neck).  An inference plan runs them
        on a second HIP stream, beside the launches that follow in plan order, and joins before the first reader."""
        pb = self

        class _Ctx:
            def __enter__(self):
                pb._side += 1

            def __exit__(self, *exc):
                pb._side -= 1
        return _Ctx()

    # ---- buffers -----------------------------------------------------------------------
    def buffer(self, widths, h, w):
        """New NHWC buffer holding a concat of slices with the given logical widths.
        Returns (whole Act, [slice Acts])."""
        segs, off = [], 0
        for n in widths:
            segs.append((off, n))
            off += _round_up(n, self.align)
        self.buffers.append((off, h, w))
        bid = len(self.buffers) - 1
        whole = Act(bid, 0, segs, h, w, self.align)
        return whole, [Act(bid, o, [(0, n)], h, w, self.align) for o, n in segs]

    def single(self, c, h, w):
        return self.buffer([c], h, w)[0]

    # ---- ops ---------------------------------------------------------------------------
    def conv(self, mod, x, out=None, pool=False):
        """cbs block (model/layer.py:4-16); pool=True applies MaxPool2d(2,2) to x on load."""
        k, s = mod.conv.kernel_size[0], mod.conv.stride[0]
        h, w = (x.h // 2, x.w // 2) if pool else (x.h, x.w)
        ho, wo = (h + 2 * ((k - 1) // 2) - k) // s + 1, (w + 2 * ((k - 1) // 2) - k) // s + 1
        cout = mod.conv.out_channels
        if out is None:
            out = self.single(cout, ho, wo)
        assert out.c == cout and out.h == ho and out.w == wo and x.c == mod.conv.in_channels
        op = dict(kind='conv', mod=mod, x=x, y=out, k=k, s=s, pool=pool, act=1, side=self._side > 0)
        self.ops.append(op)
        self.convs.append(op)
        return out

    def first(self, mod, image_hw):
        """cbs0 straight from the float32 NCHW image (3 input channels)."""
        h, w = image_hw
        out = self.single(mod.conv.out_channels, h, w)
        op = dict(kind='first', mod=mod, y=out)
        self.ops.append(op)
        self.convs.append(op)
        return out

    def head(self, mod, x):
        """head.output: 1x1 conv + bias, float32 NCHW (model/layer.py:232-235)."""
        op = dict(kind='head', mod=mod, x=x, k=1, s=1, pool=False, act=0, cout=mod.output.out_channels,
                  side=self._side > 0)
        self.ops.append(op)
        self.convs.append(op)
        return op

    def maxpool2(self, x, out=None):
        out = out or self.single(x.c, x.h // 2, x.w // 2)
        self.ops.append(dict(kind='maxpool2', x=x, y=out))
        return out

    def spp(self, x, outs):
        self.ops.append(dict(kind='spp', x=x, ys=outs))

    def upsample2(self, x, out=None):
        out = out or self.single(x.c, 2 * x.h, 2 * x.w)
        self.ops.append(dict(kind='upsample2', x=x, y=out))
        return out

    def copy(self, x, out):
        """Plain channel-slice copy (only used by a stand-alone `concat` module)."""
        self.ops.append(dict(kind='copy', x=x, y=out))
        return out

    def from_nchw(self, c, h, w):
        out = self.single(c, h, w)
        self.ops.append(dict(kind='from_nchw', y=out))
        return out

    def to_nchw(self, x):
        self.ops.append(dict(kind='to_nchw', x=x))


_shared_streams = {}
# 'side' and 'back0' are never busy in the same run (a plan whose second half goes to 'back0' keeps its side branches on
# the main streams, training plans do not run beside a pipeline): one stream serves both, which keeps a process that uses
# the default stream, one pipeline and one training plan at exactly four streams = four hardware queues
_STREAM_ALIAS = {'side': 'back0'}


def shared_stream(device, role):
    """ONE HIP stream per (device, role) for the life of the process.  A process has few hardware queues (4 by default)
    and every stream beyond them shares a queue with another one, i.e. runs in order with it: a training plan whose
    weight-gradient stream lands on the queue of its main stream loses the overlap it was built for (measured: 16.8 ms
    against 14.9 ms per bf16 step once an inference pipeline and its plans had taken their streams).  Plans, pipelines
    and training plans therefore draw from the same small set: 'side' (side branches / weight gradients), 'net<i>',
    'back<i>' and 'post<i>' (detect_pipeline)."""
    dev = torch.device(device)
    role = _STREAM_ALIAS.get(role, role)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), role)
    st = _shared_streams.get(key)
    if st is None:
        st = _shared_streams[key] = torch.cuda.Stream(device=key[0])
    return st


def _fold_bn(bn):
    """Eval-mode BatchNorm as y*scale + shift (model/layer.py:9,14)."""
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    return scale, shift


def _pack_weight(w, x_act, dtype):
    """(cout, cin, k, k) -> [cout][ky][kx][cin_phys] in the compute dtype, zero columns in gaps."""
    cout, cin, k, _ = w.shape
    wt = w.detach().float().permute(0, 2, 3, 1)
    if x_act is None or (len(x_act.segments) == 1 and x_act.cphys == cin):
        packed = wt
    else:
        packed = torch.zeros((cout, k, k, x_act.cphys), dtype=torch.float32, device=w.device)
        src = 0
        for off, n in x_act.segments:
            packed[..., off:off + n] = wt[..., src:src + n]
            src += n
    return packed.to(_TORCH_DT[dtype]).contiguous()


def _pack_fragments(w, groups, chunk_bytes):
    """Plain packed weights (cout, k, k, cin_phys) bf16 -> the fragment-ordered layout of
    odamd_conv_weight_layout: [cout / 32G][chunk][G][tap][k-step][2][32][8], zero rows up to a multiple
    of 32 * G output channels."""
    cout, k, _, cin = w.shape
    taps, per = k * k, 32 * groups
    ncb = (cout + per - 1) // per
    nch, ksn = cin * 2 // chunk_bytes, chunk_bytes // 32
    if ncb * per != cout:
        w = torch.cat([w, torch.zeros((ncb * per - cout, k, k, cin), dtype=w.dtype, device=w.device)])
    v = w.reshape(ncb, groups, 32, taps, nch, ksn, 2, 8)          # (cb, g, r, tap, chunk, s, half, e)
    return v.permute(0, 4, 1, 3, 5, 6, 2, 7).contiguous().view(-1)


class Plan:
    """Lowered module: buffers allocated, descriptors built, ready to run."""

    def __init__(self, builder, finish=None):
        self.b = builder
        dt = _TORCH_DT[builder.dtype]
        self.bufs = None
        self.es = 2 if builder.dtype == BF16 else 4
        self.launches = []
        self.meta = {}          # launch index -> {kernel, flops, bytes, shape} (algorithmic work, for profiling)
        self.keep = []          # tensors the descriptors point into
        self.weight_version = None
        self.outputs = {}
        self.finish = finish
        self._fuse_sibling_convs()
        used = set()
        for op in builder.ops:      # buffers no launch touches any more (fused away) are not allocated
            parts = ([op['pooled'], op['plain']] if op['kind'] == 'mp2' else
                     [op['first'], op['second']] if op['kind'] == 'c3pair' else [op])
            for a in [q.get(n) for q in parts for n in ('x', 'y')] + list(op.get('ys', [])):
                if a is not None and not (op['kind'] == 'first2' and a is op.get('x')):
                    used.add(a.buf)
        self.bufs = [torch.zeros((builder.batch, h, w, c) if i in used else (0, h, w, c), dtype=dt, device=builder.device)
                     for i, (c, h, w) in enumerate(builder.buffers)]
        self._build()

    def _fuse_sibling_convs(self):
        """Two consecutive 1x1 stride-1 cbs blocks that read the same activation and write adjacent
        channel ranges of one buffer (cbs0/cbs1 of elan and elan_h) become ONE launch with the
        weights stacked: the input is read from HBM once instead of twice."""
        ops, out, i = self.b.ops, [], 0
        while i < len(ops):
            a = ops[i]
            b = ops[i + 1] if i + 1 < len(ops) else None
            if (b is not None and a['kind'] == 'conv' and b['kind'] == 'conv' and a['x'] is b['x']
                    and a.get('side') == b.get('side')
                    and a['k'] == 1 and b['k'] == 1 and a['s'] == 1 and b['s'] == 1 and not a['pool'] and not b['pool']
                    and a['y'].buf == b['y'].buf and a['y'].c == a['y'].cphys and len(b['y'].segments) == 1
                    and b['y'].c0 == a['y'].c0 + a['y'].cphys):
                ya, yb = a['y'], b['y']
                f = dict(a)
                f['mods'] = [a['mod'], b['mod']]
                f['y'] = Act(ya.buf, ya.c0, [(0, ya.c + yb.c)], ya.h, ya.w, ya.align)
                out.append(f)
                i += 2
            else:
                out.append(a)
                i += 1
        out = self._fuse_first_two(out)
        out = self._fuse_mp_pairs(out)
        out = self._fuse_chain_pairs(out)
        self.b.ops = out
        self.b.convs = []
        for op in out:
            if op['kind'] in ('conv', 'head', 'first', 'first2'):
                self.b.convs.append(op)
            elif op['kind'] == 'mp2':
                self.b.convs += [op['pooled'], op['plain']]
            elif op['kind'] == 'c3pair':
                self.b.convs += [op['first'], op['second']]

    def _fuse_mp_pairs(self, ops):
        """cbs1 (on the max-pooled input) and cbs2 (on the input itself) of an mp block (model/layer.py:139-145) read
        the same tensor: in bf16 plans they become ONE launch (odamd_conv_mp_forward) where the library has the fused
        form for their shapes (decided in _build, which falls back to two launches otherwise)."""
        if not FUSE_MP or self.b.dtype != BF16:
            return ops
        out, i = [], 0
        while i < len(ops):
            a = ops[i]
            b = ops[i + 1] if i + 1 < len(ops) else None
            if (b is not None and a['kind'] == 'conv' and b['kind'] == 'conv' and a['x'] is b['x'] and 'mods' not in a
                    and not a.get('side') and not b.get('side')
                    and 'mods' not in b and a['k'] == 1 and b['k'] == 1 and a['s'] == 1 and b['s'] == 1 and a['pool']
                    and not b['pool'] and len(a['y'].segments) == 1 and len(b['y'].segments) == 1
                    and a['y'].c == a['y'].cphys and b['y'].c == b['y'].cphys):
                out.append(dict(kind='mp2', pooled=a, plain=b))
                i += 2
            else:
                out.append(a)
                i += 1
        return out

    def _fuse_chain_pairs(self, ops):
        """Two consecutive 3x3 stride-1 cbs blocks where the second reads what the first writes (the chains of elan,
        model/layer.py:74-81, and elan_h, model/layer.py:109-118) become ONE launch in bf16 plans
        (odamd_conv_chain_pair_forward) where the library has the fused form (decided in _build, which falls back to
        two launches otherwise): the second convolution reads the first one's tile from LDS.  Only chains that start
        from 32 channels: with 64 input channels the fused kernel measures behind the two launches (batch 64, 80 x 80:
        76 us against 53 us), with 32 slightly ahead (160 x 160: 150 against 159 us; 80 x 80: 41 against 45 us)."""
        if not FUSE_CHAIN or self.b.dtype != BF16:
            return ops
        out, i = [], 0
        while i < len(ops):
            a = ops[i]
            b = ops[i + 1] if i + 1 < len(ops) else None
            if (b is not None and a['kind'] == 'conv' and b['kind'] == 'conv' and b['x'] is a['y']
                    and 'mods' not in a and 'mods' not in b and not a.get('side') and not b.get('side')
                    and a['k'] == 3 and b['k'] == 3 and a['s'] == 1 and b['s'] == 1 and not a['pool'] and not b['pool']
                    and a['mod'].conv.out_channels == 32 and b['mod'].conv.out_channels == 32
                    and a['x'].cphys == 32 and len(a['y'].segments) == 1 and len(b['y'].segments) == 1
                    and a['y'].c == a['y'].cphys and b['y'].c == b['y'].cphys):
                out.append(dict(kind='c3pair', first=a, second=b))
                i += 2
            else:
                out.append(a)
                i += 1
        return out

    def _fuse_first_two(self, ops):
        """cbs0 (3 -> 16, stride 1) followed by cbs1 (16 -> 32, 3x3 stride 2) in bf16 becomes ONE launch
        (odamd_conv_first2_forward): the full-resolution 16-channel activation between them, the largest
        tensor of the network, is never written to HBM."""
        if not FUSE_FIRST2 or self.b.dtype != BF16 or len(ops) < 2:
            return ops
        a, b = ops[0], ops[1]
        if not (a['kind'] == 'first' and b['kind'] == 'conv' and b['x'] is a['y'] and 'mods' not in b
                and b['k'] == 3 and b['s'] == 2 and not b['pool'] and a['mod'].conv.out_channels == 16
                and b['mod'].conv.out_channels == 32 and len(b['y'].segments) == 1):
            return ops
        if any(op.get('x') is a['y'] or op.get('x') is not None and op['x'].buf == a['y'].buf for op in ops[2:]):
            return ops
        return [dict(kind='first2', mod=b['mod'], mod0=a['mod'], x=a['y'], y=b['y'])] + ops[2:]

    # ---- addressing ----------------------------------------------------------------------
    def _ptr(self, act):
        t = self.bufs[act.buf]
        return t.data_ptr() + act.c0 * self.es

    def _ld(self, act):
        return self.bufs[act.buf].shape[3]

    # ---- weights ---------------------------------------------------------------------------
    def _versions(self):
        v = _weights_generation[0] << 32
        for op in self.b.convs:
            for m in op.get('mods', [op['mod']]) + ([op['mod0']] if 'mod0' in op else []):
                if op['kind'] == 'head':
                    v += m.output.weight._version + m.output.bias._version
                else:
                    v += m.conv.weight._version + m.bn.weight._version + m.bn.bias._version
                    v += m.bn.running_mean._version + m.bn.running_var._version
        return v

    def refresh_weights(self):
        """(Re)pack weights and folded BN constants into the buffers the descriptors point at."""
        ver = self._versions()
        if ver == self.weight_version:
            return
        with torch.no_grad():
            for op in self.b.convs:
                m = op['mod']
                if op['kind'] == 'head':
                    w = _pack_weight(m.output.weight, op['x'], self.b.dtype)
                    op['w'].copy_(w.view(-1))
                    op['shift'].copy_(m.output.bias.detach().float())
                elif op['kind'] == 'first2':
                    m0 = op['mod0']
                    op['w0'].copy_(m0.conv.weight.detach().float().permute(0, 2, 3, 1).reshape(-1))
                    sc, sh = _fold_bn(m0.bn)
                    op['scale0'].copy_(sc)
                    op['shift0'].copy_(sh)
                    op['w'].copy_(_pack_weight(m.conv.weight, op['x'], self.b.dtype).view(-1))
                    sc, sh = _fold_bn(m.bn)
                    op['scale'].copy_(sc)
                    op['shift'].copy_(sh)
                elif op['kind'] == 'first':
                    op['w'].copy_(m.conv.weight.detach().float().permute(0, 2, 3, 1).reshape(-1))
                    sc, sh = _fold_bn(m.bn)
                    op['scale'].copy_(sc)
                    op['shift'].copy_(sh)
                else:
                    mods = op.get('mods', [m])
                    plain = torch.cat([_pack_weight(q.conv.weight, op['x'], self.b.dtype) for q in mods])
                    op['w'].copy_(_pack_fragments(plain, *op['frag']) if op.get('frag') else plain.view(-1))
                    folded = [_fold_bn(q.bn) for q in mods]
                    op['scale'].copy_(torch.cat([f[0] for f in folded]))
                    op['shift'].copy_(torch.cat([f[1] for f in folded]))
        self.weight_version = ver

    # ---- lowering ---------------------------------------------------------------------------
    def _conv_launch(self, op, allow_packed=True):
        """Descriptor, parameter buffers and profiling record of one conv / head op."""
        b, dev, dt = self.b, self.b.device, _TORCH_DT[self.b.dtype]
        B, kind = b.batch, op['kind']
        m, x = op['mod'], op['x']
        conv = m.output if kind == 'head' else m.conv
        cout, k = sum(q.conv.out_channels for q in op['mods']) if 'mods' in op else conv.out_channels, op['k']
        op['w'] = torch.empty(cout * k * k * x.cphys, dtype=dt, device=dev)
        op['shift'] = torch.empty(cout, dtype=torch.float32, device=dev)
        op['scale'] = None if kind == 'head' else torch.empty(cout, dtype=torch.float32, device=dev)
        d = _lib.ConvDesc()
        d.x, d.w = self._ptr(x), op['w'].data_ptr()
        d.scale = op['scale'].data_ptr() if op['scale'] is not None else None
        d.shift = op['shift'].data_ptr()
        d.batch, d.h, d.w_in, d.cin, d.cout = B, x.h, x.w, x.cphys, cout
        d.x_ld, d.ksize, d.stride, d.act = self._ld(x), k, op['s'], op['act']
        d.dtype, d.in_pool2 = b.dtype, int(op['pool'])
        if kind == 'head':
            out = torch.empty((B, cout, x.h, x.w), dtype=torch.float32, device=dev)
            op['out'] = out
            d.y, d.y_ld, d.out_nchw_f32 = out.data_ptr(), 0, 1
        else:
            d.y, d.y_ld, d.out_nchw_f32 = self._ptr(op['y']), self._ld(op['y']), 0
            # several layer classes want the weights in the order the kernel's matrix fragments are read
            g, cb = C.c_int(0), C.c_int(0)
            if allow_packed and lib.odamd_conv_weight_layout(C.byref(d), C.byref(g), C.byref(cb)):
                op['frag'] = (g.value, cb.value)
                per = 32 * g.value
                op['w'] = torch.empty((cout + per - 1) // per * per * k * k * x.cphys, dtype=dt, device=dev)
                d.w, d.w_packed = op['w'].data_ptr(), 1
        self.keep.append(d)
        ho = (d.h // (2 if op['pool'] else 1) + 2 * ((k - 1) // 2) - k) // op['s'] + 1
        wo = (d.w_in // (2 if op['pool'] else 1) + 2 * ((k - 1) // 2) - k) // op['s'] + 1
        cin = conv.in_channels
        meta = dict(
            kernel='conv%dx%d_s%d_%s' % (k, k, op['s'], 'c32' if cout <= 32 else 'c64' if cout <= 64 else 'c128'),
            flops=2.0 * B * ho * wo * cout * k * k * cin,
            bytes=float(B * x.h * x.w * cin * self.es + B * ho * wo * cout * (4 if kind == 'head' else self.es)
                        + cout * k * k * cin * self.es),
            shape='%dx%d %d->%d' % (x.h, x.w, cin, cout))
        return d, meta

    def _build(self):
        b, dev, dt = self.b, self.b.device, _TORCH_DT[self.b.dtype]
        B = b.batch
        self.launch_op = []     # launch index -> the op it belongs to
        for op in b.ops:
            kind = op['kind']
            if kind in ('conv', 'head'):
                d, meta = self._conv_launch(op)
                self.launches.append((lib.odamd_conv_forward, [C.byref(d)], 'conv', None))
                self.meta[len(self.launches) - 1] = meta
            elif kind == 'mp2':
                da, ma = self._conv_launch(op['pooled'], allow_packed=False)
                db, mb = self._conv_launch(op['plain'], allow_packed=False)
                if lib.odamd_conv_mp_supported(C.byref(da), C.byref(db)):
                    self.launches.append((lib.odamd_conv_mp_forward, [C.byref(da), C.byref(db)], 'conv_mp', None))
                    x = op['plain']['x']
                    self.meta[len(self.launches) - 1] = dict(      # the input counts once: that is the point of the fusion
                        kernel='mp_1x1_pair', flops=ma['flops'] + mb['flops'],
                        bytes=ma['bytes'] + mb['bytes'] - float(B * x.h * x.w * op['plain']['mod'].conv.in_channels * self.es),
                        shape=mb['shape'] + '+' + ma['shape'].split(' ')[1] + ' pooled')
                else:
                    # two launches: the pooled branch is a side branch (only the block's consumer reads its result)
                    op['pooled']['side'] = True
                    for d, meta, sub in ((da, ma, op['pooled']), (db, mb, op['plain'])):
                        self.launches.append((lib.odamd_conv_forward, [C.byref(d)], 'conv', None))
                        self.meta[len(self.launches) - 1] = meta
                        self.launch_op.append(sub)
            elif kind == 'c3pair':
                da, ma = self._conv_launch(op['first'], allow_packed=False)
                db, mb = self._conv_launch(op['second'], allow_packed=False)
                for d, sub in ((da, op['first']), (db, op['second'])):     # whole tensor as one block of fragments
                    sub['frag'] = (1, sub['x'].cphys * self.es)
                    d.w_packed = 1
                if lib.odamd_conv_chain_pair_supported(C.byref(da), C.byref(db)):
                    self.launches.append((lib.odamd_conv_chain_pair_forward, [C.byref(da), C.byref(db)], 'conv_chain_pair', None))
                    y = op['first']['y']
                    self.meta[len(self.launches) - 1] = dict(      # the tensor between the two is written, not read back
                        kernel='conv3x3_pair', flops=ma['flops'] + mb['flops'],
                        bytes=ma['bytes'] + mb['bytes'] - float(B * y.h * y.w * y.c * self.es),
                        shape=ma['shape'] + '->' + mb['shape'].split('>')[1])
                else:
                    for sub in (op['first'], op['second']):
                        sub.pop('frag')
                        d, meta = self._conv_launch(sub)
                        self.launches.append((lib.odamd_conv_forward, [C.byref(d)], 'conv', None))
                        self.meta[len(self.launches) - 1] = meta
                        self.launch_op.append(sub)
            elif kind == 'first':
                m, y = op['mod'], op['y']
                cout = m.conv.out_channels
                op['w'] = torch.empty(cout * 27, dtype=torch.float32, device=dev)
                op['scale'] = torch.empty(cout, dtype=torch.float32, device=dev)
                op['shift'] = torch.empty(cout, dtype=torch.float32, device=dev)
                self.launches.append((lib.odamd_conv_first_forward,
                                      [0, op['w'].data_ptr(), op['scale'].data_ptr(), op['shift'].data_ptr(),
                                       self._ptr(y), B, y.h, y.w, cout, self._ld(y), 1, b.dtype], 'first', 0))
            elif kind == 'first2':
                y, x = op['y'], op['x']
                op['w0'] = torch.empty(16 * 27, dtype=torch.float32, device=dev)
                op['w'] = torch.empty(32 * 9 * 16, dtype=dt, device=dev)
                for n, c in (('scale0', 16), ('shift0', 16), ('scale', 32), ('shift', 32)):
                    op[n] = torch.empty(c, dtype=torch.float32, device=dev)
                self.launches.append((lib.odamd_conv_first2_forward,
                                      [0, op['w0'].data_ptr(), op['scale0'].data_ptr(), op['shift0'].data_ptr(),
                                       op['w'].data_ptr(), op['scale'].data_ptr(), op['shift'].data_ptr(),
                                       self._ptr(y), B, x.h, x.w, self._ld(y)], 'first2', 0))
                self.meta[len(self.launches) - 1] = dict(
                    kernel='first2', flops=2.0 * B * (x.h * x.w * 16 * 27 + y.h * y.w * 32 * 9 * 16),
                    bytes=float(B * (x.h * x.w * 3 * 4 + y.h * y.w * 32 * self.es)), shape='%dx%d 3->16->32' % (x.h, x.w))
            elif kind == 'maxpool2':
                x, y = op['x'], op['y']
                self.launches.append((lib.odamd_maxpool2_forward,
                                      [self._ptr(x), self._ptr(y), B, x.h, x.w, x.cphys, self._ld(x), self._ld(y),
                                       b.dtype], 'maxpool2', None))
            elif kind == 'spp':
                x, ys = op['x'], op['ys']
                self.launches.append((lib.odamd_spp_pool3_forward,
                                      [self._ptr(x), self._ptr(ys[0]), self._ptr(ys[1]), self._ptr(ys[2]), B, x.h, x.w,
                                       x.cphys, self._ld(x), self._ld(ys[0]), b.dtype], 'spp', None))
            elif kind == 'upsample2':
                x, y = op['x'], op['y']
                self.launches.append((lib.odamd_upsample2_forward,
                                      [self._ptr(x), self._ptr(y), B, x.h, x.w, x.cphys, self._ld(x), self._ld(y),
                                       b.dtype], 'upsample2', None))
            elif kind == 'copy':
                x, y = op['x'], op['y']
                self.launches.append((None, [x, y], 'copy', None))
            elif kind == 'from_nchw':
                y = op['y']
                self.launches.append((lib.odamd_nchw_to_nhwc,
                                      [0, self._ptr(y), B, y.c, y.h, y.w, y.c, self._ld(y), b.dtype], 'from_nchw', 0))
            elif kind == 'to_nchw':
                x = op['x']
                out = torch.empty((B, x.c, x.h, x.w), dtype=torch.float32, device=dev)
                self.outputs['nchw'] = out
                dst = 0
                for off, n in x.segments:   # one launch per concat slice (gaps are padding)
                    self.launches.append((lib.odamd_nhwc_to_nchw,
                                          [self._ptr(x) + off * self.es, out.data_ptr() + dst * x.h * x.w * 4, B, n,
                                           x.h, x.w, self._ld(x), x.c, b.dtype], 'to_nchw', None))
                    dst += n
            else:
                raise AssertionError(kind)
            self.launch_op += [op] * (len(self.launches) - len(self.launch_op))
        self._plan_streams()

    def _plan_streams(self):
        """Side branches (PlanBuilder.side_branch) run on a second stream.  Per launch: 0 = main stream, 1 = side stream;
        `fork[i]`: the side stream first waits for everything enqueued on the main stream so far; `join[i]`: the main
        stream first waits for the side stream (launch i reads what a side launch wrote).  The side stream is joined at
        the end of a run as well, so consecutive runs of one plan never overlap."""
        n = len(self.launches)
        self.lane = [1 if SIDE_STREAMS and self.launch_op[i].get('side') else 0 for i in range(n)]
        self.fork, self.join = [False] * n, [False] * n
        pending = []          # (buf, lo, hi) ranges written by side launches that the main stream has not joined
        prev = 0
        for i in range(n):
            op = self.launch_op[i]
            if self.lane[i]:
                self.fork[i] = prev == 0
                if op['kind'] != 'head':
                    y = op['y']
                    pending.append((y.buf, y.c0, y.c0 + y.cphys))
            else:
                reads = [op[k] for k in ('x',) if op.get(k) is not None]
                if op['kind'] == 'mp2':
                    reads = [op['plain']['x']]
                if op['kind'] == 'c3pair':
                    reads = [op['first']['x']]
                writes = [op[k] for k in ('y',) if op.get(k) is not None] + list(op.get('ys', []))
                if op['kind'] == 'mp2':
                    writes = [op['pooled']['y'], op['plain']['y']]
                if op['kind'] == 'c3pair':
                    writes = [op['first']['y'], op['second']['y']]
                for a in reads + writes:
                    if any(bf == a.buf and lo < a.c0 + a.cphys and a.c0 < hi for bf, lo, hi in pending):
                        self.join[i] = True
                        pending = []
                        break
            prev = self.lane[i]
        self.has_side = any(self.lane)
        self._split_event = None
        self._events = []

    def first_launch_on_maps_up_to(self, h):
        """Index of the first launch whose input map is at most `h` rows high (None if there is none)."""
        for i, op in enumerate(self.launch_op):
            sub = op.get('plain') or op.get('first') or op
            x = sub.get('x')
            if x is not None and x.h <= h:
                return i
        return None

    def run(self, image, back=None, split=None):
        """Enqueue the whole plan on the current stream. `image`: contiguous float32 NCHW.
        `back`, `split`: launches from index `split` on go to the stream `back`, which first waits for the launches
        before them (a pipeline runs the small-map half of one batch beside the large-map half of the next, see
        predict.detect_pipeline); the caller orders later work after `back`.  Side branches stay on the main streams then
        (a process has four hardware queues)."""
        self.refresh_weights()
        main = torch.cuda.current_stream()
        st = C.c_void_p(main.cuda_stream)
        side = side_st = None
        if back is None or split is None or not 0 < split < len(self.launches):
            back = split = None
        use_side = self.has_side and back is None
        if use_side:
            side = shared_stream(main.device, 'side')
            side_st = C.c_void_p(side.cuda_stream)
        img = image.data_ptr()
        nev, side_busy = 0, False
        for i, (fn, args, name, image_slot) in enumerate(self.launches):
            if i == split:
                if self._split_event is None:
                    self._split_event = torch.cuda.Event()
                self._split_event.record(main)
                back.wait_event(self._split_event)
                main, st = back, C.c_void_p(back.cuda_stream)
            if name == 'copy':
                x, y = args
                with torch.cuda.stream(main):
                    self.bufs[y.buf][..., y.c0:y.c0 + x.cphys].copy_(self.bufs[x.buf][..., x.c0:x.c0 + x.cphys])
                continue
            if image_slot is not None:
                args[image_slot] = img
            if not use_side:
                rc = fn(*args, st)
                if rc:
                    m = self.meta.get(i, {})
                    check(rc, f"{name} #{i} {m.get('kernel', '')} {m.get('shape', '')}")
                continue
            if self.fork[i] or (self.join[i] and side_busy):
                if nev == len(self._events):
                    self._events.append(torch.cuda.Event())
                ev = self._events[nev]
                nev += 1
                if self.fork[i]:
                    ev.record(main)
                    side.wait_event(ev)
                else:
                    ev.record(side)
                    main.wait_event(ev)
                    side_busy = False
            if self.lane[i]:
                side_busy = True
            rc = fn(*args, side_st if self.lane[i] else st)
            if rc:
                m = self.meta.get(i, {})
                check(rc, f"{name} #{i} {m.get('kernel', '')} {m.get('shape', '')}")
        if side_busy:
            main.wait_stream(side)
        return self.finish(self, image) if self.finish else self.outputs.get('nchw')

    def profile(self, image, repeats=3):
        """Per-launch device time in ms (median of `repeats`), HIP events on the launch stream."""
        self.refresh_weights()
        st = _stream()
        img = image.data_ptr()
        times = [[] for _ in self.launches]
        for _ in range(repeats):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(self.launches) + 1)]
            evs[0].record()
            for i, (fn, args, name, image_slot) in enumerate(self.launches):
                if name != 'copy':
                    if image_slot is not None:
                        args[image_slot] = img
                    check(fn(*args, st), name)
                evs[i + 1].record()
            torch.cuda.synchronize()
            for i in range(len(self.launches)):
                times[i].append(evs[i].elapsed_time(evs[i + 1]))
        out = []
        for i, (fn, args, name, _) in enumerate(self.launches):
            m = dict(self.meta.get(i, {'kernel': name, 'flops': 0.0, 'bytes': 0.0, 'shape': ''}))
            m['ms'] = sorted(times[i])[len(times[i]) // 2]
            out.append(m)
        return out


def compute_dtype_for(module):
    if _compute_dtype is not None:
        return _compute_dtype
    for p in module.parameters():
        return F32 if p.dtype == torch.float32 else BF16
    return F32


def module_plans(module):
    """Per-module plan cache, kept ON the module (a reference cycle the collector can free, where a global
    table keyed by id(module) kept every module and all its plan buffers alive for the life of the process)."""
    return module.__dict__.setdefault('_odamd_plans', {})


def run_module(module, x, emit=None):
    """Generic forward of one block: NCHW tensor in, NCHW tensor out, through a cached plan."""
    if not x.is_cuda:
        raise RuntimeError('objectdetection_amd blocks run on the GPU only (no CPU fallback)')
    if module.training:
        return _run_module_train(module, x, emit)
    dtype = compute_dtype_for(module)
    plans = module_plans(module)
    key = (tuple(x.shape), dtype, x.device.index)
    plan = plans.get(key)
    if plan is None:
        b, c, h, w = x.shape
        pb = PlanBuilder(b, dtype, x.device)
        xin = pb.from_nchw(c, h, w)
        out = (emit or module._emit)(pb, xin)
        pb.to_nchw(out)
        plan = Plan(pb)
        plans[key] = plan
    y = plan.run(x.detach().float().contiguous())
    return y.clone() if x.dtype == torch.float32 else y.to(x.dtype)


class _BlockTrain(torch.autograd.Function):
    """Autograd node of one stand-alone block in train mode (batch-statistics BatchNorm): backward
    replays the block's reverse launch list, ADDS the parameter gradients to `param.grad` and returns
    the input gradient."""

    @staticmethod
    def forward(ctx, x, hook, plan):
        ctx.plan, ctx.x = plan, x
        return plan.forward(x).clone()

    @staticmethod
    def backward(ctx, grad_out):
        plan = ctx.plan
        plan.gout_nchw.copy_(grad_out)
        plan.backward(ctx.x)
        return plan.dx_nchw.clone(), None, None


def _run_module_train(module, x, emit=None):
    """Train-mode forward of a stand-alone block (model/layer.py:12-16: every block of the reference is an
    ordinary autograd module).  The block's parameters move into flat storage on first use (or stay in
    the flat storage of the network they already belong to); `param.grad` accumulates like torch's."""
    from . import train_engine
    if emit is not None:
        raise RuntimeError('train mode is not available for this module')
    state = None
    for p in module.parameters():
        state = train_engine.flat_state_of(p)
        break
    if state is None:
        # a block that trains on its own owns its flat storage (kept out of pickles by _Block.__getstate__)
        state = module.__dict__['_odamd_state'] = train_engine.FlatState(module)
    dtype = BF16 if getattr(module, 'train_dtype', 'f32') == 'bf16' else F32
    plans = module.__dict__.setdefault('_odamd_train_plans', {})
    key = (tuple(x.shape), dtype, x.device.index)
    plan = plans.get(key)
    if plan is None:
        b, c, h, w = x.shape
        plan = plans[key] = train_engine.TrainPlan(module, state, b, h, dtype, block_input=(c, h, w))
    hook = module.__dict__.get('_odamd_hook')
    if hook is None:
        hook = module.__dict__['_odamd_hook'] = torch.zeros(1, device=x.device, requires_grad=True)
    y = _BlockTrain.apply(x.detach().float().contiguous() if not x.requires_grad else x.float().contiguous(), hook, plan)
    return y if x.dtype == torch.float32 else y.to(x.dtype)
