"""Training-step execution plan for the YOLOv7 mirror (forward in train mode, loss, backward,
Adam + EMA) on libodamd.so.

The same block-level `_emit` lowering as inference produces the op list; here every cbs becomes
raw convolution -> train-mode BatchNorm + SiLU, the tensors backward needs are kept, and a reverse
op list is generated from it (reference autograd of train_class.py:208-210).  Parameters live in
one flat float32 buffer (the modules' parameters are views into it), gradients in a second flat
buffer that is zeroed once per step and all-reduced in contiguous buckets, Adam/EMA state in
further flat buffers updated by one fused launch (train_class.py:158-167, :477-499).
"""
import ctypes as C
import math
import os
import sys
import weakref

import torch

from . import _lib, engine, postprocess as _post
from ._lib import lib, check
from .engine import F32, BF16, Act, PlanBuilder, _round_up, CH_ALIGN

BN_EPS, BN_MOMENTUM = 1e-3, 0.03

# id(parameter) -> (weak reference to the parameter, weak reference to the FlatState that owns its storage).
# The back-pointer must not live in Parameter.__dict__: torch pickles that dict, so torch.save(model) would drag
# the whole flat state (parameters, gradients, Adam moments, EMA copies) into every checkpoint.
_FLAT_OF = {}


def _register_flat(p, state):
    pid = id(p)

    def drop(ref, pid=pid):
        e = _FLAT_OF.get(pid)
        if e is not None and e[0] is ref:
            del _FLAT_OF[pid]
    _FLAT_OF[pid] = (weakref.ref(p, drop), weakref.ref(state))


def flat_state_of(p):
    """The live FlatState whose flat buffers hold parameter `p`, or None (never a stale one: the entry must
    name this very object and the state must still list it)."""
    e = _FLAT_OF.get(id(p))
    if e is None or e[0]() is not p:
        return None
    st = e[1]()
    return st if st is not None and id(p) in st.offsets else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class FlatState:
    """Flat float32 storage for parameters (+grad, Adam moments, EMA) and BN running buffers."""

    LAYOUT = 2      # version of the order below (an optimizer state saved under another order does not fit)

    def __init__(self, model):
        dev = next(model.parameters()).device
        params = [p for p in model.parameters()]
        # The two 1x1 cbs blocks at the head of an elan / elan_h block read the same tensor and write adjacent channel
        # ranges (model/layer.py:75-76, :110-111): with their tensors stored back to back (weights, BN weights, BN
        # biases, running means, running variances) the pair is ONE cbs of the summed width to every kernel of the
        # training step (see TrainPlan._fuse_siblings).
        self.sibling_pairs = sibling_pairs(model)
        first = {id(a.conv.weight): (a, b) for a, b in self.sibling_pairs}
        handled, order = set(), []
        for p in params:
            if id(p) in handled:
                continue
            if id(p) in first:
                a, b = first[id(p)]
                group = [a.conv.weight, b.conv.weight, a.bn.weight, b.bn.weight, a.bn.bias, b.bn.bias]
                order += group
                handled.update(id(q) for q in group)
            else:
                order.append(p)
        assert len(order) == len(params)
        params = order
        self.offsets, n = {}, 0
        for p in params:
            self.offsets[id(p)] = n
            n += _round_up(p.numel(), 4)            # keep every tensor 16-byte aligned
        self.n = n
        self.param = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p in params:
                o = self.offsets[id(p)]
                view = self.param[o:o + p.numel()].view(p.shape)
                view.copy_(p.data.float())
                p.data = view
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
                _register_flat(p, self)                # a block asked to train on its own finds its network's storage
        bufs = [b for b in model.buffers() if b.dtype.is_floating_point]
        firstb = {id(a.bn.running_mean): (a, b) for a, b in self.sibling_pairs}
        handled, order = set(), []
        for t in bufs:
            if id(t) in handled:
                continue
            if id(t) in firstb:
                a, b = firstb[id(t)]
                group = [a.bn.running_mean, b.bn.running_mean, a.bn.running_var, b.bn.running_var]
                order += group
                handled.update(id(q) for q in group)
            else:
                order.append(t)
        assert len(order) == len(bufs)
        bufs = order
        self.buf_offsets, nb = {}, 0
        for b in bufs:
            self.buf_offsets[id(b)] = nb
            nb += _round_up(b.numel(), 4)
        self.nb = nb
        self.buffers = torch.zeros(max(nb, 4), dtype=torch.float32, device=dev)
        with torch.no_grad():
            for name, b in list(model.named_buffers()):
                if not b.dtype.is_floating_point:
                    continue
                o = self.buf_offsets[id(b)]
                view = self.buffers[o:o + b.numel()].view(b.shape)
                view.copy_(b)
                mod = model
                *path, leaf = name.split('.')
                for part in path:
                    mod = getattr(mod, part)
                mod._buffers[leaf] = view
                self.buf_offsets[id(view)] = o
        self.param_names = {n: self.offsets[id(p)] for n, p in model.named_parameters()}
        self.buf_names = {n: self.buf_offsets[id(b)] for n, b in model.named_buffers() if b.dtype.is_floating_point}
        # integer buffers (BatchNorm2d.num_batches_tracked) become views of one counter vector, so a
        # train-mode forward advances all of them with one add
        ints = [(n, b) for n, b in model.named_buffers() if not b.dtype.is_floating_point]
        self.counters = torch.zeros(max(len(ints), 1), dtype=torch.int64, device=dev)
        with torch.no_grad():
            for i, (name, b) in enumerate(ints):
                view = self.counters[i:i + 1].view(b.shape)
                view.copy_(b)
                mod = model
                *path, leaf = name.split('.')
                for part in path:
                    mod = getattr(mod, part)
                mod._buffers[leaf] = view
        self.ema_param = None
        self.ema_buffers = None
        self.step = 0
        self.ema_updates = 0

    def enable_ema(self):
        self.ema_param = self.param.clone()
        self.ema_buffers = self.buffers.clone()

    def grad_ptr(self, p):
        return self.grad.data_ptr() + 4 * self.offsets[id(p)]


def sibling_pairs(model):
    """(cbs0, cbs1) of every elan / elan_h block whose two head convolutions can run as one: same input width, 1x1,
    widths that keep every tensor of the pair 16-byte aligned behind its sibling."""
    from .model import layer
    out = []
    for m in model.modules():
        if isinstance(m, (layer.elan, layer.elan_h)):
            a, b = m.cbs0, m.cbs1
            if (a.conv.in_channels == b.conv.in_channels and a.conv.kernel_size == (1, 1) and b.conv.kernel_size == (1, 1)
                    and a.conv.out_channels % 4 == 0 and a.conv.weight.numel() % 4 == 0):
                out.append((a, b))
    return out


def _gloo_with_gpu_tensor(t, group=None):
    import torch.distributed as dist
    return dist.is_initialized() and dist.get_backend(group) == 'gloo' and t.is_cuda


def broadcast_from_rank0(t, group=None, async_op=False):
    """DDP's broadcast of rank 0's tensors (train_class.py:38-40).  RCCL takes device tensors; gloo (CPU tests and the
    one-GPU rehearsal of the multi-rank path) is only ever given host tensors: its own staging of device
    tensors runs on internal streams and a helper thread, and a later blocking host-to-device copy of the
    same process (legacy default-stream semantics: it waits for every stream of the process) was seen to wait
    on that machinery forever when two ranks share one GPU.
    `async_op`: returns the collective's handle (its wait() orders the current stream behind it on RCCL and blocks
    the host on gloo); the host-staged form has completed on return and returns None."""
    import torch.distributed as dist
    if _gloo_with_gpu_tensor(t, group):
        torch.cuda.current_stream().synchronize()
        host = t.cpu()
        dist.broadcast(host, 0, group=group)
        t.copy_(host)
        torch.cuda.current_stream().synchronize()
        return None
    return dist.broadcast(t, 0, group=group, async_op=async_op) if async_op else dist.broadcast(t, 0, group=group)


class DataParallel:
    """The data-parallel wiring of train_class.py:37-40 (DistributedDataParallel, default arguments) over a FlatState:
      * construction: rank 0's parameters, BatchNorm statistics and counters replace everyone's;
      * every step: DDP's broadcast_buffers=True hands rank 0's buffers to all ranks before each forward.  Buffers
        only change inside a forward, so the broadcast for step i+1 is issued right after the forward of step i:
        the float32 statistics and the int64 counters are packed into one staging buffer (typed views of one
        storage cannot be module buffers: torch.save refuses them), ONE non-blocking collective moves it behind the
        backward pass, and `finish` waits for it together with the gradient buckets and unpacks;
      * the bucketed gradient all-reduce (`reducer`, started by the backward pass).
    """

    def __init__(self, state, n_buckets=3, group=None, broadcast_buffers=True):
        import torch.distributed as dist
        self.state, self.group, self.broadcast_buffers = state, group, broadcast_buffers
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        fbytes = state.buffers.numel() * 4
        self.stage = torch.zeros(fbytes + state.counters.numel() * 8, dtype=torch.uint8, device=state.param.device)
        self.stage_f = self.stage[:fbytes].view(torch.float32)
        self.stage_i = self.stage[fbytes:].view(torch.int64)
        self.reducer = GradReducer(state.grad, n_buckets=n_buckets, group=group)
        self.world = self.reducer.world
        self.pending, self.unpack = None, False
        self.collectives = 0            # collectives issued so far (tests and diagnostics)
        if self.world > 1:
            broadcast_from_rank0(state.param, group)
            self.collectives += 1
            self._send_buffers(False)
            self._take_buffers()

    def _send_buffers(self, async_op):
        if self.rank == 0:
            self.stage_f.copy_(self.state.buffers)
            self.stage_i.copy_(self.state.counters)
        self.pending = broadcast_from_rank0(self.stage, self.group, async_op=async_op)
        self.collectives += 1
        self.unpack = True

    def _take_buffers(self):
        if self.pending is not None:
            self.pending.wait()
            self.pending = None
        if self.unpack and self.rank != 0:
            self.state.buffers.copy_(self.stage_f)
            self.state.counters.copy_(self.stage_i)
        self.unpack = False

    def after_forward(self):
        if self.broadcast_buffers and self.world > 1:
            self._send_buffers(True)

    def finish(self):
        """Waits for the buffer broadcast and every gradient bucket; returns the sum -> mean factor."""
        self._take_buffers()
        self.collectives += len(self.reducer.handles)
        return self.reducer.finish()


class GradReducer:
    """Bucketed data-parallel gradient all-reduce over torch.distributed (RCCL on GPUs, gloo in the
    CPU tests).  The flat gradient buffer is cut into contiguous buckets; bucket i is started as
    soon as the backward pass reports that everything inside it is final (`ready`), so the
    collectives run behind the remaining backward kernels (DDP of train_class.py:37-40)."""

    def __init__(self, flat_grad, n_buckets=3, group=None):
        import torch.distributed as dist
        self.dist, self.group, self.flat = dist, group, flat_grad
        n = flat_grad.numel()
        per = -(-n // n_buckets)
        if per >= 65536:
            per = -(-per // 1024) * 1024                     # large buffers: 4 KiB aligned bucket starts
        self.bounds = [(i * per, min(n, (i + 1) * per)) for i in range(n_buckets) if i * per < n]
        self.handles = []
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.host_staged = _gloo_with_gpu_tensor(flat_grad, group)

    def bucket_of(self, offset):
        for i, (a, b) in enumerate(self.bounds):
            if a <= offset < b:
                return i
        return len(self.bounds) - 1

    def buckets_of(self, offset, numel):
        """Every bucket that holds a part of [offset, offset + numel): bucket bounds are multiples of 1024
        elements and cut through tensors, so a tensor can live in two (or more) buckets."""
        return [i for i, (a, b) in enumerate(self.bounds) if a < offset + max(numel, 1) and offset < b]

    def ready(self, i):
        if self.world > 1:
            a, b = self.bounds[i]
            if self.host_staged:
                # gloo is only ever given host tensors (see broadcast_from_rank0); RCCL, the production
                # backend, reduces the device buffer in place, asynchronously and stream ordered
                torch.cuda.current_stream().synchronize()
                host = self.flat[a:b].cpu()
                self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM, group=self.group)
                self.flat[a:b].copy_(host)
                return
            self.handles.append(self.dist.all_reduce(self.flat[a:b], op=self.dist.ReduceOp.SUM, group=self.group,
                                                     async_op=True))

    def finish(self):
        """Wait for all buckets; returns the factor that turns the sum into the mean."""
        for h in self.handles:
            h.wait()
        self.handles = []
        return 1.0 / self.world


def _pack_positives(screen_list, label_expend, dev):
    """Host lists of the reference's collate format -> device arrays (row, image, label row) per positive."""
    idx, img = [], []
    for b, s in enumerate(screen_list):
        if len(s):
            idx.append(s.to(torch.int32))
            img.append(torch.full((len(s),), b, dtype=torch.int32))
    n_pos = sum(len(i) for i in idx)
    if not n_pos:
        return 0, None, None, None, None
    screen = torch.cat(idx).to(dev)
    simg = torch.cat(img).to(dev)
    label = label_expend.to(dev, torch.float32).contiguous()
    return n_pos, screen.data_ptr(), simg.data_ptr(), label.data_ptr(), (screen, simg, label)


class _PositiveStager:
    """_pack_positives without a host-device synchronisation: the three arrays go through page-locked staging
    buffers (two slots, used in turn; a slot is reused only after the copy that read it has completed) into device
    buffers that persist, with asynchronous copies on the current stream.  The Python thread can then enqueue the
    backward pass while the forward pass is still running."""

    def __init__(self, dev):
        self.dev = dev
        self.slots = [None, None]
        self.turn = 0

    def _slot(self, n_pos, n_label):
        i = self.turn
        self.turn ^= 1
        s = self.slots[i]
        if s is None or s['cap'] < n_pos or s['lcap'] < n_label:
            cap, lcap = max(1024, 2 * n_pos), max(1024, 2 * n_label)
            s = self.slots[i] = dict(
                cap=cap, lcap=lcap, event=None,
                h_idx=torch.empty(2 * cap, dtype=torch.int32).pin_memory(), h_lab=torch.empty(lcap, dtype=torch.float32).pin_memory(),
                d_idx=torch.empty(2 * cap, dtype=torch.int32, device=self.dev), d_lab=torch.empty(lcap, dtype=torch.float32, device=self.dev))
        if s['event'] is not None:
            s['event'].synchronize()             # the copy that read this slot two calls ago (long done)
        return s

    def pack(self, screen_list, label_expend):
        n_pos = sum(len(t) for t in screen_list)
        if not n_pos:
            return 0, None, None, None, None
        lab = label_expend.to(torch.float32).contiguous().view(-1)
        s = self._slot(n_pos, lab.numel())
        cap, o = s['cap'], 0
        for b, t in enumerate(screen_list):
            n = len(t)
            if n:
                s['h_idx'][o:o + n] = t
                s['h_idx'][cap + o:cap + o + n] = b
                o += n
        s['h_lab'][:lab.numel()] = lab
        s['d_idx'][:n_pos].copy_(s['h_idx'][:n_pos], non_blocking=True)
        s['d_idx'][cap:cap + n_pos].copy_(s['h_idx'][cap:cap + n_pos], non_blocking=True)
        s['d_lab'][:lab.numel()].copy_(s['h_lab'][:lab.numel()], non_blocking=True)
        s['event'] = torch.cuda.Event()
        s['event'].record()
        es = s['d_idx'].element_size()
        return n_pos, s['d_idx'].data_ptr(), s['d_idx'].data_ptr() + cap * es, s['d_lab'].data_ptr(), s


def detection_loss(pred, screen_list, label_expend, label_smooth):
    """loss_class.__call__ (train_class.py:397-474) on any (B,R,K) float32 CUDA prediction, value only
    (the validation loss, train_class.py:311).  Returns a (4,) device tensor."""
    if pred.dtype != torch.float32 or not pred.is_cuda or pred.dim() != 3:
        raise ValueError('pred must be a (B,R,K) float32 CUDA tensor')
    pred = pred.contiguous()
    dev = pred.device
    b, r, k = pred.shape
    scratch = torch.empty_like(pred)            # the kernels flag positives in the gradient buffer
    out = torch.empty(4, dtype=torch.float32, device=dev)
    ws = torch.empty(int(lib.odamd_loss_workspace_bytes()), dtype=torch.uint8, device=dev)
    if hasattr(screen_list, 'n_pos'):
        a = screen_list
        check(lib.odamd_detection_loss_dev(pred.data_ptr(), scratch.data_ptr(), a.screen.data_ptr(),
                                           a.screen_img.data_ptr(), a.label.data_ptr(), b, r, k, a.n_pos.data_ptr(),
                                           a.capacity, float(label_smooth), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                           _stream()), 'detection_loss_dev')
        return out
    n_pos, sp, ip, lp, keep = _pack_positives(screen_list, label_expend, dev)
    check(lib.odamd_detection_loss(pred.data_ptr(), scratch.data_ptr(), sp, ip, lp, b, r, k, n_pos, float(label_smooth),
                                   out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), 'detection_loss')
    out._keep = keep
    return out


def gradient_write_modes(ops, buffer_channels):
    """Walks the backward pass of a lowered op list symbolically.  Every activation gradient is written by the
    data-gradient launches of the layers that consumed the activation and by the pool / upsample backward kernels, which
    add into it.  A convolution's data gradient that is the FIRST writer of its channel range can store; afterwards
    everything adds.  `ops`: dicts with 'kind' and Act-like 'x' / 'y' / 'ys' (buf, c0, cphys, segments);
    `buffer_channels`: physical channel count of every buffer.  Returns (ids of the ops whose data gradient stores,
    buffers that still need a zero fill per step: those where an adding kernel or a reader meets channels nobody
    stored to)."""
    written = [[False] * c for c in buffer_channels]
    stores, need_zero = set(), set()

    def span(act, off=0, n=None):
        lo = act.c0 + off
        return act.buf, lo, lo + (act.cphys if n is None else n)

    def covered(buf, lo, hi):
        return all(written[buf][lo:hi])

    def mark(buf, lo, hi):
        written[buf][lo:hi] = [True] * (hi - lo)

    def read(act):
        buf, lo, hi = span(act)
        if not covered(buf, lo, hi):
            need_zero.add(buf)

    def add(act):
        buf, lo, hi = span(act)
        if not covered(buf, lo, hi):
            need_zero.add(buf)
        mark(buf, lo, hi)

    for op in reversed(ops):
        kind = op['kind']
        if kind == 'first':
            read(op['y'])
        elif kind in ('conv', 'head'):
            if kind == 'conv':
                read(op['y'])
            buf, lo, hi = span(op['x'])
            if not any(written[buf][lo:hi]):
                stores.add(id(op))
            elif not covered(buf, lo, hi):
                need_zero.add(buf)
            mark(buf, lo, hi)
        elif kind in ('maxpool2', 'upsample2'):
            read(op['y'])
            add(op['x'])
        elif kind == 'spp':
            for y in op['ys']:
                read(y)
                add(op['x'])
        elif kind == 'from_nchw':
            read(op['y'])
        elif kind == 'to_nchw':
            for off, n in op['x'].segments:
                mark(*span(op['x'], off, n))
    return stores, need_zero


class TrainPlan:
    """Forward (train mode) + backward launch lists for one (batch, size, dtype)."""

    # Switches for tests, which compare the results of both settings: first data-gradient writer stores (else every
    # writer adds into zero-filled buffers); packed-weight kernels where one applies (else plain weights everywhere).
    STORE_FIRST = True
    PACKED_WEIGHTS = True

    def __init__(self, model, state, batch, size, dtype=F32, block_input=None):
        """`model`: a yolov7 mirror (whole network incl. decode), or any block when `block_input`
        = (channels, height, width) is given (NCHW float32 in, NCHW float32 out; used by tests)."""
        self.model, self.state, self.batch, self.size, self.dtype = model, state, batch, size, dtype
        dev = state.param.device
        self.dev = dev
        pb = PlanBuilder(batch, dtype, dev)
        self.block = block_input is not None
        if self.block:
            c, h, w = block_input
            out = model._emit(pb, pb.from_nchw(c, h, w))
            pb.to_nchw(out)
            self.head_ops = []
            self.out_nchw = torch.empty((batch, out.c, out.h, out.w), dtype=torch.float32, device=dev)
            self.gout_nchw = torch.empty_like(self.out_nchw)
            self.dx_nchw = torch.empty((batch, c, h, w), dtype=torch.float32, device=dev)
        else:
            self.head_ops = model._emit(pb, (size, size))
        self.pb = pb
        self.es = 4 if dtype == F32 else 2
        self.tdt = torch.float32 if dtype == F32 else torch.bfloat16     # activations, gradients, packed weights
        self._lower()

    FUSE_SIBLINGS = True      # switch for tests, which compare both settings

    def _fuse_siblings(self, ops):
        """Two consecutive 1x1 cbs ops on the same input whose outputs are adjacent channel ranges of one buffer and
        whose parameters, gradients and BatchNorm buffers are adjacent in the flat state (FlatState orders the pairs
        that way) become ONE op of the summed width: one convolution, one BatchNorm (statistics are per channel), one
        data gradient that reads both gradients at once, one weight gradient."""
        if not self.FUSE_SIBLINGS:
            return ops
        st = self.state
        pairs = {id(a): b for a, b in getattr(st, 'sibling_pairs', [])}
        out, i = [], 0
        while i < len(ops):
            a = ops[i]
            b = ops[i + 1] if i + 1 < len(ops) else None
            ok = (b is not None and a['kind'] == 'conv' and b['kind'] == 'conv' and a['x'] is b['x'] and pairs.get(id(a['mod'])) is b['mod']
                  and a['k'] == 1 and b['k'] == 1 and a['s'] == 1 and b['s'] == 1 and not a['pool'] and not b['pool']
                  and a['x'].cphys == a['x'].c
                  and a['y'].buf == b['y'].buf and a['y'].c == a['y'].cphys and len(a['y'].segments) == 1
                  and len(b['y'].segments) == 1 and b['y'].c == b['y'].cphys and b['y'].c0 == a['y'].c0 + a['y'].cphys)
            if ok:
                ma, mb = a['mod'], b['mod']
                for ta, tb in ((ma.conv.weight, mb.conv.weight), (ma.bn.weight, mb.bn.weight), (ma.bn.bias, mb.bn.bias)):
                    ok = ok and st.offsets[id(tb)] == st.offsets[id(ta)] + ta.numel()
                for ta, tb in ((ma.bn.running_mean, mb.bn.running_mean), (ma.bn.running_var, mb.bn.running_var)):
                    ok = ok and st.buf_offsets[id(tb)] == st.buf_offsets[id(ta)] + ta.numel()
            if ok:
                ya, yb = a['y'], b['y']
                f = dict(a)
                f['mods'] = [ma, mb]
                f['cout'] = ya.c + yb.c
                f['y'] = Act(ya.buf, ya.c0, [(0, ya.c + yb.c)], ya.h, ya.w, ya.align)
                out.append(f)
                i += 2
            else:
                out.append(a)
                i += 1
        return out

    # ---- helpers ---------------------------------------------------------------------------
    def _aptr(self, act):
        return self.bufs[act.buf].data_ptr() + act.c0 * self.es

    def _gptr(self, act):
        return self.gbufs[act.buf].data_ptr() + act.c0 * self.es

    def _ld(self, act):
        return self.bufs[act.buf].shape[3]

    def _lower(self):
        pb, dev, B = self.pb, self.dev, self.batch
        st = self.state
        # conv(pool=True) -> explicit max pool + conv (its backward needs the pooled tensor)
        ops = []
        for op in pb.ops:
            if op['kind'] == 'conv' and op['pool']:
                pooled = pb.single(op['x'].c, op['x'].h // 2, op['x'].w // 2)
                ops.append(dict(kind='maxpool2', x=op['x'], y=pooled))
                op = dict(op, x=pooled, pool=False)
            ops.append(op)
        ops = self._fuse_siblings(ops)
        self.ops = ops
        # activation buffers and their gradients (one pool, zeroed once per step)
        self.bufs = [torch.zeros((B, h, w, c), dtype=self.tdt, device=dev) for c, h, w in pb.buffers]
        total = sum(t.numel() for t in self.bufs)
        self.grad_pool = torch.zeros(total, dtype=self.tdt, device=dev)
        # Gradient buffers whose first writer of the backward pass stores (instead of adding) need no zero fill; the
        # others come first in the pool and are cleared once per step.
        self.dgrad_stores, need_zero = self._gradient_write_modes()
        if not self.STORE_FIRST:
            self.dgrad_stores, need_zero = set(), set(range(len(self.bufs)))
        order = sorted(range(len(self.bufs)), key=lambda i: (i not in need_zero, i))
        self.gbufs, o = [None] * len(self.bufs), 0
        for i in order:
            t = self.bufs[i]
            self.gbufs[i] = self.grad_pool[o:o + t.numel()].view(t.shape)
            o += t.numel()
            if i in need_zero:
                self.grad_zero = self.grad_pool[:o]
        if not need_zero:
            self.grad_zero = None
        self.image8 = None if self.block else torch.zeros((B, self.size, self.size, 8), dtype=self.tdt, device=dev)
        maxc = 16
        self.fwd, self.bwd = [], []
        self.keep = []
        pack = []
        self.bucket_marks = {}
        for op in ops:
            kind = op['kind']
            if kind in ('conv', 'first'):
                m = op['mod']
                cout = op.get('cout', m.conv.out_channels)      # a fused sibling pair: the summed width
                cphys = _round_up(cout, self.pb.align)
                y_act = op['y']
                ho, wo = y_act.h, y_act.w
                raw = torch.zeros((B, ho, wo, cphys), dtype=self.tdt, device=dev)
                stats = torch.zeros((4, cout), dtype=torch.float32, device=dev)   # mean, invstd, scale, shift
                op['raw'], op['stats'] = raw, stats
                maxc = max(maxc, cphys)
                npix = B * ho * wo
                k = m.conv.kernel_size[0]
                if kind == 'first':
                    wf = torch.empty(cout * 27, dtype=torch.float32, device=dev)
                    pd = self._pack_desc(m.conv.weight, wf, None, None, cout)
                    pd.fwd_f32 = 1          # the first-layer kernel takes float32 weights in both modes
                    pack.append(pd)
                    op['wf'] = wf
                    self.fwd.append((lib.odamd_conv_first_forward,
                                     [None, wf.data_ptr(), None, None, raw.data_ptr(), B, ho, wo, cout, cphys, 0,
                                      self.dtype], 'first', 0))
                else:
                    x = op['x']        # may hold zero-padded gaps (pruned, irregular widths): the packed weights
                    #                    have zero columns there
                    d = self._conv_desc(self._aptr(x), None, None, raw.data_ptr(), x.h, x.w, x.cphys, cout, self._ld(x),
                                        cphys, k, op['s'])
                    wf, ffrag = self._weights_for(d, cout, k * k * x.cphys)
                    # data gradient: a convolution of dY (cphys channels) with the flipped, transposed weights
                    dg = self._conv_desc(raw.data_ptr(), None, None, self._gptr(x), ho, wo, cphys, x.cphys, cphys,
                                         self._ld(x), k, 1, upz=int(op['s'] == 2), accum=int(id(op) not in self.dgrad_stores))
                    wb, bfrag = self._weights_for(dg, x.cphys, k * k * cphys)
                    pd = self._pack_desc(m.conv.weight, wf, wb, x, cout)
                    pd.fwd_frag, pd.bwd_frag = ffrag, bfrag
                    pack.append(pd)
                    op['wf'], op['wb'], op['dgrad_desc'] = wf, wb, dg
                    self.fwd.append((lib.odamd_conv_forward, [C.byref(d)], 'conv', None))
                self.fwd.append(('bn_fwd', (op, npix, cout, cphys), 'bn', None))
            elif kind == 'head':
                m, x = op['mod'], op['x']
                cout = m.output.out_channels
                cphys = _round_up(cout, self.pb.align)
                wf = torch.zeros(cout * x.cphys, dtype=self.tdt, device=dev)     # gap columns stay zero
                wb = torch.zeros(x.cphys * cphys, dtype=self.tdt, device=dev)
                pack.append(self._pack_desc(m.output.weight, wf, wb, x, cout))
                out = torch.empty((B, cout, x.h, x.w), dtype=torch.float32, device=dev)
                dyb = torch.zeros((B, x.h, x.w, cphys), dtype=self.tdt, device=dev)
                op.update(wf=wf, wb=wb, out=out, dy=dyb)
                maxc = max(maxc, cphys)
                d = self._conv_desc(self._aptr(x), wf, m.output.bias, out.data_ptr(), x.h, x.w, x.cphys, cout,
                                    self._ld(x), 0, 1, 1, nchw=1)
                self.fwd.append((lib.odamd_conv_forward, [C.byref(d)], 'head', None))
            elif kind == 'maxpool2':
                x, y = op['x'], op['y']
                self.fwd.append((lib.odamd_maxpool2_forward,
                                 [self._aptr(x), self._aptr(y), B, x.h, x.w, x.cphys, self._ld(x), self._ld(y),
                                  self.dtype], 'maxpool2', None))
            elif kind == 'spp':
                x, ys = op['x'], op['ys']
                self.fwd.append((lib.odamd_spp_pool3_forward,
                                 [self._aptr(x), self._aptr(ys[0]), self._aptr(ys[1]), self._aptr(ys[2]), B, x.h, x.w,
                                  x.cphys, self._ld(x), self._ld(ys[0]), self.dtype], 'spp', None))
            elif kind == 'upsample2':
                x, y = op['x'], op['y']
                self.fwd.append((lib.odamd_upsample2_forward,
                                 [self._aptr(x), self._aptr(y), B, x.h, x.w, x.cphys, self._ld(x), self._ld(y),
                                  self.dtype], 'upsample2', None))
            elif kind == 'from_nchw':
                y = op['y']
                self.fwd.append((lib.odamd_nchw_to_nhwc,
                                 [None, self._aptr(y), B, y.c, y.h, y.w, y.c, self._ld(y), self.dtype], 'from_nchw', 0))
            elif kind == 'to_nchw':
                x, dst = op['x'], 0
                for off, n in x.segments:
                    self.fwd.append((lib.odamd_nhwc_to_nchw,
                                     [self._aptr(x) + off * self.es, self.out_nchw.data_ptr() + dst * x.h * x.w * 4, B, n, x.h,
                                      x.w, self._ld(x), x.c, self.dtype], 'to_nchw', None))
                    dst += n
            else:
                raise AssertionError(kind)
        # packing table
        arr = (_lib.PackDesc * len(pack))(*pack)
        self.pack_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self.n_pack = len(pack)
        self.ws = torch.empty(int(lib.odamd_bn_workspace_bytes(maxc)) + 8 * maxc, dtype=torch.uint8, device=dev)
        self.loss_ws = torch.empty(int(lib.odamd_loss_workspace_bytes()), dtype=torch.uint8, device=dev)
        self.loss_out = torch.zeros(4, dtype=torch.float32, device=dev)
        if not self.block:
            k = 5 + self.model.output_class
            rows = sum(o['out'].shape[1] // k * o['out'].shape[2] * o['out'].shape[3] for o in self.head_ops)
            self.pred = torch.empty((B, rows, k), dtype=torch.float32, device=dev)
            self.dpred = torch.empty_like(self.pred)
            self.dheads = [torch.empty_like(o['out']) for o in self.head_ops]
        self._lower_backward()

    def _pack_desc(self, w, fwd, bwd, x_act, cout):
        d = _lib.PackDesc()
        d.w = w.data_ptr()
        d.fwd = fwd.data_ptr()
        d.bwd = bwd.data_ptr() if bwd is not None else None
        d.cout, d.cin, d.k = cout, w.shape[1], w.shape[2]       # cout: of a fused sibling pair, whose weights follow `w`
        d.cin_phys = x_act.cphys if x_act is not None else w.shape[1]
        d.cout_phys = _round_up(cout, self.pb.align)
        segs = x_act.segments if x_act is not None else [(0, w.shape[1])]
        d.n_seg = len(segs)
        for i, (off, n) in enumerate(segs):
            d.seg_off[i], d.seg_len[i] = off, n
        return d

    def _weights_for(self, d, rows, row_len):
        """Allocates the weight copy a launch reads and decides its layout: the fragment order of the kernels with
        packed weights where one of them serves the launch (bf16; they apply scale and shift, here 1 and 0), else the
        plain [rows][row_len] order.  Returns (tensor, PackDesc fragment code)."""
        frag = 0
        if self.PACKED_WEIGHTS and self.dtype == BF16 and not d.in_upzero2 and not d.accumulate and d.cout <= 4096:
            if getattr(self, '_unit', None) is None:
                self._unit = (torch.ones(4096, dtype=torch.float32, device=self.dev),
                              torch.zeros(4096, dtype=torch.float32, device=self.dev))
            g, cb = C.c_int(0), C.c_int(0)
            d.scale, d.shift = self._unit[0].data_ptr(), self._unit[1].data_ptr()
            if lib.odamd_conv_weight_layout(C.byref(d), C.byref(g), C.byref(cb)):
                frag = g.value | (cb.value << 8)
                rows = _round_up(rows, 32 * g.value)
                d.w_packed = 1
            else:
                d.scale, d.shift = None, None
        w = torch.zeros(rows * row_len, dtype=self.tdt, device=self.dev)
        d.w = w.data_ptr()
        return w, frag

    def _conv_desc(self, x_ptr, w, shift, y_ptr, h, w_in, cin, cout, x_ld, y_ld, k, s, nchw=0, upz=0, accum=0):
        d = _lib.ConvDesc()
        d.x, d.w, d.scale = x_ptr, (w.data_ptr() if w is not None else None), None
        d.shift = shift.data_ptr() if shift is not None else None
        d.y = y_ptr
        d.batch, d.h, d.w_in, d.cin, d.cout = self.batch, h, w_in, cin, cout
        d.x_ld, d.y_ld, d.ksize, d.stride, d.act = x_ld, y_ld, k, s, 0
        d.dtype, d.in_pool2, d.out_nchw_f32, d.in_upzero2, d.accumulate = self.dtype, 0, nchw, upz, accum
        self.keep.append(d)
        return d

    def _gradient_write_modes(self):
        return gradient_write_modes(self.ops, [c for c, _, _ in self.pb.buffers])

    def _lower_backward(self):
        B, st = self.batch, self.state
        bwd = []
        for op in reversed(self.ops):
            kind = op['kind']
            if kind in ('conv', 'first'):
                m = op['mod']
                cout = op.get('cout', m.conv.out_channels)
                raw = op['raw']
                cphys = raw.shape[3]
                k = m.conv.kernel_size[0]
                npix = B * raw.shape[1] * raw.shape[2]
                bwd.append(('bn_bwd', (op, npix, cout, cphys), 'bn_bwd', None))
                if kind == 'first':
                    bwd.append((lib.odamd_conv_wgrad,
                                [self.image8.data_ptr(), 8, raw.data_ptr(), cphys, st.grad_ptr(m.conv.weight), B,
                                 self.size, self.size, 3, cout, 3, 1, self.dtype], 'wgrad', None))
                else:
                    x = op['x']
                    if x.cphys == x.c:      # no padding gaps: the whole concat is one run of real channels
                        bwd.append((lib.odamd_conv_wgrad,
                                    [self._aptr(x), self._ld(x), raw.data_ptr(), cphys, st.grad_ptr(m.conv.weight), B, x.h,
                                     x.w, x.c, cout, k, op['s'], self.dtype], 'wgrad', None))
                    else:
                        # a concat of irregular widths: every slice is a run of real channels followed by zero
                        # padding, the weight's input channels are the runs back to back.  One weight-gradient
                        # launch per run into a scratch tensor, added to its columns of the parameter gradient.
                        src = 0
                        for off, n in x.segments:
                            bwd.append((lib.odamd_conv_wgrad,
                                        [self._aptr(x) + off * self.es, self._ld(x), raw.data_ptr(), cphys, None, B, x.h,
                                         x.w, n, cout, k, op['s'], self.dtype], 'wgrad', ('columns', m.conv.weight, src, n)))
                            src += n
                    bwd.append((lib.odamd_conv_forward, [C.byref(op['dgrad_desc'])], 'dgrad', None))
                bwd.append(('mark', tuple(t for q in op.get('mods', [m]) for t in (q.conv.weight, q.bn.weight, q.bn.bias)),
                            'mark', None))
            elif kind == 'head':
                m, x = op['mod'], op['x']
                cout = m.output.out_channels
                dyb = op['dy']
                cphys = dyb.shape[3]
                i = self.head_ops.index(op)
                npix = B * x.h * x.w
                bwd.append((lib.odamd_nchw_to_nhwc,
                            [self.dheads[i].data_ptr(), dyb.data_ptr(), B, cout, x.h, x.w, cout, cphys, self.dtype],
                            'to_nhwc', None))
                bwd.append((lib.odamd_bias_grad,
                            [dyb.data_ptr(), cphys, npix, cphys, None, self.dtype, self.ws.data_ptr(), self.ws.numel()],
                            'bias_grad', ('bias', m.output.bias, cout, cphys)))
                bwd.append((lib.odamd_conv_wgrad,
                            [self._aptr(x), self._ld(x), dyb.data_ptr(), cphys, st.grad_ptr(m.output.weight), B, x.h,
                             x.w, x.c, cout, 1, 1, self.dtype], 'wgrad', None))
                d = self._conv_desc(dyb.data_ptr(), op['wb'], None, self._gptr(x), x.h, x.w, cphys, x.cphys, cphys,
                                    self._ld(x), 1, 1, accum=int(id(op) not in self.dgrad_stores))
                bwd.append((lib.odamd_conv_forward, [C.byref(d)], 'dgrad', None))
                bwd.append(('mark', (m.output.weight, m.output.bias), 'mark', None))
            elif kind == 'maxpool2':
                x, y = op['x'], op['y']
                bwd.append((lib.odamd_maxpool2_backward,
                            [self._aptr(x), self._ld(x), self._gptr(y), self._ld(y), self._gptr(x), self._ld(x), B, x.h,
                             x.w, x.cphys, self.dtype], 'maxpool2_bwd', None))
            elif kind == 'spp':
                x, ys = op['x'], op['ys']
                bwd.append((lib.odamd_spp_backward,
                            [self._aptr(x), self._ld(x), self._gptr(ys[0]), self._gptr(ys[1]), self._gptr(ys[2]), self._ld(ys[0]),
                             self._gptr(x), self._ld(x), B, x.h, x.w, x.cphys, self.dtype], 'spp_bwd', None))
            elif kind == 'from_nchw':
                y = op['y']
                bwd.append((lib.odamd_nhwc_to_nchw,
                            [self._gptr(y), self.dx_nchw.data_ptr(), B, y.c, y.h, y.w, self._ld(y), y.c, self.dtype],
                            'dx_to_nchw', None))
            elif kind == 'to_nchw':
                x, dst = op['x'], 0
                for off, n in x.segments:
                    bwd.append((lib.odamd_nchw_to_nhwc,
                                [self.gout_nchw.data_ptr() + dst * x.h * x.w * 4, self._gptr(x) + off * self.es, B, n, x.h, x.w,
                                 x.c, self._ld(x), self.dtype], 'gout_to_nhwc', None))
                    dst += n
            elif kind == 'upsample2':
                x, y = op['x'], op['y']
                bwd.append((lib.odamd_upsample2_backward,
                            [self._gptr(y), self._ld(y), self._gptr(x), self._ld(x), B, x.h, x.w, x.cphys, self.dtype],
                            'upsample2_bwd', None))
        # one workspace for the partial sums of every weight-gradient launch (they run one at a time)
        need = 16
        for fn, args, name, _ in bwd:
            if name == 'wgrad':
                need = max(need, int(lib.odamd_conv_wgrad_workspace_bytes(*args[5:13])))
        self.wgrad_ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
        for fn, args, name, _ in bwd:
            if name == 'wgrad':
                args.extend([self.wgrad_ws.data_ptr(), need])
        self.bwd = bwd
        # scratch for the head bias gradient (padded channel count), added to the real slot afterwards
        self.bias_tmp = {}

    # ---- execution -------------------------------------------------------------------------
    def _bn_fwd(self, op, npix, cout, cphys, stream):
        m, raw, stats = op['mod'], op['raw'], op['stats']
        y = op['y']
        check(lib.odamd_bn_train_forward(raw.data_ptr(), cphys, self._aptr(y), self._ld(y), npix, cout,
                                         m.bn.weight.data_ptr(), m.bn.bias.data_ptr(), BN_EPS, BN_MOMENTUM,
                                         m.bn.running_mean.data_ptr(), m.bn.running_var.data_ptr(),
                                         stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(),
                                         stats[3].data_ptr(), self.dtype, self.ws.data_ptr(), self.ws.numel(), stream),
              'bn_train_forward')

    def _bn_bwd(self, op, npix, cout, cphys, stream):
        m, raw, stats = op['mod'], op['raw'], op['stats']
        y = op['y']
        st = self.state
        check(lib.odamd_bn_silu_backward(raw.data_ptr(), cphys, self._gptr(y), self._ld(y), npix, cout,
                                         stats[2].data_ptr(), stats[3].data_ptr(), stats[0].data_ptr(),
                                         stats[1].data_ptr(), st.grad_ptr(m.bn.weight), st.grad_ptr(m.bn.bias),
                                         self.dtype, self.ws.data_ptr(), self.ws.numel(), stream), 'bn_silu_backward')

    def forward(self, image):
        """Train-mode forward; returns the decoded (B,R,K) prediction (decode leaves channels 4: raw)."""
        stream = _stream()
        check(lib.odamd_pack_weights(self.pack_table.data_ptr(), self.n_pack, self.dtype, stream), 'pack_weights')
        img = image.data_ptr()
        for fn, args, name, slot in self.fwd:
            if fn == 'bn_fwd':
                self._bn_fwd(*args, stream)
                continue
            if slot is not None:
                args[slot] = img
            check(fn(*args, stream), name)
        if self.block:                       # a block of a larger network: only its own BatchNorm layers ran
            for m in self.model.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.num_batches_tracked.add_(1)
        else:
            self.state.counters.add_(1)      # every BatchNorm2d.num_batches_tracked (views of this vector)
        engine.touch_weights()               # BN running buffers were written through raw pointers
        if self.block:
            return self.out_nchw
        k = 5 + self.model.output_class
        heads = [o['out'].view(self.batch, o['out'].shape[1] // k, o['out'].shape[2], o['out'].shape[3], k)
                 for o in self.head_ops]
        self.heads = heads
        dec = self.model.decode
        d = _post.decode_desc(heads, dec.input_size, dec.anchor, False)
        self.decode_desc = d
        check(lib.odamd_decode(C.byref(d), self.pred.data_ptr(), stream), 'decode')
        return self.pred

    def loss(self, screen_list, label_expend, label_smooth):
        """loss_class.__call__ on the last forward's prediction; fills self.dpred; returns a (4,) tensor."""
        dev = self.dev
        if hasattr(screen_list, 'n_pos'):        # DeviceAssignment: count stays on the device
            a = screen_list
            b, r, k = self.pred.shape
            self._keep_loss = a
            check(lib.odamd_detection_loss_dev(self.pred.data_ptr(), self.dpred.data_ptr(), a.screen.data_ptr(),
                                               a.screen_img.data_ptr(), a.label.data_ptr(), b, r, k, a.n_pos.data_ptr(),
                                               a.capacity, float(label_smooth), self.loss_out.data_ptr(),
                                               self.loss_ws.data_ptr(), self.loss_ws.numel(), _stream()),
                  'detection_loss_dev')
            return self.loss_out
        if getattr(self, '_stager', None) is None:
            self._stager = _PositiveStager(dev)
        n_pos, sp, ip, lp, self._keep_loss = self._stager.pack(screen_list, label_expend)
        b, r, k = self.pred.shape
        check(lib.odamd_detection_loss(self.pred.data_ptr(), self.dpred.data_ptr(), sp, ip, lp, b, r, k, n_pos,
                                       float(label_smooth), self.loss_out.data_ptr(), self.loss_ws.data_ptr(),
                                       self.loss_ws.numel(), _stream()), 'detection_loss')
        return self.loss_out

    def backward(self, image, reducer=None):
        """Backward of the last forward from self.dpred; gradients are ADDED to state.grad."""
        stream = _stream()
        if self.grad_zero is not None:
            self.grad_zero.zero_()
        main = torch.cuda.current_stream()
        side = self._side_stream()
        side_busy = False
        if not self.block:
            # the NHWC copy of the image is read by the first layer's weight gradient only (last launch of the second stream)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                check(lib.odamd_nchw_to_nhwc(image.data_ptr(), self.image8.data_ptr(), self.batch, 3, self.size, self.size, 3,
                                             8, self.dtype, _stream()), 'image_to_nhwc')
            side_busy = True
            arr = (C.c_void_p * 3)(*[g.data_ptr() for g in self.dheads])
            check(lib.odamd_decode_backward(C.byref(self.decode_desc), self.dpred.data_ptr(), C.byref(arr), stream),
                  'decode_backward')
        pending = None
        if reducer is not None:
            pending = self._bucket_plan(reducer)
        # The weight gradients are off the critical path (layer L's data gradient and layer L-1's BatchNorm backward do
        # not read them): they run on a second stream, one after the other, each behind an event recorded after the
        # BatchNorm backward that produced its dY.  Small layers do not fill the chip, so the two streams overlap.
        for i, (fn, args, name, extra) in enumerate(self.bwd):
            if fn == 'bn_bwd':
                self._bn_bwd(*args, stream)
            elif fn == 'mark':
                if pending is not None:
                    buckets = pending.pop(i, ())
                    if buckets and side_busy:
                        main.wait_stream(side)          # the buckets' weight gradients are complete
                        side_busy = False
                    for bucket in buckets:
                        reducer.ready(bucket)
            elif name == 'wgrad':
                ev = self._events[i] if i in self._events else self._events.setdefault(i, torch.cuda.Event())
                ev.record(main)
                side.wait_event(ev)
                side_busy = True
                with torch.cuda.stream(side):
                    if extra is None:
                        check(fn(*args, _stream()), name)
                        continue
                    _, weight, src, n = extra
                    cout, _, k, _ = weight.shape
                    tmp = self.bias_tmp.get(('wseg', cout, n, k))
                    if tmp is None:
                        tmp = self.bias_tmp[('wseg', cout, n, k)] = torch.zeros((cout, n, k, k), dtype=torch.float32, device=self.dev)
                    tmp.zero_()
                    args[4] = tmp.data_ptr()
                    check(fn(*args, _stream()), name)
                    weight.grad[:, src:src + n].add_(tmp)
            elif name == 'bias_grad':
                _, bias, cout, cphys = extra
                tmp = self.bias_tmp.get(id(bias))
                if tmp is None:
                    tmp = self.bias_tmp[id(bias)] = torch.zeros(cphys, dtype=torch.float32, device=self.dev)
                tmp.zero_()
                args[4] = tmp.data_ptr()
                check(fn(*args, stream), name)
                bias.grad.add_(tmp[:cout])
            else:
                check(fn(*args, stream), name)
        if side_busy:
            main.wait_stream(side)

    def _side_stream(self):
        if getattr(self, '_side', None) is None:
            self._side = engine.shared_stream(self.dev, 'side')
            self._events = {}
        return self._side

    def _bucket_plan(self, reducer):
        """launch index of the 'mark' after which a bucket's last gradient has been written."""
        st = self.state
        last = {}
        for i, (fn, args, name, _) in enumerate(self.bwd):
            if fn == 'mark':
                # a bucket is final once the LAST tensor with any element inside it has its gradient
                for p in args:
                    for b in reducer.buckets_of(st.offsets[id(p)], p.numel()):
                        last[b] = i
        plan = {}
        for bucket, i in last.items():
            plan.setdefault(i, []).append(bucket)
        for b in range(len(reducer.bounds)):
            if b not in last:
                plan.setdefault(max(last.values()), []).append(b)
        return plan


def lr_schedule(step_now, lr_start, lr_end, step_all, warmup_step):
    """lr_adjust.__call__ (train_class.py:511-521) for the already incremented step counter."""
    lr = lr_end + (lr_start - lr_end) * math.cos(math.pi / 2 * step_now / step_all)
    if step_now <= warmup_step:
        lr = lr * (0.1 + 0.9 * step_now / warmup_step)
    return max(lr, 0.000001)


def ema_decay(update_total, decay=0.9999, tau=2000):
    """model_ema.decay (train_class.py:481)."""
    return decay * (1 - math.exp(-update_total / tau))


def optimizer_step(state, lr, weight_decay=0.0005, betas=(0.937, 0.999), eps=1e-8, grad_scale=1.0, ema=True):
    """Adam with coupled L2 over the flat buffers, fused with the EMA of the parameters; the EMA
    of the BN running buffers follows in a second launch (train_class.py:211-213)."""
    state.step += 1
    engine.touch_weights()                   # parameters are about to be written through raw pointers
    d = 0.0
    ema_ptr = None
    if ema and state.ema_param is not None:
        state.ema_updates += 1
        d = ema_decay(state.ema_updates)
        ema_ptr = state.ema_param.data_ptr()
    check(lib.odamd_adam_ema_step(state.param.data_ptr(), state.grad.data_ptr(), state.exp_avg.data_ptr(),
                                  state.exp_avg_sq.data_ptr(), ema_ptr, state.n, lr, betas[0], betas[1], eps,
                                  weight_decay, state.step, d, grad_scale, _stream()), 'adam_ema_step')
    if ema_ptr is not None and state.nb:
        check(lib.odamd_ema_update(state.ema_buffers.data_ptr(), state.buffers.data_ptr(), state.nb, d, _stream()),
              'ema_update')
