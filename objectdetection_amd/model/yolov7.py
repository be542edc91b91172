"""YOLOv7 detector, API-compatible with the reference's model/yolov7.py (same constructor,
attribute names, s/m/l presets and pruned `config` form), lowered to one execution plan.

Reference: model/yolov7.py:6-146.
"""
import torch

from .. import engine
from .. import postprocess as _post
from .layer import cbs, elan, elan_h, mp, sppcspc, concat, head, decode, _w

_BASE = {'s': 16, 'm': 32, 'l': 64}     # base width
_DEPTH = {'s': 1, 'm': 2, 'l': 3}       # 3x3 convs per ELAN branch


class yolov7(torch.nn.Module):
    def __init__(self, args, config=None):
        super().__init__()
        d, n = _BASE[args.model_type], _DEPTH[args.model_type]
        self.model_type = args.model_type
        self.pruned_config = None if config is None else [int(c) for c in config]
        self.output_size = args.output_size
        self.output_class = args.output_class
        self.decode = decode(args)
        if config is None:
            self._build_standard(d, n)
        else:
            self._build_pruned(list(config), n)
        self.head0 = head(_w(self.elan_h19), self.output_size[0], self.output_class)
        self.head1 = head(_w(self.elan_h22), self.output_size[1], self.output_class)
        self.head2 = head(_w(self.elan_h25), self.output_size[2], self.output_class)
        self._plans = {}
        self._state = None          # FlatState, created by the first training forward
        self._train_plans = {}
        self._hook = None

    # ---- construction ------------------------------------------------------------------------
    def _build_standard(self, d, n):
        self.cbs0 = cbs(3, d, 3, 1)
        self.cbs1 = cbs(d, 2 * d, 3, 2)               # /2
        self.cbs2 = cbs(2 * d, 2 * d, 3, 1)
        self.cbs3 = cbs(2 * d, 4 * d, 3, 2)           # /4
        self.elan4 = elan(4 * d, 8 * d, n)
        self.mp5 = mp(8 * d, 8 * d)                   # /8
        self.elan6 = elan(8 * d, 16 * d, n)
        self.mp7 = mp(16 * d, 16 * d)                 # /16
        self.elan8 = elan(16 * d, 32 * d, n)
        self.mp9 = mp(32 * d, 32 * d)                 # /32
        self.elan10 = elan(32 * d, 32 * d, n)
        self.sppcspc11 = sppcspc(32 * d, 16 * d)
        self.cbs12 = cbs(16 * d, 8 * d, 1, 1)
        self.upsample13 = torch.nn.Upsample(scale_factor=2)
        self.l8_concat = cbs(32 * d, 8 * d, 1, 1)
        self.concat14 = concat(1)
        self.elan_h15 = elan_h(16 * d, 8 * d)
        self.cbs16 = cbs(8 * d, 4 * d, 1, 1)
        self.upsample17 = torch.nn.Upsample(scale_factor=2)
        self.l6_concat = cbs(16 * d, 4 * d, 1, 1)
        self.concat18 = concat(1)
        self.elan_h19 = elan_h(8 * d, 4 * d)          # feeds head0
        self.mp20 = mp(4 * d, 8 * d)
        self.concat21 = concat(1)
        self.elan_h22 = elan_h(16 * d, 8 * d)         # feeds head1
        self.mp23 = mp(8 * d, 16 * d)
        self.concat24 = concat(1)
        self.elan_h25 = elan_h(32 * d, 16 * d)        # feeds head2

    def _build_pruned(self, cfg, n):
        """`cfg` lists the output width of every cbs in forward order (train_class.py:367-386)."""
        pos = [0]

        def take(k=1):
            out = cfg[pos[0]:pos[0] + k]
            pos[0] += k
            return out if k > 1 else out[0]

        def rest():
            return cfg[pos[0]:]

        def block(cls, in_, **kw):
            m = cls(in_, config=rest(), **kw)
            pos[0] += m.config_len
            return m

        c0, c1, c2, c3 = take(4)
        self.cbs0 = cbs(3, c0, 3, 1)
        self.cbs1 = cbs(c0, c1, 3, 2)
        self.cbs2 = cbs(c1, c2, 3, 1)
        self.cbs3 = cbs(c2, c3, 3, 2)
        self.elan4 = block(elan, c3, n=n)
        self.mp5 = block(mp, self.elan4.last_layer)
        self.elan6 = block(elan, self.mp5.last_layer, n=n)
        self.mp7 = block(mp, self.elan6.last_layer)
        self.elan8 = block(elan, self.mp7.last_layer, n=n)
        self.mp9 = block(mp, self.elan8.last_layer)
        self.elan10 = block(elan, self.mp9.last_layer, n=n)
        self.sppcspc11 = block(sppcspc, self.elan10.last_layer)
        c12 = take()
        self.cbs12 = cbs(self.sppcspc11.last_layer, c12, 1, 1)
        self.upsample13 = torch.nn.Upsample(scale_factor=2)
        c_l8 = take()
        self.l8_concat = cbs(self.elan8.last_layer, c_l8, 1, 1)
        self.concat14 = concat(dim=1)
        self.elan_h15 = block(elan_h, c12 + c_l8)
        c16 = take()
        self.cbs16 = cbs(self.elan_h15.last_layer, c16, 1, 1)
        self.upsample17 = torch.nn.Upsample(scale_factor=2)
        c_l6 = take()
        self.l6_concat = cbs(self.elan6.last_layer, c_l6, 1, 1)
        self.concat18 = concat(dim=1)
        self.elan_h19 = block(elan_h, c16 + c_l6)
        self.mp20 = block(mp, self.elan_h19.last_layer)
        self.concat21 = concat(dim=1)
        self.elan_h22 = block(elan_h, self.mp20.last_layer + self.elan_h15.last_layer)
        self.mp23 = block(mp, self.elan_h22.last_layer)
        self.concat24 = concat(dim=1)
        self.elan_h25 = block(elan_h, self.mp23.last_layer + self.sppcspc11.last_layer)

    # ---- lowering ------------------------------------------------------------------------------
    def _emit(self, pb, size):
        """Append the whole network to `pb`; returns the three head ops."""
        h, w = size
        x = pb.first(self.cbs0, (h, w))
        x = self.cbs3._emit(pb, self.cbs2._emit(pb, self.cbs1._emit(pb, x)))
        x = self.mp5._emit(pb, self.elan4._emit(pb, x))
        l6 = self.elan6._emit(pb, x)
        l8 = self.elan8._emit(pb, self.mp7._emit(pb, l6))
        # The lateral 1x1 convolutions of the neck (l8_concat, l6_concat) only need l8 / l6, and nothing reads their
        # results before elan_h15 / elan_h19: they are emitted here, as a side branch that runs beside the 20 x 20
        # section of the network (launch-latency bound chains that leave most of the chip idle).
        cat14, (up13, l8c) = pb.buffer([_w(self.cbs12), _w(self.l8_concat)], l8.h, l8.w)
        cat18, (up17, l6c) = pb.buffer([_w(self.cbs16), _w(self.l6_concat)], l6.h, l6.w)
        with pb.side_branch():
            self.l8_concat._emit(pb, l8, l8c)
            self.l6_concat._emit(pb, l6, l6c)
        x = self.elan10._emit(pb, self.mp9._emit(pb, l8))
        # concat24 = [mp23 | l11], concat21 = [mp20 | l15]: the late consumers' buffers exist first,
        # so sppcspc11 / elan_h15 store their result where concat24 / concat21 will read it.
        cat24, (m23a, m23b, l11) = pb.buffer([_w(self.mp23.cbs1), _w(self.mp23.cbs3), _w(self.sppcspc11)], x.h, x.w)
        self.sppcspc11._emit(pb, x, l11)
        t = self.cbs12._emit(pb, l11)
        pb.upsample2(t, up13)
        cat21, (m20a, m20b, l15) = pb.buffer([_w(self.mp20.cbs1), _w(self.mp20.cbs3), _w(self.elan_h15)], l8.h, l8.w)
        self.elan_h15._emit(pb, cat14, l15)
        t = self.cbs16._emit(pb, l15)
        pb.upsample2(t, up17)
        x = self.elan_h19._emit(pb, cat18)
        # the heads feed nothing inside the network: side branches, joined when the plan ends
        with pb.side_branch():
            heads = [pb.head(self.head0, x)]
        self.mp20._emit(pb, x, (m20a, m20b))
        x = self.elan_h22._emit(pb, cat21)
        with pb.side_branch():
            heads.append(pb.head(self.head1, x))
        self.mp23._emit(pb, x, (m23a, m23b))
        x = self.elan_h25._emit(pb, cat24)
        heads.append(pb.head(self.head2, x))
        return heads

    def _plan(self, x, slot=0):
        """`slot` selects one of several independent plans (own buffers) of the same shape, so that a
        consumer may still read the outputs of one forward while the next one runs."""
        dtype = engine.compute_dtype_for(self)
        key = (tuple(x.shape), dtype, x.device.index, slot)
        plan = self._plans.get(key)
        if plan is None:
            pb = engine.PlanBuilder(x.shape[0], dtype, x.device)
            heads = self._emit(pb, (x.shape[2], x.shape[3]))
            plan = engine.Plan(pb)
            plan.head_ops = heads
            # where the maps of stride 32 begin (model/yolov7.py:27-28: mp9, elan10): from there on most launches fill only
            # part of the chip, and a pipeline may run them beside the next batch's large-map launches (measured at batch 64:
            # any split between the first stride-16 and the first stride-32 launch gives the same step, later ones lose)
            plan.split_index = plan.first_launch_on_maps_up_to(x.shape[2] // engine.SPLIT_MAP_DIV)
            self._plans[key] = plan
        return plan

    def head_tensors(self, x, slot=0, back=None):
        """Run the network; returns the three raw head tensors (B, A, H, W, K) (views of plan memory).
        `back`: a stream for the second half of the launches (Plan.run); the tensors are then final on THAT stream."""
        if not x.is_cuda:
            raise RuntimeError('objectdetection_amd runs on the GPU only (no CPU fallback)')
        if self.training:
            raise RuntimeError('head_tensors runs the inference plan: call model.eval() first (training goes through forward())')
        plan = self._plan(x, slot)
        plan.run(x.detach().float().contiguous(), back=back, split=plan.split_index if back is not None else None)
        k = 5 + self.output_class
        out = []
        for op in plan.head_ops:
            t = op['out']
            b, ck, hh, ww = t.shape
            out.append(t.view(b, ck // k, hh, ww, k))
        return out

    def _decoded(self, x, deploy):
        heads = self.head_tensors(x)
        y = _post.decode(heads, self.decode.input_size, self.decode.anchor, deploy)
        return y if x.dtype == torch.float32 else y.to(x.dtype)

    def forward(self, x):
        if self.training:
            return self.forward_train(x)
        return self._decoded(x, False)

    # ---- training ------------------------------------------------------------------------------
    def train_state(self):
        """Flat parameter / gradient / optimizer storage; the parameters become views into it."""
        from .. import train_engine
        if self._state is None:
            self._state = train_engine.FlatState(self)
        return self._state

    def train_plan(self, batch, size):
        """Training plan in `self.train_dtype`: 'f32' (the reference default) or 'bf16' (the mixed
        precision mode that `--amp True` selects: bf16 activations / gradients / MFMA operands,
        float32 accumulation, statistics, master weights and optimizer)."""
        from .. import train_engine
        dtype = engine.BF16 if getattr(self, 'train_dtype', 'f32') == 'bf16' else engine.F32
        key = (batch, size, dtype)
        plan = self._train_plans.get(key)
        if plan is None:
            plan = self._train_plans[key] = train_engine.TrainPlan(self, self.train_state(), batch, size, dtype)
        return plan

    def forward_train(self, x):
        """Train-mode forward (batch-statistics BatchNorm, running buffers updated).  The result is
        connected to torch autograd: `loss.backward()` runs the HIP backward pass and leaves the
        gradients in `param.grad` (views of the flat gradient buffer)."""
        if not x.is_cuda:
            raise RuntimeError('objectdetection_amd runs on the GPU only (no CPU fallback)')
        if self._hook is None:
            self._hook = torch.zeros(1, device=x.device, requires_grad=True)
        plan = self.train_plan(x.shape[0], x.shape[2])
        return _TrainForward.apply(x.detach().float().contiguous(), self._hook, plan)

    def forward_deploy(self, x):
        """forward + sigmoid on channels 4: (what `deploy(model)(x)` returns)."""
        return self._decoded(x, True)

    # ---- checkpoints -------------------------------------------------------------------------
    def checkpoint_meta(self):
        """Everything needed to rebuild this module from a state_dict: plain Python data only, so the
        checkpoint can be read with torch.load(..., weights_only=True) (no code runs on load)."""
        n_cbs = [m.conv.out_channels for m in self.modules() if isinstance(m, cbs)]
        depth = len(self.elan4.sequential2)
        config = getattr(self, 'pruned_config', None)
        if config is None:
            # a module restored from a reference pickle does not say how it was built: the widths do (the
            # pruned constructor takes the width of every cbs in module order, train_class.py:367-386)
            d = _BASE[{1: 's', 2: 'm', 3: 'l'}[depth]]
            if n_cbs[:4] != [d, 2 * d, 2 * d, 4 * d] or any(w % d for w in n_cbs):
                config = n_cbs
        return {'model_type': getattr(self, 'model_type', {1: 's', 2: 'm', 3: 'l'}[depth]),
                'input_size': int(self.decode.input_size), 'output_class': int(self.output_class),
                'output_size': tuple(int(v) for v in self.output_size),
                'output_layer': tuple(int(v) for v in self.decode.output_layer),
                'anchor_px': self.decode.anchor.tolist(),
                'config': config, 'cbs_widths': n_cbs}

    @classmethod
    def from_checkpoint(cls, ckpt):
        """Inverse of {'state_dict': model.state_dict(), 'meta': model.checkpoint_meta()}."""
        import argparse
        meta = ckpt['meta']
        anchor_px = torch.tensor(meta['anchor_px'], dtype=torch.float32)
        args = argparse.Namespace(model_type=meta['model_type'], input_size=meta['input_size'],
                                  output_class=meta['output_class'], output_size=tuple(meta['output_size']),
                                  output_layer=tuple(meta['output_layer']),
                                  anchor=(anchor_px / meta['input_size']).tolist())
        net = cls(args, config=meta['config'])
        net.decode.anchor = anchor_px          # the exact pixel values, not input_size * (anchor / input_size)
        net.load_state_dict(ckpt['state_dict'])
        return net

    def __getstate__(self):
        state = self.__dict__.copy()
        state['_plans'] = {}
        state['_train_plans'] = {}
        state['_state'] = None
        state['_hook'] = None
        return state

    def __setstate__(self, state):
        """Unpickling (checkpoints pickle whole modules, train_class.py:262-276): the execution state is
        rebuilt on first use; a checkpoint written by the reference has none of these attributes."""
        super().__setstate__(state)
        self.__dict__.update(_plans={}, _train_plans={}, _state=None, _hook=None)


class _TrainForward(torch.autograd.Function):
    """Autograd node of the whole train-mode forward: backward replays the reverse plan."""

    @staticmethod
    def forward(ctx, x, hook, plan):
        ctx.plan, ctx.x = plan, x
        pred = plan.forward(x)
        pred._odamd_plan = plan
        return pred

    @staticmethod
    def backward(ctx, grad_pred):
        plan = ctx.plan
        if grad_pred.data_ptr() != plan.dpred.data_ptr():
            plan.dpred.copy_(grad_pred)
        plan.backward(ctx.x, reducer=getattr(plan, 'reducer', None))
        return None, None, None
