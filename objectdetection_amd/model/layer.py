"""Building blocks of the detector, API-compatible with the reference's model/layer.py.

Every class keeps the reference's name, constructor signature and sub-module attribute names
(these are the state_dict keys and the pickle layout), including the pruned `config` forms with
their `config_len` / `last_layer` attributes.  `forward` does not run torch operators: a block
lowers itself (`_emit`) into an execution plan of HIP kernels, see objectdetection_amd/engine.py.

Reference: cbs model/layer.py:4-16, concat :19-27, elan :49-81, elan_h :84-118, mp :121-145,
sppcspc :178-223, head :226-236, decode :263-293, deploy :296-305.
"""
import torch

from .. import engine
from .. import postprocess as _post


def _w(module):
    """Output channels of a block (derived from its convolutions, so that modules restored from a
    reference-format pickle, which carry only the reference's attributes, work as well)."""
    if isinstance(module, cbs):
        return module.conv.out_channels
    if isinstance(module, elan):
        return module.cbs5.conv.out_channels
    if isinstance(module, elan_h):
        return module.cbs7.conv.out_channels
    if isinstance(module, mp):
        return module.cbs1.conv.out_channels + module.cbs3.conv.out_channels
    if isinstance(module, sppcspc):
        return module.cbs11.conv.out_channels
    raise TypeError(type(module))


class _Block(torch.nn.Module):
    def forward(self, x):
        return engine.run_module(self, x)

    def __getstate__(self):
        """Pickling (checkpoints pickle whole modules): execution plans and their buffers stay behind."""
        state = self.__dict__.copy()
        for k in ('_odamd_plans', '_odamd_train_plans', '_odamd_hook', '_odamd_state'):
            state.pop(k, None)
        return state


class cbs(_Block):
    """Conv2d(no bias, 'same' padding) -> BatchNorm2d(eps 1e-3, momentum 0.03) -> SiLU."""

    def __init__(self, in_, out_, kernel_size, stride):
        super().__init__()
        self.conv = torch.nn.Conv2d(in_, out_, kernel_size, stride, (kernel_size - 1) // 2, bias=False)
        self.bn = torch.nn.BatchNorm2d(out_, eps=0.001, momentum=0.03)
        self.silu = torch.nn.SiLU(inplace=True)   # parameter-free; kept so module trees line up

    def _emit(self, pb, x, out=None, pool=False):
        return pb.conv(self, x, out, pool)


class concat(_Block):
    """Channel concatenation.  Inside the other blocks it is free (producers write at channel
    offsets of one buffer); called on its own it copies, like the reference."""

    def __init__(self, dim=1):
        super().__init__()
        self.concat = torch.concat
        self.dim = dim

    def forward(self, x):
        if self.dim != 1 or x[0].dim() != 4:
            return torch.concat(x, dim=self.dim)
        out = torch.empty((x[0].shape[0], sum(t.shape[1] for t in x)) + tuple(x[0].shape[2:]), dtype=x[0].dtype,
                          device=x[0].device)
        c = 0
        for t in x:
            out[:, c:c + t.shape[1]].copy_(t)
            c += t.shape[1]
        return out


def _chain(pb, mods, x, last_out):
    for i, m in enumerate(mods):
        x = m._emit(pb, x, last_out if i == len(mods) - 1 else None)
    return x


class elan(_Block):  # in_ -> out_, same spatial size
    def __init__(self, in_, out_=None, n=1, config=None):
        super().__init__()
        if config is None:
            width = [out_ // 4] * (2 + 2 * n) + [out_]
        else:   # pruned: cbs0, cbs1, n x sequential2, n x sequential3, cbs5
            width = list(config[:3 + 2 * n])
            self.config_len = 3 + 2 * n
            self.last_layer = width[-1]
        self.cbs0 = cbs(in_, width[0], 1, 1)
        self.cbs1 = cbs(in_, width[1], 1, 1)
        self.sequential2 = torch.nn.Sequential(*(cbs(width[1 + i], width[2 + i], 3, 1) for i in range(n)))
        self.sequential3 = torch.nn.Sequential(*(cbs(width[1 + n + i], width[2 + n + i], 3, 1) for i in range(n)))
        self.concat4 = concat()
        self.cbs5 = cbs(width[0] + width[1] + width[1 + n] + width[1 + 2 * n], width[2 + 2 * n], 1, 1)

    def _emit(self, pb, x, out=None):
        n = len(self.sequential2)
        cat, s = pb.buffer([_w(self.cbs0), _w(self.cbs1), _w(self.sequential2[n - 1]), _w(self.sequential3[n - 1])],
                           x.h, x.w)
        self.cbs0._emit(pb, x, s[0])
        self.cbs1._emit(pb, x, s[1])
        _chain(pb, self.sequential2, s[1], s[2])
        _chain(pb, self.sequential3, s[2], s[3])
        return self.cbs5._emit(pb, cat, out)


class elan_h(_Block):  # in_ -> out_, same spatial size
    def __init__(self, in_, out_=None, config=None):
        super().__init__()
        if config is None:
            width = [in_ // 2, in_ // 2] + [in_ // 4] * 4 + [out_]
        else:
            width = list(config[:7])
            self.config_len = 7
            self.last_layer = width[6]
        self.cbs0 = cbs(in_, width[0], 1, 1)
        self.cbs1 = cbs(in_, width[1], 1, 1)
        self.cbs2 = cbs(width[1], width[2], 3, 1)
        self.cbs3 = cbs(width[2], width[3], 3, 1)
        self.cbs4 = cbs(width[3], width[4], 3, 1)
        self.cbs5 = cbs(width[4], width[5], 3, 1)
        self.concat6 = concat()
        self.cbs7 = cbs(sum(width[:6]), width[6], 1, 1)

    def _emit(self, pb, x, out=None):
        stages = [self.cbs0, self.cbs1, self.cbs2, self.cbs3, self.cbs4, self.cbs5]
        cat, s = pb.buffer([_w(m) for m in stages], x.h, x.w)
        self.cbs0._emit(pb, x, s[0])
        self.cbs1._emit(pb, x, s[1])
        for i in range(2, 6):
            stages[i]._emit(pb, s[i - 1], s[i])
        return self.cbs7._emit(pb, cat, out)


class mp(_Block):  # in_ -> out_, half spatial size
    def __init__(self, in_, out_=None, config=None):
        super().__init__()
        if config is None:
            width = [out_ // 2] * 3
        else:
            width = list(config[:3])
            self.config_len = 3
            self.last_layer = width[0] + width[2]
        self.maxpool0 = torch.nn.MaxPool2d(kernel_size=2, stride=2, padding=0, dilation=1)
        self.cbs1 = cbs(in_, width[0], 1, 1)
        self.cbs2 = cbs(in_, width[1], 1, 1)
        self.cbs3 = cbs(width[1], width[2], 3, 2)
        self.concat4 = concat(dim=1)

    def _emit(self, pb, x, out=None):
        """`out`: optional pair of slices (pooled branch, strided branch) of a wider buffer."""
        whole = None
        if out is None:
            whole, out = pb.buffer([_w(self.cbs1), _w(self.cbs3)], x.h // 2, x.w // 2)
        self.cbs1._emit(pb, x, out[0], pool=True)          # max pool folded into the 1x1 conv's load
        self.cbs3._emit(pb, self.cbs2._emit(pb, x), out[1])
        return whole


class sppcspc(_Block):  # in_ -> out_, same spatial size
    def __init__(self, in_, out_=None, config=None):
        super().__init__()
        if config is None:
            width = [in_ // 2] * 6 + [out_]
        else:
            width = list(config[:7])
            self.config_len = 7
            self.last_layer = width[6]
        self.cbs0 = cbs(in_, width[0], 1, 1)
        self.cbs1 = cbs(in_, width[1], 1, 1)
        self.cbs2 = cbs(width[1], width[2], 3, 1)
        self.cbs3 = cbs(width[2], width[3], 1, 1)
        self.MaxPool2d4 = torch.nn.MaxPool2d(kernel_size=5, stride=1, padding=2, dilation=1)
        self.MaxPool2d5 = torch.nn.MaxPool2d(kernel_size=9, stride=1, padding=4, dilation=1)
        self.MaxPool2d6 = torch.nn.MaxPool2d(kernel_size=13, stride=1, padding=6, dilation=1)
        self.concat7 = concat(dim=1)
        self.cbs8 = cbs(4 * width[3], width[4], 1, 1)
        self.cbs9 = cbs(width[4], width[5], 3, 1)
        self.concat10 = concat(dim=1)
        self.cbs11 = cbs(width[0] + width[5], width[6], 1, 1)

    def _emit(self, pb, x, out=None):
        w3 = _w(self.cbs3)
        cat10, (main, side) = pb.buffer([_w(self.cbs9), _w(self.cbs0)], x.h, x.w)
        cat7, pyramid = pb.buffer([w3] * 4, x.h, x.w)
        with pb.side_branch():                            # only cbs11 reads it: beside the cbs1 .. cbs9 chain
            self.cbs0._emit(pb, x, side)
        t = self.cbs2._emit(pb, self.cbs1._emit(pb, x))
        self.cbs3._emit(pb, t, pyramid[0])
        pb.spp(pyramid[0], pyramid[1:])                   # 5/9/13 pools of the SAME tensor
        self.cbs9._emit(pb, self.cbs8._emit(pb, cat7), main)
        return self.cbs11._emit(pb, cat10, out)


class head(_Block):  # in_ -> (batch, layer, output_size, output_size, 5 + output_class)
    def __init__(self, in_, output_size, output_class, layer=3):
        super().__init__()
        self.layer = layer
        self.output_size = output_size
        self.output_class = output_class
        self.output = torch.nn.Conv2d(in_, layer * (5 + output_class), kernel_size=1, stride=1, padding=0)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError('objectdetection_amd blocks run on the GPU only (no CPU fallback)')
        dtype = engine.compute_dtype_for(self)
        plans = engine.module_plans(self)
        key = (tuple(x.shape), dtype, x.device.index)
        plan = plans.get(key)
        if plan is None:
            b, c, h, w = x.shape
            pb = engine.PlanBuilder(b, dtype, x.device)
            op = pb.head(self, pb.from_nchw(c, h, w))
            plan = engine.Plan(pb)
            plan.head_op = op
            plans[key] = plan
        plan.run(x.detach().float().contiguous())
        y = plan.head_op['out']
        # raw reinterpretation of the NCHW result, exactly as the reference reshapes it
        return y.reshape(x.shape[0], self.layer, self.output_size, self.output_size, 5 + self.output_class).clone()


class decode(torch.nn.Module):
    """Raw head outputs -> (cx, cy, w, h, conf, cls...) rows in input-image pixels."""

    def __init__(self, args):
        super().__init__()
        self.anchor = args.input_size * torch.tensor(args.anchor)     # pixels; the kernels take these values
        self.input_size = args.input_size
        self.output_size = args.output_size
        self.output_layer = args.output_layer
        self.stride = [self.input_size / _ for _ in self.output_size]
        self.sigmoid = torch.nn.Sigmoid()         # parameter-free; kept so module trees line up

    def forward(self, x):
        heads = [t.detach().float().contiguous() for t in x]
        return _post.decode(heads, self.input_size, self.anchor, deploy=False)


class deploy(torch.nn.Module):
    """Inference wrapper: model output with sigmoid on objectness and class channels."""

    def __init__(self, model):
        super().__init__()
        self.model = model
        self.sigmoid = torch.nn.Sigmoid()

    def forward(self, x):
        if hasattr(self.model, 'forward_deploy'):
            return self.model.forward_deploy(x)       # sigmoid fused into the decode kernel
        y = self.model(x)
        return _post.deploy_sigmoid_(y)
