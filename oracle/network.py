"""torch-CPU fp32 restatement of the YOLOv7 network forward (test infrastructure).

Works on a plain state_dict (name -> tensor) so it never needs the reference's
classes; autograd through these functions gives the oracle gradients.

Follows: cbs `model/layer.py:4-16`, elan `:74-81`, elan_h `:109-118`, mp `:139-145`,
sppcspc `:210-223`, head `:234-236`, decode `:273-293`, deploy `:302-305`,
yolov7.forward `model/yolov7.py:113-146`.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-3        # model/layer.py:9
BN_MOMENTUM = 0.03   # model/layer.py:9


class Net:
    """Functional view of a yolov7 state_dict.

    sd        : dict name -> tensor (parameters may require grad)
    training  : BN uses batch statistics and updates running buffers in `sd`
    taps      : optional dict filled with named intermediate activations
    act_bf16  : emulate the storage precision of the bf16 compute path: convolution weights, the raw
                convolution output and every cbs output are rounded to bf16 (straight-through gradient),
                all arithmetic stays in the tensors' own (float32 / float64) precision.  Max-pool arg-max
                decisions are then taken on the values that path actually pools, and batch-statistics
                BatchNorm sees the rounded raw tensor, as the kernels do.
    """

    def __init__(self, sd, training=False, taps=None, act_bf16=False):
        self.sd = sd
        self.training = training
        self.taps = taps
        self.act_bf16 = act_bf16

    def _q(self, x):
        if not self.act_bf16:
            return x
        return x + (x.to(torch.bfloat16).to(x.dtype) - x).detach()

    def _tap(self, name, x):
        if self.taps is not None:
            self.taps[name] = x
        return x

    # ---- model/layer.py:4-16 -------------------------------------------------------
    def cbs(self, p, x, stride=1):
        w = self._q(self.sd[p + '.conv.weight'])
        k = w.shape[-1]
        y = self._q(F.conv2d(x, w, None, stride=stride, padding=(k - 1) // 2))
        y = F.batch_norm(y, self.sd[p + '.bn.running_mean'], self.sd[p + '.bn.running_var'],
                         self.sd[p + '.bn.weight'], self.sd[p + '.bn.bias'],
                         self.training, BN_MOMENTUM, BN_EPS)
        if self.training and (p + '.bn.num_batches_tracked') in self.sd:
            self.sd[p + '.bn.num_batches_tracked'] += 1
        return self._tap(p, self._q(F.silu(y)))

    def _chain(self, p, x):
        i = 0
        while f'{p}.{i}.conv.weight' in self.sd:
            x = self.cbs(f'{p}.{i}', x)
            i += 1
        return x

    # ---- model/layer.py:74-81 ------------------------------------------------------
    def elan(self, p, x):
        a = self.cbs(p + '.cbs0', x)
        b = self.cbs(p + '.cbs1', x)
        c = self._chain(p + '.sequential2', b)
        d = self._chain(p + '.sequential3', c)
        return self.cbs(p + '.cbs5', torch.cat([a, b, c, d], 1))

    # ---- model/layer.py:109-118 ----------------------------------------------------
    def elan_h(self, p, x):
        outs = [self.cbs(p + '.cbs0', x), self.cbs(p + '.cbs1', x)]
        for i in (2, 3, 4, 5):
            outs.append(self.cbs(f'{p}.cbs{i}', outs[-1]))
        return self.cbs(p + '.cbs7', torch.cat(outs, 1))

    # ---- model/layer.py:139-145 ----------------------------------------------------
    def mp(self, p, x):
        a = self.cbs(p + '.cbs1', F.max_pool2d(x, 2, 2))
        b = self.cbs(p + '.cbs3', self.cbs(p + '.cbs2', x), stride=2)
        return torch.cat([a, b], 1)

    # ---- model/layer.py:210-223 ----------------------------------------------------
    def sppcspc(self, p, x):
        side = self.cbs(p + '.cbs0', x)
        t = self.cbs(p + '.cbs3', self.cbs(p + '.cbs2', self.cbs(p + '.cbs1', x)))
        pools = [F.max_pool2d(t, k, 1, k // 2) for k in (5, 9, 13)]   # all three pool `t`
        y = self.cbs(p + '.cbs9', self.cbs(p + '.cbs8', torch.cat([t] + pools, 1)))
        return self.cbs(p + '.cbs11', torch.cat([y, side], 1))

    # ---- model/layer.py:234-236 ----------------------------------------------------
    def head(self, p, x, n_anchor, k):
        y = F.conv2d(x, self._q(self.sd[p + '.output.weight']), self.sd[p + '.output.bias'])
        b, _, h, w = y.shape
        return y.reshape(b, n_anchor, h, w, k)   # raw reshape, no permute

    # ---- model/yolov7.py:113-146 (decode left to the caller) ---------------------------
    def trunk(self, x, n_class, n_anchor=3):
        k = 5 + n_class
        x = self.cbs('cbs0', x)
        x = self.cbs('cbs1', x, 2)
        x = self.cbs('cbs2', x)
        x = self.cbs('cbs3', x, 2)
        x = self.elan('elan4', x)
        x = self.mp('mp5', x)
        l6 = self.elan('elan6', x)
        x = self.mp('mp7', l6)
        l8 = self.elan('elan8', x)
        x = self.mp('mp9', l8)
        x = self.elan('elan10', x)
        l11 = self.sppcspc('sppcspc11', x)
        x = self.cbs('cbs12', l11)
        x = F.interpolate(x, scale_factor=2.0, mode='nearest')
        x = torch.cat([x, self.cbs('l8_concat', l8)], 1)
        l15 = self.elan_h('elan_h15', x)
        x = self.cbs('cbs16', l15)
        x = F.interpolate(x, scale_factor=2.0, mode='nearest')
        x = torch.cat([x, self.cbs('l6_concat', l6)], 1)
        x = self.elan_h('elan_h19', x)
        h0 = self.head('head0', x, n_anchor, k)
        x = torch.cat([self.mp('mp20', x), l15], 1)
        x = self.elan_h('elan_h22', x)
        h1 = self.head('head1', x, n_anchor, k)
        x = torch.cat([self.mp('mp23', x), l11], 1)
        x = self.elan_h('elan_h25', x)
        h2 = self.head('head2', x, n_anchor, k)
        return [h0, h1, h2]


def decode(heads, input_size, anchor, differentiable=True):
    """Box decode of the three raw head tensors (B,A,H,W,K) -> (B,R,K).

    Follows model/layer.py:273-293: t = sigmoid(raw[...,0:4]);
    cx = (2 t0 - 0.5 + gx) * stride with gx the index along the LAST spatial axis,
    cy likewise with the second-to-last; w = 4 t2^2 * aw, h = 4 t3^2 * ah with
    (aw, ah) = input_size * anchor (computed in float32); channels 4: copied.
    Rows per image: scale0[a,y,x], scale1[...], scale2[...].
    """
    outs = []
    anc = input_size * torch.tensor(anchor, dtype=torch.float32)
    for i, raw in enumerate(heads):
        b, a, h, w, k = raw.shape
        stride = input_size / h
        t = torch.sigmoid(raw[..., 0:4])
        gx = torch.arange(w, dtype=torch.float32).view(1, 1, 1, w)
        gy = torch.arange(h, dtype=torch.float32).view(1, 1, h, 1)
        cx = (2 * t[..., 0] - 0.5 + gx) * stride
        cy = (2 * t[..., 1] - 0.5 + gy) * stride
        bw = 4 * t[..., 2] ** 2 * anc[i, :, 0].view(1, a, 1, 1)
        bh = 4 * t[..., 3] ** 2 * anc[i, :, 1].view(1, a, 1, 1)
        full = torch.cat([torch.stack([cx, cy, bw, bh], -1), raw[..., 4:]], -1)
        outs.append(full.reshape(b, a * h * w, k))
    return torch.cat(outs, 1)


def deploy_sigmoid(pred):
    """model/layer.py:304 -- sigmoid on objectness and class channels."""
    return torch.cat([pred[..., :4], torch.sigmoid(pred[..., 4:])], -1)


def yolov7_forward(sd, x, args, training=False, deploy=False, taps=None, act_bf16=False):
    """Whole reference forward on CPU: trunk -> decode (-> deploy sigmoid)."""
    net = Net(sd, training, taps, act_bf16)
    heads = net.trunk(x, args.output_class, args.output_layer[0])
    out = decode(heads, args.input_size, args.anchor)
    return deploy_sigmoid(out) if deploy else out
