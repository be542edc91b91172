"""Training entry, API-compatible with the reference's train_class.py for the parts on the hot
path: train_class (model / optimizer / loss / EMA / data-parallel wiring and the step loop),
loss_class, focal_loss (folded into loss_class), model_ema, lr_adjust, plus a synthetic dataset.

Reference: train_class.py:15-51 (wiring), :66-83 (iou / nms), :158-171 (optimizer), :177-294 (loop),
:296-365 (validation / metric), :397-474 (loss), :477-499 (EMA), :502-521 (LR).  Image decoding /
augmentation (`torch_dataset.__getitem__`), wandb and pruning surgery are outside this build; the
label -> grid assignment rule (train_class.py:578-607) is restated in `assign_labels` for synthetic
labels.

One training step = forward (train-mode BN) -> loss (+ gradient) -> backward -> bucketed gradient
all-reduce (RCCL, overlapped with backward) -> fused Adam + EMA; every arithmetic step is a HIP
kernel of libodamd.so.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import postprocess as _post
from . import train_engine as te
from ._lib import lib, check
from .model.layer import decode


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, plan, screen_list, label_expend, smooth):
        out = plan.loss(screen_list, label_expend, smooth)
        ctx.plan = plan
        return out.clone()

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan
        # d(loss)/d(pred) was produced together with the loss value; only out[0] carries gradient
        plan.dpred.mul_(g[0])      # no host read: the step stays asynchronous
        return plan.dpred, None, None, None, None


class loss_class:
    """loss_class of the reference: __call__(pred, screen_list, label_expend) ->
    (loss, frame_loss, confidence_loss, class_loss)."""

    def __init__(self, args):
        self.device = args.device
        self.input_size = args.input_size
        self.output_size = args.output_size
        self.label_smooth = args.label_smooth

    def __call__(self, pred, screen_list, label_expend=None):
        """`screen_list` may also be a DeviceAssignment (then `label_expend` is not used)."""
        plan = getattr(pred, '_odamd_plan', None)
        if plan is None:
            # an eval-mode prediction (validation, train_class.py:311): value only, no gradient
            if pred.requires_grad:
                raise RuntimeError('loss_class differentiates only the prediction of a training-mode '
                                   'objectdetection_amd yolov7')
            out = te.detection_loss(pred, screen_list, label_expend, self.label_smooth)
            return out[0], out[1], out[2], out[3]
        out = _LossFn.apply(pred, plan, screen_list, label_expend, self.label_smooth)
        return out[0], out[1], out[2], out[3]


class lr_adjust:
    """Per-step cosine decay with linear warm-up (train_class.py:502-521)."""

    def __init__(self, args, step_epoch, epoch_finished):
        self.lr_start = args.lr_start
        self.lr_end = args.lr_end_ratio * args.lr_start
        self.lr_end_epoch = args.lr_end_epoch
        self.step_all = self.lr_end_epoch * step_epoch
        self.step_finished = epoch_finished * step_epoch
        self.warmup_step = max(5, int(args.warmup_ratio * self.step_all))

    def __call__(self, optimizer):
        self.step_finished += 1
        lr = te.lr_schedule(self.step_finished, self.lr_start, self.lr_end, self.step_all, self.warmup_step)
        for group in optimizer.param_groups:
            group['lr'] = lr
        return optimizer


class FusedAdam:
    """torch.optim.Adam(lr, betas=(0.937, 0.999), weight_decay) over the model's flat buffers, fused
    with the EMA update; `step()` is one kernel launch (+ one for the EMA of the BN buffers)."""

    def __init__(self, model, lr, betas=(0.937, 0.999), weight_decay=0.0, eps=1e-8):
        self.state = model.train_state()
        self.params = list(model.parameters())
        self.param_groups = [{'lr': lr, 'betas': betas, 'weight_decay': weight_decay, 'eps': eps}]
        self.grad_scale = 1.0
        self.ema = False

    def step(self):
        g = self.param_groups[0]
        te.optimizer_step(self.state, g['lr'], g['weight_decay'], g['betas'], g['eps'], self.grad_scale, self.ema)

    def zero_grad(self):
        self.state.grad.zero_()

    def state_dict(self):
        s = self.state
        return {'step': s.step, 'exp_avg': s.exp_avg.clone(), 'exp_avg_sq': s.exp_avg_sq.clone(),
                'param_groups': [dict(g) for g in self.param_groups]}

    def load_state_dict(self, d):
        """Takes this class's own state_dict, or torch.optim.Adam's (what a reference checkpoint holds,
        train_class.py:166-167: per-parameter entries in model.parameters() order)."""
        s = self.state
        if 'state' in d:
            step = 0
            for i, p in enumerate(self.params):
                e = d['state'].get(i)
                if e is None:
                    continue
                o = s.offsets[id(p)]
                s.exp_avg[o:o + p.numel()].copy_(e['exp_avg'].reshape(-1))
                s.exp_avg_sq[o:o + p.numel()].copy_(e['exp_avg_sq'].reshape(-1))
                step = max(step, int(e['step']))
            s.step = step
            return
        s.step = d['step']
        s.exp_avg.copy_(d['exp_avg'])
        s.exp_avg_sq.copy_(d['exp_avg_sq'])


class model_ema:
    """Exponential moving average of every floating state_dict entry (train_class.py:477-499).
    The averages live in flat buffers updated inside the optimizer launch; `ema_model` builds a
    module holding them on demand (saving / validation)."""

    def __init__(self, model, decay=0.9999, tau=2000, update_total=0):
        self.model = model.module if hasattr(model, 'module') else model
        self.state = self.model.train_state()
        self.state.enable_ema()
        self.state.ema_updates = update_total

    @property
    def update_total(self):
        return self.state.ema_updates

    @update_total.setter
    def update_total(self, v):
        self.state.ema_updates = v

    def update(self, model=None):
        """No-op: the EMA is applied by FusedAdam.step() in the same launch as the parameters."""

    @property
    def ema_model(self):
        import copy
        st = self.state
        twin = copy.deepcopy(self.model).eval()
        with torch.no_grad():
            for (n, p), (_, q) in zip(self.model.named_parameters(), twin.named_parameters()):
                o = st.param_names[n]
                q.copy_(st.ema_param[o:o + p.numel()].view(p.shape))
                q.requires_grad_(False)
            for (n, b), (_, c) in zip(self.model.named_buffers(), twin.named_buffers()):
                if b.dtype.is_floating_point:
                    o = st.buf_names[n]
                    c.copy_(st.ema_buffers[o:o + b.numel()].view(b.shape))
        return twin


# ---------------------------------------------------------------------------------------------
# labels
# ---------------------------------------------------------------------------------------------
def prediction_ranges(args):
    """Box range every grid row can express: decode(-4) .. decode(+4) (train_class.py:537-543)."""
    anchor = args.input_size * torch.tensor(args.anchor, dtype=torch.float32)
    lo, hi = [], []
    for i, hw in enumerate(args.output_size):
        stride = args.input_size / hw
        gy, gx = torch.meshgrid(torch.arange(hw, dtype=torch.float32), torch.arange(hw, dtype=torch.float32),
                                indexing='ij')
        for t, dst in ((torch.sigmoid(torch.tensor(-4.0)), lo), (torch.sigmoid(torch.tensor(4.0)), hi)):
            rows = []
            for a in range(args.output_layer[i]):
                cx = (2 * t - 0.5 + gx) * stride
                cy = (2 * t - 0.5 + gy) * stride
                w = torch.full_like(cx, float(4 * t ** 2 * anchor[i, a, 0]))
                h = torch.full_like(cx, float(4 * t ** 2 * anchor[i, a, 1]))
                rows.append(torch.stack([cx, cy, w, h], -1).reshape(-1, 4))
            dst.append(torch.cat(rows))
    return torch.cat(lo), torch.cat(hi)


def assign_labels(label, range_floor, range_upper):
    """Label -> grid-row assignment exactly as train_class.py:578-607 behaves: a label matches every
    row whose expressible (cx, cy, w, h) range strictly contains it.  The reference's conflict
    branch never fires (its `label_set` is never filled), so a row matched by several labels appears
    once per label in the index list and carries the LAST matching label every time; that is
    reproduced here because the loss counts such rows once per occurrence.
    label: (L, K) float32 rows (cx,cy,w,h,conf,cls...).  Returns (sorted row indices int64 with
    duplicates, label rows aligned with them)."""
    k = label.shape[1]
    box = label[:, None, 0:4]
    hit = ((range_floor[None] < box) & (box < range_upper[None])).all(dim=2)        # (L, R)
    li, rows = hit.nonzero(as_tuple=True)
    if len(rows) == 0:
        return torch.zeros(0, dtype=torch.int64), torch.zeros((0, k), dtype=torch.float32)
    last = torch.full((hit.shape[1],), -1, dtype=torch.int64)
    last[rows] = li                      # nonzero() is ordered by label, so the last write wins
    last.scatter_reduce_(0, rows, li, 'amax', include_self=True)
    rows_sorted, _ = torch.sort(rows)
    return rows_sorted, label[last[rows_sorted]]


class DeviceAssignment:
    """Result of `device_labels`: positives of a batch, all on the GPU (no host copy of the count)."""

    def __init__(self, screen, screen_img, label, n_pos, capacity):
        self.screen, self.screen_img, self.label, self.n_pos, self.capacity = screen, screen_img, label, n_pos, capacity

    def per_image(self, batch):
        """Host view in the reference's format: (list of int64 row vectors, concatenated label rows)."""
        n = int(self.n_pos[0])
        if int(self.n_pos[1]) > self.capacity:
            raise RuntimeError(f'label assignment overflow: {int(self.n_pos[1])} entries, capacity {self.capacity}')
        rows, img = self.screen[:n].cpu().long(), self.screen_img[:n].cpu()
        return [rows[img == b] for b in range(batch)], self.label[:n].cpu()


class device_labels:
    """Label -> grid-row assignment of torch_dataset (train_class.py:537-543, :578-607) for whole
    batches on the GPU (odamd_assign_labels).  __call__(labels) takes one (L_b, K) float32 tensor per
    image (cx, cy, w, h, conf, classes..., already smoothed) and returns a DeviceAssignment that
    `loss_class` accepts in place of (screen_list, label_expend)."""

    def __init__(self, args, lmax=64, capacity=None):
        from ._lib import DecodeDesc
        self.args, self.lmax = args, lmax
        self.k = 5 + args.output_class
        d = DecodeDesc()
        anc = (args.input_size * torch.tensor(args.anchor, dtype=torch.float32)).tolist()
        for i, hw in enumerate(args.output_size):
            d.h[i] = d.w[i] = hw
            d.stride[i] = args.input_size / hw
            for j in range(args.output_layer[i]):
                d.anchor_px[i][j][0], d.anchor_px[i][j][1] = anc[i][j][0], anc[i][j][1]
        d.n_anchor, d.kdim, d.deploy = args.output_layer[0], self.k, 0
        self.desc = d
        self.sig_lo = float(torch.sigmoid(torch.tensor(-4.0)))
        self.sig_hi = float(torch.sigmoid(torch.tensor(4.0)))
        self.capacity = capacity

    def __call__(self, labels, device='cuda'):
        import ctypes as C
        from ._lib import lib, check
        b = len(labels)
        cap = self.capacity or b * self.lmax * 64
        packed = torch.zeros((b, self.lmax, self.k), dtype=torch.float32)
        counts = torch.zeros(b, dtype=torch.int32)
        for i, lab in enumerate(labels):
            if len(lab) > self.lmax:
                raise ValueError(f'{len(lab)} labels in one image, lmax is {self.lmax}')
            packed[i, :len(lab)] = lab
            counts[i] = len(lab)
        packed, counts = packed.to(device, non_blocking=True), counts.to(device, non_blocking=True)
        screen = torch.empty(cap, dtype=torch.int32, device=device)
        simg = torch.empty(cap, dtype=torch.int32, device=device)
        out = torch.empty((cap, self.k), dtype=torch.float32, device=device)
        n_pos = torch.zeros(2, dtype=torch.int32, device=device)
        totals = torch.empty(b, dtype=torch.int32, device=device)
        self.desc.batch = b
        check(lib.odamd_assign_labels(C.byref(self.desc), self.sig_lo, self.sig_hi, packed.data_ptr(), counts.data_ptr(),
                                      self.lmax, screen.data_ptr(), simg.data_ptr(), out.data_ptr(), cap, n_pos.data_ptr(),
                                      totals.data_ptr(), te._stream()), 'assign_labels')
        keep = (packed, counts, totals)        # alive until the stream has consumed them
        res = DeviceAssignment(screen, simg, out, n_pos, cap)
        res._keep = keep
        return res


class synthetic_dataset(torch.utils.data.Dataset):
    """Synthetic images and labels: uniform-noise images; per image `boxes` boxes with cx,cy ~
    U(0.1 S, 0.9 S), w,h ~ U(20, 200), uniform class; labels smoothed like train_class.py:573-577."""

    def __init__(self, args, length, boxes=8, seed=999):
        self.args, self.length, self.boxes, self.seed = args, length, boxes, seed
        self.range_floor, self.range_upper = prediction_ranges(args)

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        a = self.args
        g = torch.Generator().manual_seed(self.seed + index)
        s, c, ls = a.input_size, a.output_class, a.label_smooth
        image = torch.rand(3, s, s, generator=g)
        frame = torch.cat([torch.rand(self.boxes, 2, generator=g) * 0.8 * s + 0.1 * s,
                           torch.rand(self.boxes, 2, generator=g) * 180 + 20], 1)
        cls = torch.full((self.boxes, c), ls)
        cls[torch.arange(self.boxes), torch.randint(0, c, (self.boxes,), generator=g)] = 1 - ls
        label = torch.cat([frame, torch.full((self.boxes, 1), 1 - ls), cls], 1).float()
        screen, expend = assign_labels(label, self.range_floor, self.range_upper)
        return image, screen, expend, label

    @staticmethod
    def collate_fn(items):
        return (torch.stack([i[0] for i in items]), [i[1] for i in items], torch.cat([i[2] for i in items]),
                [i[3] for i in items])


# ---------------------------------------------------------------------------------------------
# driver
# ---------------------------------------------------------------------------------------------
class train_class:
    """Wiring and step loop of the reference's train_class (train_class.py:15-51, :177-294)."""

    def __init__(self, args, dataloader=None, val_dataloader=None):
        if not torch.cuda.is_available():
            raise RuntimeError('objectdetection_amd.train_class needs a GPU: there is no CPU fallback')
        self.args = args
        self.model_dict = self.model_load()
        self.model_dict['model'] = self.model_dict['model'].to(args.device)
        self.model = self.model_dict['model']
        # --amp (train_class.py:29, fp16 autocast in the reference) selects the bf16 mixed-precision step
        self.model.train_dtype = 'bf16' if getattr(args, 'amp', False) else 'f32'
        self.train_dataloader = dataloader if dataloader is not None else self.dataloader_load()
        self.val_dataloader = val_dataloader if val_dataloader is not None else self.dataloader_load(val=True)
        self.optimizer, self.optimizer_adjust = self.optimizer_load()
        self.loss = loss_class(args)
        self.ema = None
        if args.local_rank == 0 and args.ema:
            self.ema = model_ema(self.model)
            self.ema.update_total = self.model_dict['ema_update']
            self.optimizer.ema = True
        self.reducer = self.parallel = None
        if args.distributed:
            # every rank starts from rank 0's parameters and buffers (DDP constructor broadcast)
            self.parallel = te.DataParallel(self.model.train_state(), n_buckets=3,
                                            broadcast_buffers=getattr(args, 'broadcast_buffers', True))
            self.reducer = self.parallel.reducer
            self.optimizer.grad_scale = 1.0 / self.reducer.world

    def model_load(self):
        args = self.args
        if os.path.exists(getattr(args, 'weight_path', '')):
            from .predict import load_checkpoint
            model_dict = load_checkpoint(args.weight_path)  # weights-only form, or the reference's pickled module
            for k, v in (('epoch_finished', 0), ('optimizer_state_dict', None), ('ema_update', 0), ('standard', 0)):
                model_dict.setdefault(k, v)
            for p in model_dict['model'].parameters():
                p.requires_grad_(True)
            if args.weight_again:
                model_dict.update(epoch_finished=0, optimizer_state_dict=None, ema_update=0, standard=0)
            return model_dict
        from .model.yolov7 import yolov7
        return {'model': yolov7(args), 'epoch_finished': 0, 'optimizer_state_dict': None, 'ema_update': 0,
                'standard': 0}

    # ---- train_class.py:66-83 ---------------------------------------------------------------------
    @staticmethod
    def iou(pred, label):
        """IoU of (x_min,y_min,w,h) CUDA rows, `label` usually a single row (train_class.py:67-74)."""
        a = pred[:, 0:4].float().contiguous()
        b = label[:, 0:4].float().contiguous().to(a.device)
        n = max(len(a), len(b))
        out = torch.empty(n, dtype=torch.float32, device=a.device)
        check(lib.odamd_iou_pairs(a.data_ptr(), len(a), b.data_ptr(), len(b), out.data_ptr(), n, te._stream()),
              'iou_pairs')
        return out

    @staticmethod
    def nms(pred, iou_threshold):
        """(N,K) CUDA rows (x_min,y_min,w,h,conf,cls..) -> kept rows, best score first (train_class.py:77-83).
        Unlike the reference, `pred` itself is left unchanged."""
        if len(pred) == 0:
            return pred
        res = _post.metric_nms_rows(pred.float(), iou_threshold)
        return res.rows[0, :int(res.count[0])]

    def dataloader_load(self, val=False):
        args = self.args
        n = getattr(args, 'synthetic_images', args.batch * 8)
        if val:
            ds = synthetic_dataset(args, getattr(args, 'synthetic_val_images', args.batch * 2), seed=4242)
            return torch.utils.data.DataLoader(ds, batch_size=args.batch, shuffle=False, drop_last=False,
                                               collate_fn=ds.collate_fn, num_workers=getattr(args, 'num_worker', 0))
        ds = synthetic_dataset(args, n)
        sampler = torch.utils.data.distributed.DistributedSampler(ds) if args.distributed else None
        return torch.utils.data.DataLoader(ds, batch_size=args.batch, shuffle=sampler is None, drop_last=True,
                                           sampler=sampler, collate_fn=ds.collate_fn,
                                           num_workers=getattr(args, 'num_worker', 0))

    def optimizer_load(self):
        args = self.args
        wd = args.r_value if args.regularization == 'L2' else 0.0
        opt = FusedAdam(self.model, args.lr_start, (0.937, 0.999), wd)
        if self.model_dict['optimizer_state_dict'] is not None:
            opt.load_state_dict(self.model_dict['optimizer_state_dict'])
        step_epoch = max(len(self.train_dataloader), 1)
        adjust = lr_adjust(args, step_epoch, self.model_dict['epoch_finished'])
        return adjust(opt), adjust

    def train_step(self, image_batch, screen_list, label_expend):
        """One step of train_class.py:196-218; returns the (4,) device tensor of losses (no host sync)."""
        args = self.args
        model = self.model
        image_batch = image_batch.to(args.device, non_blocking=True)
        plan = model.train_plan(image_batch.shape[0], image_batch.shape[2])
        plan.reducer = self.reducer
        pred = model(image_batch)
        if self.parallel is not None:
            # DistributedDataParallel's default broadcast_buffers=True (train_class.py:38-40 does not override it):
            # rank 0's BatchNorm statistics and counters replace everyone's before the NEXT forward; issued here,
            # the one collective runs behind this step's backward pass
            self.parallel.after_forward()
        losses = self.loss(pred, screen_list, label_expend)
        self.optimizer.zero_grad()
        losses[0].backward()
        if self.parallel is not None:
            self.parallel.finish()
        self.optimizer.step()
        self.optimizer = self.optimizer_adjust(self.optimizer)
        return torch.stack(losses).detach()

    # ---- train_class.py:296-365 -------------------------------------------------------------------
    def metric(self, pred_batch, label_list, confidence_threshold, iou_threshold):
        """(tp, fn, fp) of one batch.  pred_batch: (B,R,K) CUDA prediction of the eval-mode model
        (cx,cy,w,h,raw conf,raw cls..; not modified, the reference shifts it in place); label_list:
        one (L_b,K) tensor per image (cx,cy,w,h,conf,cls..).  Candidate selection, NMS, the top-100
        cut and the matching all run on the device; three integers come back."""
        return tuple(int(v) for v in self.metric_device(pred_batch, label_list, confidence_threshold,
                                                         iou_threshold).sum(0).tolist())

    def metric_device(self, pred_batch, label_list, confidence_threshold, iou_threshold):
        """Same, left on the device: (B,3) int32 rows (tp, fn, fp) per image, no host synchronisation."""
        dev = pred_batch.device
        b, _, k = pred_batch.shape
        lmax = max(1, max((len(l) for l in label_list), default=1))
        packed = torch.zeros((b, lmax, k), dtype=torch.float32)
        counts = torch.zeros(b, dtype=torch.int32)
        for i, lab in enumerate(label_list):
            packed[i, :len(lab)] = lab
            counts[i] = len(lab)
        packed, counts = packed.to(dev, non_blocking=True), counts.to(dev, non_blocking=True)
        res = _post.metric_nms(pred_batch.float(), confidence_threshold, iou_threshold, max_keep=100)
        return _post.metric_match(res, packed, counts, iou_threshold)

    def validation(self):
        """One pass over the validation loader with the EMA weights when enabled: mean loss, precision,
        recall and the reference's 'm_ap' = precision * recall (train_class.py:296-333)."""
        args = self.args
        with torch.no_grad():
            # cached inference plans re-fold their weights by themselves when a training step has
            # written the parameters (engine.touch_weights), so the model's own plan is reused
            model = self.ema.ema_model.eval() if self.ema else self.model.eval()
            loss_sum = torch.zeros((), device=args.device)
            counts = torch.zeros(3, dtype=torch.int64, device=args.device)
            steps = 0
            for image_batch, screen_list, label_expend, label_list in self.val_dataloader:
                image_batch = image_batch.to(args.device, non_blocking=True)
                pred_batch = model(image_batch)
                loss_sum += self.loss(pred_batch, screen_list, label_expend)[0]
                counts += self.metric_device(pred_batch, label_list, args.confidence_threshold,
                                             args.iou_threshold).sum(0)
                steps += 1
            val_loss = float(loss_sum) / max(steps, 1)                       # the pass's only host reads
            tp, fn, fp = counts.tolist()
            precision = tp / (tp + fp + 1e-6)
            recall = tp / (tp + fn + 1e-6)
            m_ap = precision * recall
            if getattr(args, 'print_info', True):
                print(f'| val | val_loss:{val_loss:.4f} | threshold:{args.confidence_threshold:.2f}+'
                      f'{args.iou_threshold:.2f} | val_precision:{precision:.4f} | val_recall:{recall:.4f} |'
                      f' val_m_ap:{m_ap:.4f} |')
        self.model.train()
        return val_loss, precision, recall, m_ap

    @staticmethod
    def save_checkpoint(model_dict, path):
        """Writes `path` in the reference's format (train_class.py:262-276: the dict with the pickled module)
        and, next to it, `<path minus extension>.weights.pt`: the same dict with the module replaced by its
        state_dict + rebuild data, loadable with torch.load(weights_only=True)."""
        torch.save(model_dict, path)
        model = model_dict['model']
        # the optimizer state is tensors and plain numbers: it stays, so that resuming from the light file
        # continues Adam's moments and step count like the reference does (train_class.py:166-167)
        light = {k: v for k, v in model_dict.items() if k != 'model'}
        light['state_dict'] = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        light['meta'] = model.checkpoint_meta()
        torch.save(light, os.path.splitext(path)[0] + '.weights.pt')

    def train(self):
        args = self.args
        history = []
        for epoch in range(self.model_dict['epoch_finished'] + 1, args.epoch + 1):
            self.model.train()
            sums, steps = torch.zeros(4, device=args.device), 0
            sampler = getattr(self.train_dataloader, 'sampler', None)
            if hasattr(sampler, 'set_epoch'):
                sampler.set_epoch(epoch)
            for image_batch, screen_list, label_expend, _ in self.train_dataloader:
                sums += self.train_step(image_batch, screen_list, label_expend)   # accumulated on the device
                steps += 1
            means = (sums / max(steps, 1)).tolist()                                # one host read per epoch
            history.append(means)
            if args.local_rank == 0 and getattr(args, 'print_info', True):
                print(f'| train | epoch:{epoch} | train_loss:{means[0]:.4f} | frame_loss:{means[1]:.4f} |'
                      f' confidence_loss:{means[2]:.4f} | class_loss:{means[3]:.4f} |'
                      f' lr:{self.optimizer.param_groups[0]["lr"]:.6f} |')
            if args.local_rank == 0:
                validate = len(self.val_dataloader) > 0 and hasattr(args, 'confidence_threshold')
                if validate:                                   # one rank validates (train_class.py:258)
                    val_loss, precision, recall, m_ap = self.validation()
                    self.model_dict.update(val_loss=val_loss, val_precision=precision, val_recall=recall, val_m_ap=m_ap)
                self.model_dict['epoch_finished'] = epoch
                self.model_dict['train_loss'] = means[0]
                self.model_dict['ema_update'] = self.ema.update_total if self.ema else self.model_dict['ema_update']
                save, best = getattr(args, 'save_path', None), getattr(args, 'save_best', None)
                is_best = validate and m_ap >= self.model_dict['standard'] and m_ap >= 0.25   # train_class.py:274
                if is_best:
                    self.model_dict['standard'] = m_ap
                if (save and (epoch % getattr(args, 'save_epoch', 5) == 0 or epoch == args.epoch)) or (best and is_best):
                    out = dict(self.model_dict)
                    out['model'] = self.ema.ema_model if self.ema else self.model
                    out['optimizer_state_dict'] = self.optimizer.state_dict()
                    if save and (epoch % getattr(args, 'save_epoch', 5) == 0 or epoch == args.epoch):
                        self.save_checkpoint(out, save)
                    if best and is_best:
                        self.save_checkpoint(out, best)
            if args.distributed:
                torch.distributed.barrier()
        return history
