"""torch-CPU restatement of the training-step arithmetic (test infrastructure).

Follows train_class.py:410-427 (loss_class.__call__), :429-460 (iou / L1_L2 / ciou),
:463-474 (focal_loss), :502-521 (lr_adjust), :158-167 (Adam settings),
:477-499 (model_ema).
"""
import math

import torch
import torch.nn.functional as F


def _xywh_min(b):
    return torch.cat([b[:, 0:2] - b[:, 2:4] / 2, b[:, 2:4]], 1)


def ciou(pred, label):
    """train_class.py:449-460: CIoU variant with corner-based distance term."""
    p = _xywh_min(pred)
    l = _xywh_min(label)
    x1 = torch.maximum(p[:, 0], l[:, 0])
    y1 = torch.maximum(p[:, 1], l[:, 1])
    x2 = torch.minimum(p[:, 0] + p[:, 2], l[:, 0] + l[:, 2])
    y2 = torch.minimum(p[:, 1] + p[:, 3], l[:, 1] + l[:, 3])
    inter = torch.clamp(x2 - x1, 0) * torch.clamp(y2 - y1, 0)
    iou = inter / (p[:, 2] * p[:, 3] + l[:, 2] * l[:, 3] - inter)
    ex1 = torch.minimum(p[:, 0], l[:, 0])
    ey1 = torch.minimum(p[:, 1], l[:, 1])
    ex2 = torch.maximum(p[:, 0] + p[:, 2], l[:, 0] + l[:, 2])
    ey2 = torch.maximum(p[:, 1] + p[:, 3], l[:, 1] + l[:, 3])
    dist = ((p[:, 0] - l[:, 0]) ** 2 + (p[:, 1] - l[:, 1]) ** 2) / ((ex2 - ex1) ** 2 + (ey2 - ey1) ** 2)
    v = (4 / (3.1415926 ** 2)) * (torch.atan(l[:, 2] / l[:, 3]) - torch.atan(p[:, 2] / p[:, 3])) ** 2
    with torch.no_grad():
        alpha = v / (v + 1 - iou + 0.00001)
    return iou - dist - alpha * v


def focal(logit, target, alpha=0.5, gamma=2, scale=100):
    """train_class.py:470-474."""
    bce = F.binary_cross_entropy_with_logits(logit, target, reduction='none')
    a = alpha * target + (1 - alpha) * (1 - target)
    return scale * torch.mean(a * (1 - torch.exp(-bce)) ** gamma * bce)


def detection_loss(pred, screen_list, label_expend, label_smooth, n_class):
    """train_class.py:410-427. Returns (loss, frame, confidence, class)."""
    conf_label = torch.full_like(pred[:, :, 4], label_smooth)
    picked = []
    for b, screen in enumerate(screen_list):
        picked.append(pred[b][screen.long()])
        conf_label[b, screen.long()] = 1 - label_smooth
    picked = torch.cat(picked)
    conf_loss = focal(pred[:, :, 4], conf_label)
    if len(picked) == 0:
        frame_loss = torch.tensor(1)
        class_loss = torch.tensor(1)
    else:
        frame_loss = 1 - torch.mean(ciou(picked[:, 0:4], label_expend[:, 0:4]))
        if n_class == 1:
            class_loss = F.binary_cross_entropy_with_logits(picked[:, 5:], label_expend[:, 5:])
        else:
            class_loss = F.cross_entropy(picked[:, 5:], label_expend[:, 5:])
    return frame_loss + conf_loss + class_loss, frame_loss, conf_loss, class_loss


class LrSchedule:
    """train_class.py:502-521; call step() once at optimizer creation and after each step."""

    def __init__(self, lr_start, lr_end_ratio, lr_end_epoch, warmup_ratio, step_epoch, epoch_finished=0):
        self.lr_start = lr_start
        self.lr_end = lr_end_ratio * lr_start
        self.step_all = lr_end_epoch * step_epoch
        self.step_finished = epoch_finished * step_epoch
        self.warmup_step = max(5, int(warmup_ratio * self.step_all))

    def step(self):
        self.step_finished += 1
        n = self.step_finished
        lr = self.lr_end + (self.lr_start - self.lr_end) * math.cos(math.pi / 2 * n / self.step_all)
        if n <= self.warmup_step:
            lr = lr * (0.1 + 0.9 * n / self.warmup_step)
        return max(lr, 0.000001)


def adam_step(p, g, m, v, step, lr, beta1=0.937, beta2=0.999, eps=1e-8, weight_decay=0.0005):
    """One torch.optim.Adam update with coupled L2 (train_class.py:161-162), in place.

    Restates torch's single-tensor Adam: g += wd*p; m = lerp(m, g, 1-b1);
    v = b2*v + (1-b2) g^2; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps).
    """
    g = g + weight_decay * p
    m.lerp_(g, 1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def ema_decay(update_total, decay=0.9999, tau=2000):
    """train_class.py:481."""
    return decay * (1 - math.exp(-update_total / tau))


def ema_update(ema_sd, model_sd, update_total):
    """train_class.py:485-493 (update_total is the already incremented counter)."""
    d = ema_decay(update_total)
    for k, v in ema_sd.items():
        if v.dtype.is_floating_point:
            v.mul_(d).add_((1 - d) * model_sd[k].detach())
