"""How many columns of a 64-column tile survive the wave-level windows of the IoU mask kernel on the bench
distribution: the tests in use (area both ways, shape, bounding box of the wave's rows) and, added, the
centre-distance window (|cx_a - cx_c| <= g (w_a + w_c), g = (1 - t) / (2 (1 + t)), per axis).  numpy on one image."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import engine
from objectdetection_amd.model.yolov7 import yolov7
from oracle import patterns

t = 0.5
args = patterns.default_args(input_size=640, output_class=80)
torch.manual_seed(999)
m = yolov7(args).cuda().eval()
engine.set_compute_dtype('bf16')
x = torch.rand(64, 3, 640, 640, generator=torch.Generator().manual_seed(999)).cuda()
pred = m.forward_deploy(x[:2])[1].float().cpu().numpy()          # (rows, 85): cx, cy, w, h, conf, classes
conf = pred[:, 4]
keep = conf > 0.5
p = pred[keep]
xmin, ymin = p[:, 0] - p[:, 2] / 2, p[:, 1] - p[:, 3] / 2
ok = (xmin > 0) & (ymin > 0)
p, xmin, ymin = p[ok], xmin[ok], ymin[ok]
score = p[:, 4] * p[:, 5:].max(1)
order = np.argsort(-score, kind='stable')
bx, by, bw, bh = xmin[order], ymin[order], p[order, 2], p[order, 3]
n = len(bx)
print('candidates', n)
x2, y2 = bx + bw, by + bh
area = bw * bh
k_lo = t / (1 + t)
g = (1 - t) / (2 * (1 + t)) * 1.01
cx, cy = bx + bw / 2, by + bh / 2


def spread(v):
    v = v.astype(np.uint64)
    out = np.zeros_like(v)
    for b in range(10):
        out |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(2 * b)
    return out


cls = (area.astype(np.float32).view(np.uint32) >> 22).astype(np.uint64)
mort = spread(np.clip(cx / 16, 0, 1023).astype(np.uint32)) | (spread(np.clip(cy / 16, 0, 1023).astype(np.uint32)) << np.uint64(1))
tot_old = tot_new = tiles = 0
for blk in range(0, min(n, 4096), 1024):
    idx = np.arange(blk, min(blk + 1024, n))
    key = (cls[idx] << np.uint64(52)) | (mort[idx] << np.uint64(32)) | idx.astype(np.uint64)
    perm = idx[np.argsort(key)]
    for w0 in range(0, len(perm), 64):
        rows = perm[w0:w0 + 64]
        wmin = (k_lo * area[rows]).min()
        wmax = area[rows].max()
        wmw, wmh = bw[rows].max(), bh[rows].max()
        bx0, by0, bx1, by1 = bx[rows].min(), by[rows].min(), x2[rows].max(), y2[rows].max()
        cxa, cxb, cya, cyb = cx[rows].min(), cx[rows].max(), cy[rows].min(), cy[rows].max()
        first = rows.min() // 64 * 64
        c = np.arange(first, n)
        ck = k_lo * area[c]
        old = ~(area[c] < wmin) & ~(wmax < ck) & ~(np.minimum(wmw, bw[c]) * np.minimum(wmh, bh[c]) < wmin + ck) & \
            ~(x2[c] < bx0) & ~(bx[c] > bx1) & ~(y2[c] < by0) & ~(by[c] > by1)
        rx, ry = g * (wmw + bw[c]), g * (wmh + bh[c])
        new = old & ~(cx[c] < cxa - rx) & ~(cx[c] > cxb + rx) & ~(cy[c] < cya - ry) & ~(cy[c] > cyb + ry)
        tot_old += old.sum()
        tot_new += new.sum()
        tiles += (len(c) + 63) // 64
print(f'tiles {tiles}: surviving columns per tile {tot_old / tiles:.1f} now, {tot_new / tiles:.1f} with the centre window')
