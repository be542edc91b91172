"""numpy restatement of the predict.py post-processing (test infrastructure).

Follows predict.py:33-41 (iou), :67-80 (_decode), :82-92 (_nms).  Unlike the
reference, which hands back rows only, every function here also returns the
indices of the surviving rows in the tensor it was given, so kept-index parity
can be asserted bit-exactly.

Tie rule (the reference's np.argsort is unstable, so ties are unpinned there):
descending score, then descending original index.
"""
import numpy as np


def iou_xywh(one, many):
    """IoU of one (x_min,y_min,w,h) box against many, float32, op order of predict.py:35-41."""
    one = one.astype(np.float32)
    many = many.astype(np.float32)
    x1 = np.maximum(one[0], many[:, 0])
    y1 = np.maximum(one[1], many[:, 1])
    x2 = np.minimum(one[0] + one[2], many[:, 0] + many[:, 2])
    y2 = np.minimum(one[1] + one[3], many[:, 1] + many[:, 3])
    inter = np.maximum(x2 - x1, np.float32(0)) * np.maximum(y2 - y1, np.float32(0))
    union = one[2] * one[3] + many[:, 2] * many[:, 3] - inter
    with np.errstate(divide='ignore', invalid='ignore'):
        return inter / union


def score_order(rows):
    """predict.py:83-84 with the tie rule above. rows: (N,K) -> order (N,) int64."""
    score = rows[:, 4] * np.max(rows[:, 5:], axis=1)
    return np.argsort(score, kind='stable')[::-1]


def nms_indices(rows, iou_threshold):
    """Greedy class-agnostic NMS of predict.py:82-92.

    rows: (N,K) float array of (x_min,y_min,w,h,conf,cls...).
    Returns indices into `rows` of the kept boxes, in keep order.
    A candidate survives a kept box iff `iou < threshold` (NaN IoU => suppressed).
    """
    thr = np.float32(iou_threshold)
    order = score_order(rows)
    boxes = rows[order][:, 0:4].astype(np.float32)
    alive = np.arange(len(order))
    keep = []
    while len(alive):
        head = alive[0]
        keep.append(head)
        alive = alive[1:]
        if len(alive) == 0:
            break
        ok = iou_xywh(boxes[head], boxes[alive]) < thr
        alive = alive[ok]
    return order[np.asarray(keep, dtype=np.int64)] if keep else np.zeros((0,), np.int64)


def filter_indices(pred, confidence_threshold):
    """predict.py:68-70. pred: (R,K) decoded+deployed rows (cx,cy,w,h,conf,cls...).

    Returns (cand_rows, cand_idx): rows converted to (x_min,y_min,...) that pass
    `conf > thr` and `x_min > 0 and y_min > 0`, and their indices into `pred`.
    """
    p = pred.astype(np.float32, copy=True)
    p[:, 0:2] = p[:, 0:2] - p[:, 2:4] / 2
    m = (p[:, 4] > np.float32(confidence_threshold)) & (p[:, 0] > 0) & (p[:, 1] > 0)
    idx = np.nonzero(m)[0]
    return p[idx], idx


def predict_decode(pred, image_shape, input_size, confidence_threshold, iou_threshold):
    """Whole predict_class._decode (predict.py:67-80).

    Returns (rows, kept_idx) or (None, empty): rows are (kept,K) float32
    (cx,cy,w,h,...) scaled to the original image, kept_idx index into `pred`.
    """
    cand, idx = filter_indices(pred, confidence_threshold)
    if len(cand) == 0:
        return None, np.zeros((0,), np.int64)
    keep = nms_indices(cand, iou_threshold)
    out = cand[keep].astype(np.float32)
    out[:, 0:2] = out[:, 0:2] + out[:, 2:4] / 2
    xs = np.float32(image_shape[1] / input_size)
    ys = np.float32(image_shape[0] / input_size)
    out[:, 0] *= xs
    out[:, 1] *= ys
    out[:, 2] *= xs
    out[:, 3] *= ys
    return out, idx[keep]


def resize_u8_linear(image, size):
    """cv2.resize(image, (size, size)) for uint8 images, INTER_LINEAR, restated from OpenCV's fixed-point
    scheme (predict.py:61 calls it): taps at (d + 0.5) * scale - 0.5, weights rounded to 11 fractional
    bits, vertical pass (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.
    Parity with cv2 itself is unpinned: cv2 is not installed where this build runs."""
    h0, w0 = image.shape[:2]
    if h0 == size and w0 == size:
        return image.copy()

    def taps(n_dst, n_src):
        f = ((np.arange(n_dst, dtype=np.float32) + np.float32(0.5)) * np.float32(n_src / n_dst) - np.float32(0.5)).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        lo, hi = s < 0, s >= n_src - 1
        s = np.clip(s, 0, n_src - 1)
        f[lo | hi] = 0
        s1 = np.minimum(s + 1, n_src - 1)
        w0_ = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        w1_ = np.rint(f * np.float32(2048)).astype(np.int64)
        return s, s1, w0_, w1_

    x0, x1, a0, a1 = taps(size, w0) if w0 != size else (np.arange(size), np.arange(size), np.full(size, 2048), np.zeros(size, np.int64))
    y0, y1, b0, b1 = taps(size, h0) if h0 != size else (np.arange(size), np.arange(size), np.full(size, 2048), np.zeros(size, np.int64))
    img = image.astype(np.int64)
    hz = img[:, x0] * a0[None, :, None] + img[:, x1] * a1[None, :, None]          # (h0, size, 3)
    r0, r1 = hz[y0], hz[y1]
    v = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)
