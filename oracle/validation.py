"""Restatement of the reference's validation metric (test infrastructure, never the product path).

Follows train_class.py:66-83 (iou, nms) and :335-365 (metric).  The greedy NMS itself lives in a
dependency that is not installed here: torchvision.ops.nms (the reference does not pin a version).
`nms_corner` restates its published CPU algorithm: candidates in stable descending score order,
areas (x2 - x1) * (y2 - y1), a later box is suppressed iff inter / (area_i + area_j - inter) is
greater than the threshold (a NaN ratio therefore survives).
Parity unpinned: the reference module imports torchvision, cv2 and wandb at load time, none of
which is present, so no output of the reference exists for this path.
"""
import numpy as np

F = np.float32


def nms_corner(boxes, scores, threshold):
    """boxes: (N,4) float32 (x1,y1,x2,y2); returns kept indices, best score first."""
    boxes = np.asarray(boxes, F)
    order = np.argsort(-np.asarray(scores, F), kind='stable')
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    dead = np.zeros(len(boxes), bool)
    keep = []
    with np.errstate(invalid='ignore', divide='ignore'):
        for n, i in enumerate(order):
            if dead[i]:
                continue
            keep.append(i)
            rest = order[n + 1:]
            w = np.maximum(F(0), np.minimum(boxes[i, 2], boxes[rest, 2]) - np.maximum(boxes[i, 0], boxes[rest, 0]))
            h = np.maximum(F(0), np.minimum(boxes[i, 3], boxes[rest, 3]) - np.maximum(boxes[i, 1], boxes[rest, 1]))
            inter = w * h
            ovr = inter / (area[i] + area[rest] - inter)
            dead[rest[ovr > F(threshold)]] = True
    return np.asarray(keep, np.int64)


def iou(pred, label):
    """train_class.py:67-74 on (N,4+) and (1,4+) float32 rows (x_min,y_min,w,h)."""
    x1 = np.maximum(pred[:, 0], label[:, 0])
    y1 = np.maximum(pred[:, 1], label[:, 1])
    x2 = np.minimum(pred[:, 0] + pred[:, 2], label[:, 0] + label[:, 2])
    y2 = np.minimum(pred[:, 1] + pred[:, 3], label[:, 1] + label[:, 3])
    inter = np.maximum(x2 - x1, F(0)) * np.maximum(y2 - y1, F(0))
    with np.errstate(invalid='ignore', divide='ignore'):
        return inter / (pred[:, 2] * pred[:, 3] + label[:, 2] * label[:, 3] - inter)


def nms(pred, iou_threshold):
    """train_class.py:77-83; returns (kept rows, their indices into `pred`)."""
    pred = pred.copy()
    score = pred[:, 4] * pred[:, 5:].max(axis=1)
    pred[:, 2:4] = pred[:, 0:2] + pred[:, 2:4]
    index = nms_corner(pred[:, 0:4], score, F(1 - iou_threshold))
    pred = pred[index]
    pred[:, 2:4] = pred[:, 2:4] - pred[:, 0:2]
    return pred, index


def metric_image(pred, label, confidence_threshold, iou_threshold):
    """One image of train_class.py:339-364.  pred: (R,K) rows (cx,cy,w,h,conf,cls..), label: (L,K).
    Returns (tp, fn, fp, kept rows after NMS and the top-100 cut, their row indices)."""
    pred = np.asarray(pred, F).copy()
    label = np.asarray(label, F).copy()
    pred[:, 0:2] = pred[:, 0:2] - pred[:, 2:4] / F(2)
    label[:, 0:2] = label[:, 0:2] - label[:, 2:4] / F(2)
    rows = np.nonzero(pred[:, 4] > F(confidence_threshold))[0]
    pred = pred[rows]
    if len(pred) == 0:
        return 0, len(label), 0, pred, rows
    pred, index = nms(pred, iou_threshold)
    pred, rows = pred[:100], rows[index][:100]
    if len(label) == 0:
        return 0, 0, len(pred), pred, rows
    pred_cls = np.argmax(pred[:, 5:], axis=1)
    true_cls = np.argmax(label[:, 5:], axis=1)
    tp = 0
    for i, target in enumerate(label):
        hit = (iou(pred, target[None]) > F(iou_threshold)) & (pred_cls == true_cls[i])
        tp += min(int(hit.sum()), 1)
    return tp, len(label) - tp, len(pred) - tp, pred, rows


def metric(pred_batch, label_list, confidence_threshold, iou_threshold):
    tp = fn = fp = 0
    for pred, label in zip(pred_batch, label_list):
        a, b, c, _, _ = metric_image(pred, label, confidence_threshold, iou_threshold)
        tp, fn, fp = tp + a, fn + b, fp + c
    return tp, fn, fp
