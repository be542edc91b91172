"""Known-answer checks of the validation oracle (CPU): the restated corner-box NMS and the metric's
counting, on cases small enough to verify by hand."""
import numpy as np

from oracle import validation as oval


def test_nms_corner_by_hand():
    # boxes 0 and 1 overlap with IoU 81/119 = 0.68; box 2 is far away; box 3 equals box 0
    boxes = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [50, 50, 60, 60], [0, 0, 10, 10]], np.float32)
    scores = np.array([0.9, 0.8, 0.7, 0.9], np.float32)
    assert oval.nms_corner(boxes, scores, 0.5).tolist() == [0, 2]          # tie 0/3: the lower index leads
    assert oval.nms_corner(boxes, scores, 0.7).tolist() == [0, 1, 2]       # 0.68 is not > 0.7; IoU 1 is
    assert oval.nms_corner(boxes, scores, 1.0).tolist() == [0, 3, 1, 2]    # nothing exceeds 1
    zero = np.array([[5, 5, 5, 5], [5, 5, 5, 5]], np.float32)              # 0/0: NaN never suppresses
    assert oval.nms_corner(zero, np.array([1, 2], np.float32), 0.5).tolist() == [1, 0]


def test_metric_counts_by_hand():
    k = 7
    pred = np.zeros((4, k), np.float32)
    pred[:, 0:4] = [[50, 50, 20, 20], [51, 51, 20, 20], [200, 200, 40, 40], [300, 300, 10, 10]]
    pred[:, 4] = [2.0, 1.5, 3.0, -1.0]                      # the last row fails conf > 0.5
    pred[:, 5:] = [[1, 0], [1, 0], [0, 1], [1, 0]]
    label = np.zeros((3, k), np.float32)
    label[:, 0:4] = [[50, 50, 20, 20], [200, 200, 40, 40], [300, 300, 10, 10]]
    label[:, 5:] = [[1, 0], [1, 0], [1, 0]]                 # label 1 has the wrong class for its box
    # iou_threshold 0.5 -> NMS threshold 0.5: rows 0/1 (IoU 0.82) collapse to row 0; kept rows 2, 0
    tp, fn, fp, kept, rows = oval.metric_image(pred, label, 0.5, 0.5)
    assert rows.tolist() == [2, 0] and (tp, fn, fp) == (1, 2, 1)
    assert oval.metric([pred, pred[:0]], [label, label], 0.5, 0.5) == (1, 5, 1)
    assert oval.metric([pred], [label[:0]], 0.5, 0.5) == (0, 0, 2)
