"""Restatement of the reference's label -> grid-row assignment (test infrastructure).

Follows train_class.py:537-543 (range_floor / range_upper = decode(-4) / decode(+4)) and
:578-607 (per-label strict containment test, set bookkeeping, nearest-centre conflict rule).
Parity unpinned: the reference runs this inside torch_dataset.__getitem__, which decodes image
files with cv2 (absent here), so no fixture of the reference exists for it.
"""
import torch

from . import network


def ranges(args):
    floor = [torch.full((1, args.output_layer[i], s, s, 4), -4.0) for i, s in enumerate(args.output_size)]
    upper = [torch.full_like(t, 4.0) for t in floor]
    return (network.decode(floor, args.input_size, args.anchor)[0], network.decode(upper, args.input_size, args.anchor)[0])


def assign(label, range_floor, range_upper):
    label_expend = torch.zeros((len(range_floor), label.shape[1]), dtype=torch.float32)
    label_screen = []
    label_set = set()
    for label_ in label:
        cx, cy, w, h = label_[0:4]
        screen = ((range_floor[:, 0] < cx) & (cx < range_upper[:, 0]) & (range_floor[:, 1] < cy) & (cy < range_upper[:, 1])
                  & (range_floor[:, 2] < w) & (w < range_upper[:, 2]) & (range_floor[:, 3] < h) & (h < range_upper[:, 3]))
        index = set(torch.nonzero(screen)[:, 0].tolist())
        both = label_set.intersection(index)
        if both:
            index -= both
            for r in both:
                old = label_expend[r]
                centre = (range_floor[r][0:2] + range_upper[r][0:2]) / 2
                if torch.linalg.norm(old[0:2] - centre) > torch.linalg.norm(label_[0:2] - centre):
                    label_expend[r] = label_
        index = list(index)
        label_expend[index] = label_
        label_screen.extend(index)
        # NOTE: the reference never adds to label_set (train_class.py:581-604), so its conflict branch
        # is dead code and a row claimed twice simply keeps the LAST label written.
    label_screen, _ = torch.sort(torch.tensor(label_screen, dtype=torch.int64))
    return label_screen, label_expend[label_screen]
