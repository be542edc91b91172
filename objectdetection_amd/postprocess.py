"""Device-side box decode and post-processing (host wrappers over libodamd.so).

Mirrors `decode.forward` / `deploy.forward` (model/layer.py:273-305) and
`predict_class._decode/_nms` (predict.py:67-92); torch is used only to own device memory
and name the current HIP stream.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import lib, check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def decode_desc(heads, input_size, anchor, deploy):
    """Build an odamd_decode_desc for three raw head tensors (B, A, H, W, K) float32.
    `anchor`: the relative anchors of the args Namespace (nested tuples), or a tensor that already holds
    them in pixels (`decode.anchor` = input_size * tensor(args.anchor), model/layer.py:266)."""
    d = _lib.DecodeDesc()
    b, a, _, _, k = heads[0].shape
    if torch.is_tensor(anchor):
        anc = anchor.detach().float().cpu().tolist()
    else:
        anc = (input_size * torch.tensor(anchor, dtype=torch.float32)).tolist()   # model/layer.py:266
    for i, hd in enumerate(heads):
        if hd.dtype != torch.float32 or not hd.is_contiguous() or not hd.is_cuda:
            raise ValueError('head tensors must be contiguous float32 CUDA tensors')
        d.head[i] = hd.data_ptr()
        d.h[i], d.w[i] = hd.shape[2], hd.shape[3]
        d.stride[i] = input_size / hd.shape[2]                                # model/layer.py:270
        for j in range(a):
            d.anchor_px[i][j][0] = anc[i][j][0]
            d.anchor_px[i][j][1] = anc[i][j][1]
    d.n_anchor, d.batch, d.kdim, d.deploy = a, b, k, int(deploy)
    return d


def decode(heads, input_size, anchor, deploy=False):
    """(B,A,H,W,K)x3 raw heads -> (B,R,K) decoded rows; deploy=True also sigmoids channels 4:."""
    d = decode_desc(heads, input_size, anchor, deploy)
    rows = sum(h.shape[1] * h.shape[2] * h.shape[3] for h in heads)
    out = torch.empty((d.batch, rows, d.kdim), dtype=torch.float32, device=heads[0].device)
    check(lib.odamd_decode(C.byref(d), _ptr(out), _stream()), 'decode')
    return out


def deploy_sigmoid_(pred):
    """In-place sigmoid of channels 4: of a contiguous float32 (..., K) CUDA tensor (model/layer.py:304)."""
    if pred.dtype != torch.float32 or not pred.is_contiguous() or not pred.is_cuda:
        raise ValueError('pred must be a contiguous float32 CUDA tensor')
    k = pred.shape[-1]
    check(lib.odamd_deploy_sigmoid(_ptr(pred), pred.numel() // k, k, _stream()), 'deploy_sigmoid')
    return pred


def decode_backward(heads, input_size, anchor, grad_out):
    d = decode_desc(heads, input_size, anchor, False)
    grads = [torch.empty_like(h) for h in heads]
    arr = (C.c_void_p * 3)(*[g.data_ptr() for g in grads])
    check(lib.odamd_decode_backward(C.byref(d), _ptr(grad_out.contiguous()), C.byref(arr), _stream()), 'decode_backward')
    return grads


class HostStager:
    """Page-locked host buffers for the kept rows of whole batches, `slots` of them used in turn: a device-to-host
    copy from pageable memory runs at a fraction of the link rate and blocks, one into page-locked memory is
    asynchronous and runs at the link rate.  Arrays handed out without a copy are views of a slot: they stay valid
    until the slot comes round again (`slots` calls later)."""

    def __init__(self, slots=2):
        self.slots = [None] * slots
        self.turn = 0

    def take(self, n_rows, kdim, batch):
        i = self.turn
        self.turn = (i + 1) % len(self.slots)
        s = self.slots[i]
        if s is None or s['rows'].shape[0] < n_rows or s['rows'].shape[1] != kdim or s['cnt'].shape[0] < batch:
            cap = max(4096, int(n_rows * 1.5))
            s = self.slots[i] = dict(rows=torch.empty((cap, kdim), dtype=torch.float32).pin_memory(),
                                     idx=torch.empty(cap, dtype=torch.int32).pin_memory(),
                                     sel=torch.empty(cap, dtype=torch.int64).pin_memory(),
                                     cnt=torch.empty(max(batch, 64), dtype=torch.int32).pin_memory())
        return s


_default_stager = HostStager(2)


class PostResult:
    """Device-side result of a batched post-processing call."""

    def __init__(self, rows, idx, count, cand):
        self.rows, self.idx, self.count, self.cand = rows, idx, count, cand

    def to_host(self, stager=None, copy=True):
        """Host copy: list of (rows ndarray (kept,K) or None, kept row indices ndarray int32).  Only the kept rows
        cross to the host (the device buffers are sized for the worst case): the counts come first (one small
        synchronising copy), then ONE device-side gather of the kept rows of all images and two asynchronous copies
        into page-locked staging.  copy=False returns views of the staging slot (see HostStager)."""
        stager = stager or _default_stager
        b, max_keep, k = self.rows.shape
        s = stager.take(0, k, b) if stager.slots[stager.turn] is not None else stager.take(4096, k, b)
        s['cnt'][:b].copy_(self.count, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        cnt = np.minimum(s['cnt'][:b].numpy(), max_keep)
        total = int(cnt.sum())
        if total == 0:
            return [(None, np.zeros((0,), np.int32)) for _ in range(b)]
        if s['rows'].shape[0] < total:
            stager.turn = (stager.turn - 1) % len(stager.slots)         # grow this slot in place
            s = stager.take(total, k, b)
        starts = np.arange(b, dtype=np.int64) * max_keep
        offs = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        sel = np.repeat(starts - offs, cnt) + np.arange(total, dtype=np.int64)     # flat (image, slot) index of every kept row
        s['sel'][:total].copy_(torch.from_numpy(sel))
        sel_dev = s['sel'][:total].to(self.rows.device, non_blocking=True)
        s['rows'][:total].copy_(self.rows.view(b * max_keep, k).index_select(0, sel_dev), non_blocking=True)
        s['idx'][:total].copy_(self.idx.view(-1).index_select(0, sel_dev), non_blocking=True)
        torch.cuda.current_stream().synchronize()
        rows, idx = s['rows'][:total].numpy(), s['idx'][:total].numpy()
        if copy:
            rows, idx = rows.copy(), idx.copy()
        out, o = [], 0
        for n in cnt.tolist():
            out.append((rows[o:o + n] if n else None, idx[o:o + n]))
            o += n
        return out

    def per_image(self):
        """Host copy with arrays of their own: list of (rows ndarray (kept,K) or None, kept row indices int64)."""
        return [(r, i.astype('int64')) for r, i in self.to_host(copy=True)]


_ws_cache = {}


def _workspace(nbytes, device):
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def _outputs(batch, kdim, max_keep, device):
    rows = torch.empty((batch, max_keep, kdim), dtype=torch.float32, device=device)
    idx = torch.empty((batch, max_keep), dtype=torch.int32, device=device)
    cnt = torch.empty((batch,), dtype=torch.int32, device=device)
    cand = torch.empty((batch,), dtype=torch.int32, device=device)
    return rows, idx, cnt, cand


def postprocess(pred, confidence_threshold, iou_threshold, scale_xy=None, max_keep=None, n_pad=0):
    """Batched predict.py:67-92 on a deployed (B,R,K) float32 CUDA tensor (not modified)."""
    if pred.dim() == 2:
        pred = pred.unsqueeze(0)
    if pred.dtype != torch.float32 or not pred.is_cuda:
        raise ValueError('pred must be a float32 CUDA tensor')
    pred = pred.contiguous()
    b, r, k = pred.shape
    max_keep = max_keep or r
    ws = _workspace(lib.odamd_postprocess_workspace_bytes(b, r, k, n_pad), pred.device)
    rows, idx, cnt, cand = _outputs(b, k, max_keep, pred.device)
    check(lib.odamd_postprocess(_ptr(pred), b, r, k, float(confidence_threshold), float(iou_threshold),
                                _ptr(scale_xy), n_pad, _ptr(rows), _ptr(idx), _ptr(cnt), _ptr(cand), max_keep,
                                _ptr(ws), ws.numel(), _stream()), 'postprocess')
    return PostResult(rows, idx, cnt, cand)


def postprocess_heads(heads, input_size, anchor, confidence_threshold, iou_threshold, scale_xy=None, max_keep=None,
                      n_pad=0, outputs=None):
    """Same chain straight from the raw heads; the (B,R,K) tensor is never materialised.
    `outputs`: (rows, idx, count, cand) tensors of an earlier call with the same sizes to write into (a serving loop
    that has consumed that result), instead of new ones."""
    d = decode_desc(heads, input_size, anchor, True)
    r = sum(h.shape[1] * h.shape[2] * h.shape[3] for h in heads)
    max_keep = max_keep or r
    dev = heads[0].device
    ws = _workspace(lib.odamd_postprocess_workspace_bytes(d.batch, r, d.kdim, n_pad), dev)
    if outputs is not None and tuple(outputs[0].shape) == (d.batch, max_keep, d.kdim):
        rows, idx, cnt, cand = outputs
    else:
        rows, idx, cnt, cand = _outputs(d.batch, d.kdim, max_keep, dev)
    check(lib.odamd_postprocess_heads(C.byref(d), float(confidence_threshold), float(iou_threshold), _ptr(scale_xy),
                                      n_pad, _ptr(rows), _ptr(idx), _ptr(cnt), _ptr(cand), max_keep, _ptr(ws),
                                      ws.numel(), _stream()), 'postprocess_heads')
    return PostResult(rows, idx, cnt, cand)


def nms_rows(rows, iou_threshold, max_keep=None, n_pad=0):
    """Stand-alone predict.py:82-92 on (B,N,K) or (N,K) float32 CUDA rows (x_min,y_min,w,h,conf,cls...)."""
    if rows.dim() == 2:
        rows = rows.unsqueeze(0)
    rows = rows.contiguous()
    b, n, k = rows.shape
    max_keep = max_keep or n
    ws = _workspace(lib.odamd_postprocess_workspace_bytes(b, n, k, n_pad), rows.device)
    out, idx, cnt, cand = _outputs(b, k, max_keep, rows.device)
    check(lib.odamd_nms_rows(_ptr(rows), b, n, k, float(iou_threshold), n_pad, _ptr(out), _ptr(idx), _ptr(cnt),
                             max_keep, _ptr(ws), ws.numel(), _stream()), 'nms_rows')
    return PostResult(out, idx, cnt, cnt)


def metric_nms(pred, confidence_threshold, iou_threshold, max_keep=100, n_pad=0):
    """Candidate selection and NMS of the validation metric (train_class.py:340-346, :76-83) on the
    NON-deployed (B,R,K) float32 CUDA prediction (not modified).  `iou_threshold` is the metric's
    argument; the suppression threshold handed to the NMS is 1 - iou_threshold as in the reference."""
    if pred.dtype != torch.float32 or not pred.is_cuda or pred.dim() != 3:
        raise ValueError('pred must be a (B,R,K) float32 CUDA tensor')
    pred = pred.contiguous()
    b, r, k = pred.shape
    ws = _workspace(lib.odamd_postprocess_workspace_bytes(b, r, k, n_pad), pred.device)
    rows, idx, cnt, cand = _outputs(b, k, max_keep, pred.device)
    check(lib.odamd_metric_nms(_ptr(pred), b, r, k, float(confidence_threshold), float(1 - iou_threshold), n_pad,
                               _ptr(rows), _ptr(idx), _ptr(cnt), _ptr(cand), max_keep, _ptr(ws), ws.numel(), _stream()),
          'metric_nms')
    return PostResult(rows, idx, cnt, cand)


def metric_nms_rows(rows, iou_threshold, max_keep=None, n_pad=0):
    """Stand-alone train_class.nms (train_class.py:77-83) on (B,N,K) or (N,K) float32 CUDA rows
    (x_min,y_min,w,h,conf,cls..); suppression threshold 1 - iou_threshold as in the reference."""
    if rows.dim() == 2:
        rows = rows.unsqueeze(0)
    rows = rows.contiguous()
    b, n, k = rows.shape
    max_keep = max_keep or n
    ws = _workspace(lib.odamd_postprocess_workspace_bytes(b, n, k, n_pad), rows.device)
    out, idx, cnt, _ = _outputs(b, k, max_keep, rows.device)
    check(lib.odamd_metric_nms_rows(_ptr(rows), b, n, k, float(1 - iou_threshold), n_pad, _ptr(out), _ptr(idx), _ptr(cnt),
                                    max_keep, _ptr(ws), ws.numel(), _stream()), 'metric_nms_rows')
    return PostResult(out, idx, cnt, cnt)


def metric_match(result, labels, n_label, iou_threshold):
    """(tp, fn, fp) per image, (B,3) int32 on the device (train_class.py:347-365).  `result`: PostResult of
    metric_nms; labels: (B,lmax,K) float32 CUDA rows (cx,cy,w,h,conf,cls..); n_label: (B,) int32."""
    b, max_keep, k = result.rows.shape
    out = torch.empty((b, 3), dtype=torch.int32, device=result.rows.device)
    check(lib.odamd_metric_match(_ptr(result.rows), _ptr(result.count), max_keep, _ptr(labels), _ptr(n_label),
                                 labels.shape[1], b, k, float(iou_threshold), _ptr(out), _stream()), 'metric_match')
    return out
