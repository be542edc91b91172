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


class PostResult:
    """Device-side result of a batched post-processing call."""

    def __init__(self, rows, idx, count, cand):
        self.rows, self.idx, self.count, self.cand = rows, idx, count, cand

    def per_image(self):
        """Host copy: list of (rows ndarray (kept,K) or None, kept row indices ndarray).  Only the kept rows
        cross to the host (the device buffers are sized for the worst case: every row kept)."""
        cnt = [min(n, self.idx.shape[1]) for n in self.count.cpu().tolist()]
        if sum(cnt) == 0:
            return [(None, np.zeros((0,), np.int64)) for _ in cnt]
        # two device-side gathers and two copies for the whole batch
        rows = torch.cat([self.rows[b, :n] for b, n in enumerate(cnt)]).cpu().numpy()
        idx = torch.cat([self.idx[b, :n] for b, n in enumerate(cnt)]).cpu().numpy().astype('int64')
        out, o = [], 0
        for n in cnt:
            out.append((rows[o:o + n] if n else None, idx[o:o + n]))
            o += n
        return out


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
                      n_pad=0):
    """Same chain straight from the raw heads; the (B,R,K) tensor is never materialised."""
    d = decode_desc(heads, input_size, anchor, True)
    r = sum(h.shape[1] * h.shape[2] * h.shape[3] for h in heads)
    max_keep = max_keep or r
    dev = heads[0].device
    ws = _workspace(lib.odamd_postprocess_workspace_bytes(d.batch, r, d.kdim, n_pad), dev)
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
