"""Inference entry point, API-compatible with the reference's predict.py (predict_class with
predict / _decode / _nms / iou), running the network and the whole post-processing on the GPU.

Reference: predict.py:22-92.  Differences that no caller on the path observes are deliberate:
the numpy input of `_decode` is not modified in place (predict.py:68 overwrites the caller's
array), and a whole batch can be post-processed at once (`predict_batch`); the reference parses a
`--batch` flag (predict.py:11) but only ever decodes image 0 (predict.py:56).
"""
import argparse
import ctypes as C

import numpy as np
import torch

from . import engine, postprocess as _post
from ._lib import lib, check
from .model.layer import deploy

parser = argparse.ArgumentParser(description='model prediction')
parser.add_argument('--input_size', default=640, type=int)
parser.add_argument('--batch', default=1, type=int)
parser.add_argument('--device', default='cuda', type=str)
parser.add_argument('--float16', default=True, type=bool)
parser.add_argument('--confidence_threshold', default=0.5, type=float)
parser.add_argument('--iou_threshold', default=0.5, type=float)
args, _ = parser.parse_known_args()


def load_checkpoint(model_path, allow_pickle=True):
    """Checkpoint file -> its dict with the yolov7 module of this build under 'model'.  The file is read once.
    Two forms are accepted:
      * {'state_dict': ..., 'meta': ..., ...} (what train_class writes next to every checkpoint as *.weights.pt):
        tensors and plain Python data only, read with weights_only=True, nothing in the file is executed;
      * the reference's own format (train_class.py:262-276, read like predict.py:29): a pickled dict whose
        'model' entry is the module object.  Its class paths model.yolov7.yolov7 / model.layer.* resolve to
        this build's classes through the top-level `model` package; unpickling runs code from the file, so
        only load checkpoints you trust.  This form is only tried when the safe loader refused the file
        BECAUSE it holds pickled classes (an unreadable or truncated file is an error, not a reason to unpickle),
        a warning says so, and `allow_pickle=False` (or ODAMD_NO_PICKLE=1) turns it off.
    Anything that does not come out as this build's yolov7 is refused: nothing here runs a network through
    torch's own operators."""
    import os
    import pickle
    import warnings
    from .model.yolov7 import yolov7
    try:
        ckpt = torch.load(model_path, map_location='cpu', weights_only=True)
    except pickle.UnpicklingError as e:
        text = str(e)
        if not ('Unsupported global' in text or 'weights_only' in text or 'GLOBAL' in text):
            raise
        if not allow_pickle or os.environ.get('ODAMD_NO_PICKLE'):
            raise RuntimeError(f'{model_path}: holds pickled objects and loading them is disabled; use the '
                               '*.weights.pt file written next to it') from e
        warnings.warn(f'{model_path}: reference-format checkpoint (a pickled module): unpickling executes code '
                      'from the file', stacklevel=2)
        ckpt = torch.load(model_path, map_location='cpu', weights_only=False)
    if isinstance(ckpt, dict) and 'state_dict' in ckpt and 'meta' in ckpt:
        out = {k: v for k, v in ckpt.items() if k not in ('state_dict', 'meta')}
        out['model'] = yolov7.from_checkpoint(ckpt)
        return out
    out = dict(ckpt) if isinstance(ckpt, dict) else {'model': ckpt}
    model = getattr(out.get('model'), 'module', out.get('model'))
    if not isinstance(model, yolov7):
        raise TypeError(f'{model_path}: holds a {type(model).__module__}.{type(model).__name__}, not the yolov7 of '
                        'objectdetection_amd (put this repository ahead of the reference tree on sys.path so that '
                        'model.yolov7 resolves to it); there is no fallback through torch operators')
    out['model'] = model
    return out


def load_model(model_path, allow_pickle=True):
    """The yolov7 module of a checkpoint (see load_checkpoint)."""
    return load_checkpoint(model_path, allow_pickle)['model']


class predict_class:
    def __init__(self, model_path, args=args, model=None):
        """`model_path`: checkpoint written by training (a dict holding the module under 'model').
        `model`: alternatively an already built yolov7 module (no file needed)."""
        if not torch.cuda.is_available():
            raise RuntimeError('objectdetection_amd.predict needs a GPU: there is no CPU fallback')
        self.device = args.device
        self.float16 = args.float16
        self.input_size = args.input_size
        self.confidence_threshold = args.confidence_threshold
        self.iou_threshold = args.iou_threshold
        if model is None:
            model = load_model(model_path)
        if not hasattr(model, 'head_tensors'):
            raise TypeError('predict_class needs the yolov7 module of objectdetection_amd (no fallback through torch operators)')
        self.model = deploy(model).eval().to(args.device)
        # the reference runs fp16 on a GPU (predict.py:18,31); the MI355X build computes in bf16
        # on the matrix cores instead (float32 parameters stay the master copy)
        self.dtype = 'bf16' if args.float16 else 'f32'

    # ---- predict.py:33-41 -------------------------------------------------------------------
    @staticmethod
    def iou(pred, label):
        """IoU of (x_min,y_min,w,h) boxes, row-wise with broadcasting of a single `pred` row."""
        a = torch.as_tensor(np.ascontiguousarray(pred[:, 0:4], dtype=np.float32)).cuda()
        b = torch.as_tensor(np.ascontiguousarray(label[:, 0:4], dtype=np.float32)).cuda()
        n = max(len(a), len(b))
        out = torch.empty(n, dtype=torch.float32, device='cuda')
        check(lib.odamd_iou_pairs(a.data_ptr(), len(a), b.data_ptr(), len(b), out.data_ptr(), n,
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'iou_pairs')
        return out.cpu().numpy()

    # ---- predict.py:53-65 -------------------------------------------------------------------
    def _image_process(self, image):
        """(H, W, 3) uint8 image -> (1, 3, S, S) float32 CUDA tensor: 8-bit bilinear resize, / 255 and the
        HWC -> CHW transpose in one kernel on the device (the image crosses PCIe as bytes)."""
        image = np.ascontiguousarray(image)
        if image.dtype != np.uint8 or image.ndim != 3 or image.shape[2] != 3:
            raise ValueError('image must be an (H, W, 3) uint8 array')
        src = torch.from_numpy(image).to(self.device)
        out = torch.empty((1, 3, self.input_size, self.input_size), dtype=torch.float32, device=self.device)
        check(lib.odamd_preprocess_u8(src.data_ptr(), image.shape[0], image.shape[1], out.data_ptr(), self.input_size,
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'preprocess_u8')
        return out

    def _forward_heads(self, tensor):
        engine.set_compute_dtype(self.dtype)
        try:
            return self.model.model.head_tensors(tensor)
        finally:
            engine.set_compute_dtype(None)

    def predict(self, image):
        tensor = self._image_process(image)
        scale = torch.tensor([[image.shape[1] / self.input_size, image.shape[0] / self.input_size]],
                             dtype=torch.float32, device=self.device)
        with torch.no_grad():
            heads = self._forward_heads(tensor)
            dec = self.model.model.decode
            res = _post.postprocess_heads([h[:1] for h in heads], dec.input_size, dec.anchor,
                                          self.confidence_threshold, self.iou_threshold, scale)
        return res.per_image()[0][0]

    def predict_batch(self, tensor, shapes=None):
        """(B,3,S,S) float tensor already on the device -> list of (kept,K) arrays / None."""
        b = tensor.shape[0]
        if shapes is None:
            scale = None
        else:
            scale = torch.tensor([[s[1] / self.input_size, s[0] / self.input_size] for s in shapes],
                                 dtype=torch.float32, device=tensor.device)
        with torch.no_grad():
            heads = self._forward_heads(tensor)
            dec = self.model.model.decode
            res = _post.postprocess_heads(heads, dec.input_size, dec.anchor, self.confidence_threshold,
                                          self.iou_threshold, scale)
        return [r for r, _ in res.per_image()][:b]

    # ---- predict.py:67-92 -------------------------------------------------------------------
    def _decode(self, pred, shape):
        p = torch.as_tensor(np.ascontiguousarray(pred, dtype=np.float32)).cuda()
        scale = torch.tensor([[shape[1] / self.input_size, shape[0] / self.input_size]], dtype=torch.float32,
                             device='cuda')
        return _post.postprocess(p, self.confidence_threshold, self.iou_threshold, scale).per_image()[0][0]

    def _nms(self, pred):
        p = torch.as_tensor(np.ascontiguousarray(pred, dtype=np.float32)).cuda()
        return _post.nms_rows(p, self.iou_threshold).per_image()[0][0]


class PostResultList:
    """Results of the sub-batches of one `detect_pipeline.submit` call, in image order."""

    def __init__(self, parts):
        self.parts = parts
        self.done = [p.done for p in parts]

    @property
    def count(self):
        return torch.cat([p.count for p in self.parts])

    @property
    def cand(self):
        return torch.cat([p.cand for p in self.parts])

    def per_image(self):
        out = []
        for p in self.parts:
            out.extend(p.per_image())
        return out


OVERLAP_HALVES = True      # detect_pipeline: small-map half of batch i beside the large-map half of batch i + 1


class detect_pipeline:
    """Batched detection as a pipeline on several HIP streams: the network of batch i+1 runs while
    batch i is decoded, filtered, sorted and NMS-ed (predict.py:67-92 per image).  The network is
    HBM-bound and the NMS mask VALU-bound, so the stages overlap well.  Two network plans alternate, so
    the heads a post-processing pass reads are not overwritten by the next forward.

    `splits` > 1 additionally cuts every batch into that many sub-batches (images are independent) that
    run concurrently on their own streams: the launches of the small 40x40 / 20x20 maps are latency
    bound and fill only part of the chip each, several of them in flight at different layers fill it.

        pipe = detect_pipeline(model, 0.5, 0.5)
        results = [pipe.submit(batch) for batch in batches]     # asynchronous
        pipe.synchronize()                                      # results' tensors are now final
    """

    def __init__(self, model, confidence_threshold, iou_threshold, anchor=None, input_size=None, splits=1, max_keep=None,
                 reuse_outputs=0, overlap_halves=None):
        """`max_keep`: bound of kept rows per image (sizes the output buffers; default: every row).
        `reuse_outputs` = n > 0: the result tensors of a submit are written over by the n-th submit after it (a
        serving loop that has read a result by then), instead of allocating (batch, max_keep, K) per submit."""
        self.model = model
        self.max_keep, self.reuse_outputs = max_keep, reuse_outputs
        self._outs, self._out_turn = {}, 0
        self.conf, self.iou = confidence_threshold, iou_threshold
        self.anchor = anchor if anchor is not None else model.decode.anchor
        self.input_size = input_size if input_size is not None else model.decode.input_size
        self.splits = splits
        dev = next(model.parameters()).device        # the streams belong to the model's device, not to the current one
        self.net_streams = [engine.shared_stream(dev, f'net{i}') for i in range(splits)]
        self.post_streams = [engine.shared_stream(dev, f'post{i}') for i in range(splits)]
        # the launches on the maps of stride 16 and coarser (about half of the network's time) fill only part of the chip:
        # they run on a stream of their own, beside the large-map half of the NEXT batch (the two batches use different plans)
        if overlap_halves is None:
            overlap_halves = OVERLAP_HALVES
        self.back_streams = [engine.shared_stream(dev, f'back{i}') for i in range(splits)] if overlap_halves else None
        self.slot = 0
        self.heads_free = {}                # (slot, split) -> event: post-processing has finished reading those heads

    def submit(self, x, scale_xy=None):
        """Enqueue one batch; returns its result (device tensors, final once `result.done` has completed or
        after `synchronize()`): a PostResult, or a PostResultList when splits > 1."""
        slot = self.slot
        self.slot ^= 1
        b = x.shape[0]
        n = self.splits if b % self.splits == 0 else 1
        hb = b // n
        cur = torch.cuda.current_stream()
        parts = []
        for s in range(n):
            xs = x if n == 1 else x[s * hb:(s + 1) * hb]
            sc = None if scale_xy is None else scale_xy[s * hb:(s + 1) * hb]
            net, post = self.net_streams[s], self.post_streams[s]
            net.wait_stream(cur)
            xs.record_stream(net)           # the caller may free x while the network still reads it
            with torch.cuda.stream(net):
                free = self.heads_free.get((slot, s))
                if free is not None:
                    net.wait_event(free)
                back = self.back_streams[s] if self.back_streams else None
                heads = self.model.head_tensors(xs, slot=(slot, s), back=back)
                ready = torch.cuda.Event()
                ready.record(back if back is not None else net)
            with torch.cuda.stream(post):
                post.wait_event(ready)
                outs = self._outs.get((self._out_turn, s)) if self.reuse_outputs else None
                res = _post.postprocess_heads(heads, self.input_size, self.anchor, self.conf, self.iou, sc,
                                              max_keep=self.max_keep, outputs=outs)
                if self.reuse_outputs:
                    self._outs[(self._out_turn, s)] = (res.rows, res.idx, res.count, res.cand)
                done = torch.cuda.Event()
                done.record(post)
            self.heads_free[(slot, s)] = done
            res.done = done
            parts.append(res)
        if self.reuse_outputs:
            self._out_turn = (self._out_turn + 1) % self.reuse_outputs
        return parts[0] if n == 1 else PostResultList(parts)

    def synchronize(self):
        for st in self.net_streams + self.post_streams + (self.back_streams or []):
            st.synchronize()
