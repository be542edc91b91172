"""Whole forwards back to back (HIP events around N of them) with an engine switch off and on, twice; and the pipelined
detection step with both settings.  usage: net_probe.py [batch] [SWITCH]   (SIDE_STREAMS, FUSE_CHAIN, FUSE_MP ...)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import engine
from objectdetection_amd.model.yolov7 import yolov7
from objectdetection_amd.predict import detect_pipeline
from oracle import patterns

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
switch = sys.argv[2] if len(sys.argv) > 2 else 'SIDE_STREAMS'
if len(sys.argv) > 3:         # e.g. SIDE_STREAMS=0: further engine switches held fixed for the whole run
    for kv in sys.argv[3:]:
        k, v = kv.split('=')
        setattr(engine, k, bool(int(v)))
args = patterns.default_args(input_size=640, output_class=80)
engine.set_compute_dtype('bf16')
x = torch.rand(batch, 3, 640, 640).cuda()
for side in (False, True, False, True):
    if '.' in switch:
        import importlib
        modname, attr = switch.rsplit('.', 1)
        setattr(importlib.import_module('objectdetection_amd.' + modname), attr, side)
    else:
        setattr(engine, switch, side)
    torch.manual_seed(999)
    m = yolov7(args).cuda().eval()
    for _ in range(5):
        m.head_tensors(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        m.head_tensors(x)
    e1.record()
    torch.cuda.synchronize()
    net = e0.elapsed_time(e1) / 50
    pipe = detect_pipeline(m, 0.5, 0.5, anchor=args.anchor, input_size=640)
    for _ in range(5):
        pipe.submit(x)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        pipe.submit(x)
    pipe.synchronize()
    e1.record()
    torch.cuda.synchronize()
    print(f'{switch} {side}: network alone {net:.3f} ms per forward, pipelined step {e0.elapsed_time(e1) / 50:.3f} ms', flush=True)
    del m, pipe
