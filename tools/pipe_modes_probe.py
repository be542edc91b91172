"""Pipelined detection step under the combinations of side-branch stream and half overlap, in one process
(the stream set grows as the modes need it: halves only first, then side only, then both).
usage: pipe_modes_probe.py [batch] [rounds]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import engine, predict
from objectdetection_amd.model.yolov7 import yolov7
from oracle import patterns

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
args = patterns.default_args(input_size=640, output_class=80)
engine.set_compute_dtype('bf16')
x = torch.rand(batch, 3, 640, 640).cuda()
print('GPU_MAX_HW_QUEUES', os.environ.get('GPU_MAX_HW_QUEUES'), flush=True)
for rnd in range(rounds):
    for side, halves, div in ((False, False, 32), (False, True, 8), (False, True, 16), (False, True, 32), (True, False, 32)):
        engine.SIDE_STREAMS = side
        engine.SPLIT_MAP_DIV = div
        torch.manual_seed(999)
        m = yolov7(args).cuda().eval()
        pipe = predict.detect_pipeline(m, 0.5, 0.5, anchor=args.anchor, input_size=640, overlap_halves=halves)
        for _ in range(6):
            pipe.submit(x)
        pipe.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(60):
            pipe.submit(x)
        pipe.synchronize()
        e1.record()
        torch.cuda.synchronize()
        print(f'side {side!s:5} halves {halves!s:5} split at maps of 1/{div}: pipelined step {e0.elapsed_time(e1) / 60:.3f} ms', flush=True)
        del pipe, m
