"""Run one launch of the batch-64 bf16 network plan repeatedly (target for rocprofv3 counter passes).
usage: python tools/layer_probe.py LAUNCH_INDEX [REPEATS]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import engine                      # noqa: E402
from objectdetection_amd._lib import check                  # noqa: E402
from objectdetection_amd.model.yolov7 import yolov7         # noqa: E402
from oracle import patterns                                  # noqa: E402

idx = int(sys.argv[1])
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 5
margs = patterns.default_args(input_size=640, output_class=80)
torch.manual_seed(999)
model = yolov7(margs).cuda().eval()
engine.set_compute_dtype('bf16')
x = torch.rand(64, 3, 640, 640).cuda()
model.head_tensors(x)
plan = model._plan(x)
fn, args, name, slot = plan.launches[idx]
if slot is not None:
    args[slot] = x.data_ptr()
st = engine._stream()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(rep):
    check(fn(*args, st), name)
ev[1].record()
torch.cuda.synchronize()
print(name, plan.meta.get(idx, {}), 'ms per launch', ev[0].elapsed_time(ev[1]) / rep)
