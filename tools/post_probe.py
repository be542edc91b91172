"""Post-processing alone (decode + filter + sort + NMS + emit from fixed raw heads of the batch-64 bench
workload): target for rocprofv3 --kernel-trace --stats / --pmc.  usage: post_probe.py [REPEATS]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import engine, postprocess          # noqa: E402
from objectdetection_amd.model.yolov7 import yolov7         # noqa: E402
from oracle import patterns                                  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
margs = patterns.default_args(input_size=640, output_class=80)
torch.manual_seed(999)
model = yolov7(margs).cuda().eval()
engine.set_compute_dtype('bf16')
x = torch.rand(64, 3, 640, 640, generator=torch.Generator().manual_seed(999)).cuda()
heads = model.head_tensors(x)
postprocess.postprocess_heads(heads, 640, margs.anchor, 0.5, 0.5)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(n):
    res = postprocess.postprocess_heads(heads, 640, margs.anchor, 0.5, 0.5)
ev[1].record()
torch.cuda.synchronize()
print('post ms per batch', ev[0].elapsed_time(ev[1]) / n, 'kept', float(res.count.float().mean()))
