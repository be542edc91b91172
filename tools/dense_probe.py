"""Post-processing alone on the dense-candidate configuration (1280 x 1280, batch 16, 80 classes, confidence 0.001):
target for rocprofv3 --kernel-trace --stats.  usage: dense_probe.py [REPEATS]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import engine, postprocess          # noqa: E402
from objectdetection_amd.model.yolov7 import yolov7         # noqa: E402
from oracle import patterns                                  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
margs = patterns.default_args(input_size=1280, output_class=80)
torch.manual_seed(999)
model = yolov7(margs).cuda().eval()
engine.set_compute_dtype('bf16')
x = torch.rand(16, 3, 1280, 1280, generator=torch.Generator().manual_seed(999)).cuda()
heads = model.head_tensors(x)
postprocess.postprocess_heads(heads, 1280, margs.anchor, 0.001, 0.5)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(n):
    res = postprocess.postprocess_heads(heads, 1280, margs.anchor, 0.001, 0.5)
ev[1].record()
torch.cuda.synchronize()
print('post ms per batch', ev[0].elapsed_time(ev[1]) / n, 'candidates', float(res.cand.float().mean()), 'kept',
      float(res.count.float().mean()))
