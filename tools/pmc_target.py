"""Short target for rocprofv3 counter passes: a few batch-64 bf16 640x640 network forwards plus
post-processing (the bench workload without timing), so one --pmc pass takes seconds."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import engine, postprocess          # noqa: E402
from objectdetection_amd.model.yolov7 import yolov7         # noqa: E402
from oracle import patterns                                  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
margs = patterns.default_args(input_size=640, output_class=80)
torch.manual_seed(999)
model = yolov7(margs).cuda().eval()
engine.set_compute_dtype('bf16')
x = torch.rand(64, 3, 640, 640, generator=torch.Generator().manual_seed(999)).cuda()
for _ in range(n):
    heads = model.head_tensors(x)
    postprocess.postprocess_heads(heads, 640, margs.anchor, 0.5, 0.5)
torch.cuda.synchronize()
print('done', n)
