"""Per-layer device time of one YOLOv7 forward (HIP events); prints a table."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import engine
from objectdetection_amd.model.yolov7 import yolov7
from oracle import patterns

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dtype = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
args = patterns.default_args(input_size=640, output_class=80)
torch.manual_seed(999)
m = yolov7(args).cuda().eval()
engine.set_compute_dtype(dtype)
x = torch.rand(batch, 3, 640, 640).cuda()
for _ in range(3):
    m.head_tensors(x)
plan = m._plan(x)
prof = plan.profile(x, repeats=5)
tot = sum(p['ms'] for p in prof)
print(f'total {tot:.3f} ms  ({batch / tot * 1e3:.0f} img/s network only)')
for i, p in enumerate(prof):
    gf = p['flops'] / (p['ms'] * 1e-3) / 1e12 if p['ms'] > 0 else 0
    gb = p['bytes'] / (p['ms'] * 1e-3) / 1e9 if p['ms'] > 0 else 0
    print(f"{i:3d} {p['kernel']:18s} {p['shape']:22s} {p['ms']:8.4f} ms {gf:8.1f} TF/s {gb:8.0f} GB/s")
