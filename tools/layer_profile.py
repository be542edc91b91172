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
preset = sys.argv[3] if len(sys.argv) > 3 else 's'
args = patterns.default_args(input_size=640, output_class=80, model_type=preset)
torch.manual_seed(999)
m = yolov7(args).cuda().eval()
engine.set_compute_dtype(dtype)
x = torch.rand(batch, 3, 640, 640).cuda()
for _ in range(3):
    m.head_tensors(x)
plan = m._plan(x)
prof = plan.profile(x, repeats=5)
tot = sum(p['ms'] for p in prof)
ideal = sum(max(p['flops'] / 2.5e15, p['bytes'] / 8e12) for p in prof) * 1e3
flops = sum(p['flops'] for p in prof)
print(f'total {tot:.3f} ms  ({batch / tot * 1e3:.0f} img/s network only)  preset {preset}  per-layer bound {ideal:.3f} ms '
      f'(fraction {ideal / tot:.3f})  {flops / tot * 1e-9:.0f} TFLOP/s')
for i, p in enumerate(prof):
    gf = p['flops'] / (p['ms'] * 1e-3) / 1e12 if p['ms'] > 0 else 0
    gb = p['bytes'] / (p['ms'] * 1e-3) / 1e9 if p['ms'] > 0 else 0
    bound = max(p['flops'] / 2.5e15, p['bytes'] / 8e12) * 1e3
    print(f"{i:3d} {p['kernel']:18s} {p['shape']:22s} {p['ms']:8.4f} ms {gf:8.1f} TF/s {gb:8.0f} GB/s  frac {bound / p['ms'] if p['ms'] > 0 else 0:5.2f}")
