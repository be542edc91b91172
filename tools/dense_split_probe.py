"""Dense-candidate post-processing (1280 x 1280, batch 16, confidence 0.001) with the batch cut into slices that run on
streams of their own: the greedy scan of one slice (one workgroup per image) beside the IoU mask of another."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import engine, postprocess          # noqa: E402
from objectdetection_amd.model.yolov7 import yolov7         # noqa: E402
from oracle import patterns                                  # noqa: E402

margs = patterns.default_args(input_size=1280, output_class=80)
torch.manual_seed(999)
model = yolov7(margs).cuda().eval()
engine.set_compute_dtype('bf16')
x = torch.rand(16, 3, 1280, 1280, generator=torch.Generator().manual_seed(999)).cuda()
heads = model.head_tensors(x)
torch.cuda.synchronize()
ref = None
for slices in (1, 2, 4, 8, 16, 1):
    streams = [torch.cuda.Stream() for _ in range(slices)]
    per = 16 // slices

    def run():
        out = []
        cur = torch.cuda.current_stream()
        for s, st in enumerate(streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                out.append(postprocess.postprocess_heads([h[s * per:(s + 1) * per] for h in heads], 1280, margs.anchor, 0.001, 0.5))
        for st in streams:
            cur.wait_stream(st)
        return out
    run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(3):
        res = run()
    ev[1].record()
    torch.cuda.synchronize()
    cnt = torch.cat([r.count for r in res]).cpu()
    if ref is None:
        ref = cnt
    print(f'slices {slices}: post {ev[0].elapsed_time(ev[1]) / 3:.2f} ms per batch, kept identical {bool(torch.equal(cnt, ref))}', flush=True)
