"""Does replaying the training forward as a captured HIP graph shorten it?  (bench training workload, bf16)
usage: graph_probe.py"""
import os
import sys
import time
import types
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402

a = types.SimpleNamespace(size=640, train_classes=5, train_batch=32)
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
bench.note = lambda *args, **kw: None
res, tc = bench.train_leg(a, dev, 0, 1, None, 2, 2, amp=True)
batch = tc.train_dataloader[0]
image = batch[0].to(dev)
plan = tc.model.train_plan(image.shape[0], image.shape[2])


def timed(fn, n=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for _ in range(3):
    plan.forward(image)
print('eager forward  %.3f ms' % timed(lambda: plan.forward(image)))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    plan.forward(image)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    plan.forward(image)
print('graph replay   %.3f ms' % timed(g.replay))
