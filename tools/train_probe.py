"""Host-side enqueue cost of one training step against its device time (bench training workload).
usage: train_probe.py [amp: 0|1] [STEPS]"""
import os
import sys
import time
import types
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402

amp = bool(int(sys.argv[1])) if len(sys.argv) > 1 else False
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
a = types.SimpleNamespace(size=640, train_classes=5, train_batch=32)
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
bench.note = lambda *args, **kw: None
res, tc = bench.train_leg(a, dev, 0, 1, None, 3, 3, amp=amp)
from oracle import patterns                                    # noqa: E402
batch = tc.train_dataloader[0]
image = batch[0].to(dev)
screen, expend = batch[1], batch[2]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    tc.train_step(image, screen, expend)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
single = []
for _ in range(5):                      # one step into an empty queue: the host-side cost alone
    torch.cuda.synchronize()
    a0 = time.perf_counter()
    tc.train_step(image, screen, expend)
    single.append(1e3 * (time.perf_counter() - a0))
torch.cuda.synchronize()
print('amp=%s: enqueue of one step into an empty queue: %s ms' % (amp, ' '.join('%.2f' % v for v in single)))
print(f'amp={amp}: host enqueue {1e3 * (t1 - t0) / n:.2f} ms per step; with device drain {1e3 * (t2 - t0) / n:.2f} ms per step')
