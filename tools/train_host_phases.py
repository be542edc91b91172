"""Host time of the phases of train_class.train_step (bench training workload, bf16): where the Python thread waits.
usage: train_host_phases.py [amp: 0|1] [STEPS]"""
import os
import sys
import time
import types
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402

amp = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
a = types.SimpleNamespace(size=640, train_classes=5, train_batch=32)
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
bench.note = lambda *args, **kw: None
res, tc = bench.train_leg(a, dev, 0, 1, None, 3, 3, amp=amp)
batch = tc.train_dataloader[0]
image = batch[0].to(dev)
screen, expend = batch[1], batch[2]
torch.cuda.synchronize()
acc = {}
t_all = time.perf_counter()
for _ in range(n):
    t = [time.perf_counter()]
    pred = tc.model(image); t.append(time.perf_counter())
    losses = tc.loss(pred, screen, expend); t.append(time.perf_counter())
    tc.optimizer.zero_grad(); t.append(time.perf_counter())
    losses[0].backward(); t.append(time.perf_counter())
    tc.optimizer.step(); t.append(time.perf_counter())
    tc.optimizer = tc.optimizer_adjust(tc.optimizer); t.append(time.perf_counter())
    for name, d in zip(('forward', 'loss', 'zero_grad', 'backward', 'optimizer', 'lr_adjust'), [b - a2 for a2, b in zip(t, t[1:])]):
        acc[name] = acc.get(name, 0.0) + d
torch.cuda.synchronize()
total = time.perf_counter() - t_all
print('amp=%s  %.2f ms per step;' % (amp, 1e3 * total / n), '  '.join('%s %.2f' % (k, 1e3 * v / n) for k, v in acc.items()), '(host ms per step)')
