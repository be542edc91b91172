"""HIP-event time of every launch-list entry of one training step (bench training workload, batch 32 @640, C=5).
usage: train_layer_profile.py [amp: 0|1] [REPS]
Prints one line per entry of TrainPlan.fwd / TrainPlan.bwd (forward order, then backward order) and totals per kind."""
import os
import sys
import types
import collections
import ctypes as C
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402
from objectdetection_amd import train_engine as te             # noqa: E402
from objectdetection_amd._lib import lib, check                # noqa: E402

amp = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
a = types.SimpleNamespace(size=640, train_classes=5, train_batch=32)
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
bench.note = lambda *args, **kw: None
res, tc = bench.train_leg(a, dev, 0, 1, None, 2, 2, amp=amp)
batch = tc.train_dataloader[0]
image = batch[0].to(dev)
tc.train_step(image, batch[1], batch[2])
torch.cuda.synchronize()
plan = tc.model.train_plan(image.shape[0], image.shape[2])
stream = te._stream()


def describe(fn, args, name):
    if fn in ('bn_fwd', 'bn_bwd'):
        op, npix, cout, cphys = args
        return '%s npix=%d c=%d' % (name, npix, cout)
    if name in ('conv', 'dgrad', 'head'):
        d = args[0]._obj
        return '%s k%d s%d %dx%d %d->%d%s' % (name, d.ksize, d.stride, d.h, d.w_in, d.cin, d.cout, ' upz' if d.in_upzero2 else '')
    if name == 'wgrad':
        return 'wgrad k%d s%d %dx%d %d->%d' % (args[10], args[11], args[6], args[7], args[8], args[9])
    return name


def run(fn, args, name, slot_or_extra, backward):
    if fn == 'bn_fwd':
        plan._bn_fwd(*args, stream)
    elif fn == 'bn_bwd':
        plan._bn_bwd(*args, stream)
    elif fn == 'mark':
        return False
    elif backward and slot_or_extra is not None:
        return False                      # scratch-tensor variants (irregular widths): not in the bench network
    else:
        if not backward and slot_or_extra is not None:
            args[slot_or_extra] = image.data_ptr()
        check(fn(*args, stream), name)
    return True


rows = []
for backward, entries in ((False, plan.fwd), (True, plan.bwd)):
    for fn, args, name, extra in entries:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        if not run(fn, args, name, extra, backward):
            continue
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(reps):
            run(fn, args, name, extra, backward)
        ev[1].record()
        torch.cuda.synchronize()
        rows.append((('bwd ' if backward else 'fwd ') + describe(fn, args, name), ev[0].elapsed_time(ev[1]) / reps * 1e3))
tot = collections.defaultdict(float)
for label, us in rows:
    print('%-60s %9.1f us' % (label, us))
    tot[' '.join(label.split()[:2])] += us
print('---- totals (ms)')
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print('%-30s %8.3f' % (k, v / 1e3))
print('sum %.3f ms' % (sum(tot.values()) / 1e3))
