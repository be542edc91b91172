"""Find the first layer whose bf16 train-mode output is not finite for the pruned configuration."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import patterns
from objectdetection_amd.model.yolov7 import yolov7
from objectdetection_amd import train_engine as te
from objectdetection_amd.engine import BF16, F32

cfg = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'pruned.npz'))['cfg'].tolist()
args = patterns.default_args(input_size=128, output_class=3)
torch.manual_seed(9)
net = yolov7(args, config=cfg).cuda().train()
names = {m: n for n, m in net.named_modules()}
state = te.FlatState(net)
x = torch.rand(4, 3, 128, 128, generator=torch.Generator().manual_seed(1)).cuda()
for dt, label in ((F32, 'f32'), (BF16, 'bf16')):
    plan = te.TrainPlan(net, state, 4, 128, dtype=dt)
    pred = plan.forward(x)
    torch.cuda.synchronize()
    print('==', label, 'pred finite', bool(torch.isfinite(pred).all()))
    for op in plan.ops:
        if op['kind'] not in ('conv', 'first'):
            continue
        y = op['y']
        raw = op['raw'].float()
        act = plan.bufs[y.buf][..., y.c0:y.c0 + y.cphys].float()
        bad = (not torch.isfinite(raw).all()) or (not torch.isfinite(act).all())
        pad = float(act[..., y.c:].abs().max()) if y.cphys > y.c else 0.0
        if bad or pad != 0.0:
            print(names[op['mod']], 'cin', op['mod'].conv.in_channels, 'cout', op['mod'].conv.out_channels, 'raw finite',
                  bool(torch.isfinite(raw).all()), 'act finite', bool(torch.isfinite(act).all()), 'padding max', pad,
                  'stats', op['stats'][:, :3].tolist())
            if bad:
                break
