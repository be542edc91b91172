import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import patterns
from objectdetection_amd.model.yolov7 import yolov7
from objectdetection_amd.train_class import train_class, synthetic_dataset

cfg = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'pruned.npz'))['cfg'].tolist()
for amp in (False, True):
    a = patterns.default_args(input_size=128, output_class=3)
    a.__dict__.update(device='cuda', batch=4, epoch=1, lr_start=0.001, lr_end_ratio=0.1, lr_end_epoch=10, warmup_ratio=0.01,
                      regularization='L2', r_value=0.0005, ema=True, distributed=False, local_rank=0, weight_path='',
                      weight_again=True, print_info=False, synthetic_images=8, save_path=None, amp=amp)
    torch.manual_seed(9)
    tc = train_class(a)
    tc.model = yolov7(a, config=cfg).cuda()
    tc.model.train_dtype = 'bf16' if amp else 'f32'
    tc.model_dict['model'] = tc.model
    tc.optimizer, tc.optimizer_adjust = tc.optimizer_load()
    ds = synthetic_dataset(a, 8)
    image, screen, expend, _ = ds.collate_fn([ds[i] for i in range(4)])
    tc.model.train()
    pred = tc.model(image.cuda())
    torch.cuda.synchronize()
    print('amp', amp, 'pred finite', bool(torch.isfinite(pred).all()), float(pred.abs().max()))
    plan = pred._odamd_plan
    for op in plan.ops:
        if op['kind'] in ('conv', 'first'):
            raw = op['raw'].float()
            if not torch.isfinite(raw).all():
                names = {m: n for n, m in tc.model.named_modules()}
                print(' first non-finite raw at', names[op['mod']], op['mod'].conv.in_channels, op['mod'].conv.out_channels,
                      'x segments', op['x'].segments if 'x' in op else None)
                xin = plan.bufs[op['x'].buf].float() if 'x' in op else None
                if xin is not None:
                    print('  input finite', bool(torch.isfinite(xin).all()), 'wf finite', bool(torch.isfinite(op['wf'].float()).all()))
                break
    losses = tc.loss(pred, screen, expend)
    print(' losses', [float(v) for v in losses])
