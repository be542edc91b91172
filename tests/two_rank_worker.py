"""Worker of tests/test_gpu_two_ranks.py: one data-parallel training step on two ranks (gloo, both ranks on
cuda:0 - the form a one-GPU box allows).  Every rank checks that (1) the parameters after the step are bit
identical on both ranks and (2) they equal Adam applied to the MEAN of the two ranks' local gradients
(DDP semantics, train_class.py:37-40 / :158-167).  Exit code 0 = both hold."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import patterns, training as otrain  # noqa: E402


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    dist.init_process_group('gloo')
    from objectdetection_amd.train_class import train_class, synthetic_dataset
    a = patterns.default_args(input_size=128, output_class=5)
    a.__dict__.update(device='cuda', batch=2, epoch=1, lr_start=0.001, lr_end_ratio=0.1, lr_end_epoch=10, warmup_ratio=0.01,
                      regularization='L2', r_value=0.0005, ema=True, distributed=True, local_rank=rank, weight_path='',
                      weight_again=True, print_info=False, synthetic_images=8, save_path=None, amp=False)
    torch.manual_seed(100 + rank)                 # different initial weights per rank: the broadcast must equalise them
    ds = synthetic_dataset(a, 8, seed=5)
    batch = ds.collate_fn([ds[rank * 2 + i] for i in range(2)])
    tc = train_class(a, dataloader=[batch] * 4)
    st = tc.model.train_state()
    p0 = st.param.clone()
    gathered = [torch.empty_like(p0.cpu()) for _ in range(world)]
    dist.all_gather(gathered, p0.cpu())
    assert all(torch.equal(g, gathered[0]) for g in gathered), 'rank 0 broadcast did not equalise the parameters'
    image, screen, expend = batch[0].cuda(), batch[1], batch[2]
    # local gradient of this rank (no reducer), then put the BatchNorm buffers back
    buffers, counters = st.buffers.clone(), st.counters.clone()
    tc.model.train()
    plan = tc.model.train_plan(image.shape[0], image.shape[2])
    plan.reducer = None
    pred = tc.model(image)
    tc.optimizer.zero_grad()
    tc.loss(pred, screen, expend)[0].backward()
    torch.cuda.synchronize()
    g_local = st.grad.clone().cpu()
    st.buffers.copy_(buffers)
    st.counters.copy_(counters)
    # the real data-parallel step
    lr = tc.optimizer.param_groups[0]['lr']
    tc.train_step(image, screen, expend)
    torch.cuda.synchronize()
    p1 = st.param.clone().cpu()
    grads = [torch.empty_like(g_local) for _ in range(world)]
    dist.all_gather(grads, g_local)
    params = [torch.empty_like(p1) for _ in range(world)]
    dist.all_gather(params, p1)
    assert float((grads[0] - grads[1]).abs().max()) > 0, 'the two ranks saw the same data'
    assert all(torch.equal(p, params[0]) for p in params), 'parameters differ between the ranks after the step'
    # Adam on the mean gradient: the kernel adds the ranks' gradients and scales by 1 / world
    g_mean = (grads[0] + grads[1]) * 0.5
    p, m, v = p0.cpu().clone(), torch.zeros_like(g_mean), torch.zeros_like(g_mean)
    otrain.adam_step(p, g_mean, m, v, 1, lr, weight_decay=a.r_value)
    # Adam's first step moves every element by lr * g / (|g| + eps): where the summed gradient is within rounding
    # of zero the direction is decided by its last bits, so those few elements may differ by up to 2 lr
    settled = (g_mean + a.r_value * p0.cpu()).abs() > 1e-6
    np.testing.assert_allclose(p1[settled].numpy(), p[settled].numpy(), rtol=1e-5, atol=0.01 * lr)   # 1 % of a step
    assert float((p1 - p).abs().max()) <= 2.5 * lr
    assert float(settled.float().mean()) > 0.98
    moved = float((p1 - p0.cpu()).abs().max())
    assert moved > 0
    dist.barrier()
    dist.destroy_process_group()
    print(f'rank {rank}: ok, parameters moved by up to {moved:.2e}', flush=True)


if __name__ == '__main__':
    main()
