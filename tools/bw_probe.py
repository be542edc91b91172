"""Practical HBM ceilings on this box: device copy, read-only sum, write-only fill (torch kernels)."""
import torch
x = torch.empty(1 << 29, dtype=torch.float16, device='cuda').normal_()     # 1 GiB
y = torch.empty_like(x)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
gb = x.numel() * 2 / 1e9
print('copy  r+w GB/s', 2 * gb / t(lambda: y.copy_(x)))
print('sum   r   GB/s', gb / t(lambda: x.sum()))
print('fill  w   GB/s', gb / t(lambda: y.fill_(1.0)))
print('silu  r+w GB/s', 2 * gb / t(lambda: torch.nn.functional.silu(x, inplace=False)))
for mb in (64, 256):
    a = x[: mb << 19]; b = y[: mb << 19]
    print(f'copy {mb} MiB r+w GB/s', 2 * a.numel() * 2 / 1e9 / t(lambda: b.copy_(a), 50))
