"""Time of odamd_spp_backward alone (bench training shape: batch 32, 20 x 20, 256 channels)."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd._lib import lib, check

B, H, W, Cn = 32, 20, 20, int(sys.argv[1]) if len(sys.argv) > 1 else 256
for dt, code in ((torch.bfloat16, 1), (torch.float32, 0)):
    x = torch.randn(B, H, W, Cn, device='cuda').to(dt)
    dys = [torch.randn(B, H, W, 4 * Cn, device='cuda').to(dt) for _ in range(1)]
    dy = dys[0]
    dx = torch.zeros(B, H, W, Cn, device='cuda').to(dt)
    es = x.element_size()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run():
        check(lib.odamd_spp_backward(x.data_ptr(), Cn, dy.data_ptr() + Cn * es, dy.data_ptr() + 2 * Cn * es, dy.data_ptr() + 3 * Cn * es,
                                     4 * Cn, dx.data_ptr(), Cn, B, H, W, Cn, code, st), 'spp')
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(dt, f'{e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call', flush=True)
