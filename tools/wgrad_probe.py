"""Weight-gradient launches of the bench training shapes on their own (bf16, batch 32), for rocprofv3 --kernel-trace --stats.
usage: wgrad_probe.py [REPS]"""
import os
import sys
import ctypes as C
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd._lib import lib, check                # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device('cuda', 0)
B = 32
shapes = [(20, 128, 128, 3, 1), (40, 64, 64, 3, 1), (80, 64, 64, 3, 1), (160, 32, 32, 3, 1), (320, 32, 32, 3, 1),
          (80, 256, 256, 1, 1), (40, 512, 512, 1, 1), (640, 16, 32, 3, 2), (160, 64, 64, 3, 2), (20, 512, 256, 1, 1)]
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for hw, cin, cout, k, s in shapes:
    ho = hw // s
    x = torch.randn((B, hw, hw, cin), device=dev).to(torch.bfloat16)
    dy = torch.randn((B, ho, ho, cout), device=dev).to(torch.bfloat16)
    dw = torch.zeros((cout, cin, k, k), device=dev)
    need = int(lib.odamd_conv_wgrad_workspace_bytes(B, hw, hw, cin, cout, k, s, 1))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for i in range(reps + 1):
        if i == 1:
            ev[0].record()
        check(lib.odamd_conv_wgrad(x.data_ptr(), cin, dy.data_ptr(), cout, dw.data_ptr(), B, hw, hw, cin, cout, k, s, 1,
                                   ws.data_ptr(), need, stream), 'wgrad')
    ev[1].record()
    torch.cuda.synchronize()
    print('%4d %4d->%4d k%d s%d  %8.1f us  workspace %6.1f MB' % (hw, cin, cout, k, s, ev[0].elapsed_time(ev[1]) / reps * 1e3, need / 1e6))
