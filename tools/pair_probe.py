"""Time of odamd_conv_chain_pair_forward alone (random data, concat-slice addressing) against the two odamd_conv_forward
launches, per shape.  usage: pair_probe.py"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import _lib
from objectdetection_amd.engine import _pack_fragments
lib = _lib.lib


def desc(x_ptr, y_ptr, w, sc, sh, B, H, W, cin, ld, act, packed):
    d = _lib.ConvDesc()
    d.x, d.w, d.scale, d.shift, d.y = x_ptr, w.data_ptr(), sc.data_ptr(), sh.data_ptr(), y_ptr
    d.batch, d.h, d.w_in, d.cin, d.cout = B, H, W, cin, 32
    d.x_ld, d.y_ld, d.ksize, d.stride, d.act, d.dtype = ld, ld, 3, 1, act, 1
    d.w_packed = packed
    return d


def timed(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B, H, W, cin, ld in ((64, 160, 160, 32, 128), (64, 80, 80, 64, 256), (64, 80, 80, 32, 256), (64, 160, 160, 32, 32)):
    buf = torch.randn(B, H, W, ld, device='cuda').bfloat16()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    w1 = (torch.randn(32, 3, 3, cin, device='cuda') * 0.1).bfloat16()
    w2 = (torch.randn(32, 3, 3, 32, device='cuda') * 0.1).bfloat16()
    f1, f2 = _pack_fragments(w1, 1, cin * 2), _pack_fragments(w2, 1, 64)
    sc, sh = torch.ones(32, device='cuda'), torch.zeros(32, device='cuda')
    x0 = buf.data_ptr()
    for act in (1, 0):
        if ld == 32:
            buf2, buf3 = torch.empty_like(buf), torch.empty_like(buf)
            y1p, y2p = buf2.data_ptr(), buf3.data_ptr()
        else:
            y1p, y2p = x0 + cin * 2, x0 + cin * 2 + 64
        da, db = desc(x0, y1p, f1, sc, sh, B, H, W, cin, ld, act, 1), desc(y1p, y2p, f2, sc, sh, B, H, W, 32, ld, act, 1)
        assert lib.odamd_conv_chain_pair_supported(C.byref(da), C.byref(db))
        sa, sb = desc(x0, y1p, w1, sc, sh, B, H, W, cin, ld, act, 0), desc(y1p, y2p, w2, sc, sh, B, H, W, 32, ld, act, 0)

        def pair():
            rc = lib.odamd_conv_chain_pair_forward(C.byref(da), C.byref(db), st)
            assert rc == 0, rc

        def two():
            assert lib.odamd_conv_forward(C.byref(sa), st) == 0
            assert lib.odamd_conv_forward(C.byref(sb), st) == 0
        print(f'{H}x{W} b{B} {cin}->32->32 ld {ld} act {act}: pair {timed(pair):7.1f} us   two launches {timed(two):7.1f} us', flush=True)
