"""objectdetection_amd -- MI355X (gfx950) native YOLOv7 hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all
arithmetic on the path runs in hand-written HIP kernels of libodamd.so, reached through the
C ABI declared in include/odamd.h.  There is no CPU or stock-PyTorch fallback: importing
`objectdetection_amd._lib` raises if the library is missing.
"""
__version__ = '0.1.0'
