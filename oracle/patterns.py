"""Closed-form parameter and input patterns (test infrastructure).

Every value is a pure function of (tensor name, flat index), so the container
that holds the reference and the GPU box (which does not) fill a model with
bit-identical numbers without sharing a random generator or a weight file.
"""
import zlib

import numpy as np
import torch


def _phase(name: str) -> float:
    """Stable per-tensor phase in [0, 2*pi) derived from the tensor's name."""
    return (zlib.crc32(name.encode()) % 6283) / 1000.0


def wave(name: str, shape, scale=1.0, offset=0.0, freq=0.7310585786):
    """offset + scale*sin(freq*i + phase(name)) as float32 of the given shape."""
    n = int(np.prod(shape)) if len(shape) else 1
    i = np.arange(n, dtype=np.float64)
    v = offset + scale * np.sin(freq * i + _phase(name) + 0.001 * (i % 97))
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def fill_state_dict(module: torch.nn.Module) -> None:
    """Overwrite every parameter and buffer of `module` with the pattern.

    conv weights: amplitude sqrt(3/fan_in) (keeps activations O(1));
    BN gamma: 1 +- 0.25, beta: +-0.2, running_mean: +-0.2, running_var: 0.6..1.4;
    conv bias (head): +-0.5; num_batches_tracked left as it is.
    """
    sd = module.state_dict()
    with torch.no_grad():
        for name, t in sd.items():
            if not t.dtype.is_floating_point:
                continue
            if name.endswith('running_var'):
                v = wave(name, t.shape, 0.4, 1.0)
            elif name.endswith('running_mean'):
                v = wave(name, t.shape, 0.2)
            elif name.endswith('bn.weight'):
                v = wave(name, t.shape, 0.25, 1.0)
            elif name.endswith('bn.bias'):
                v = wave(name, t.shape, 0.2)
            elif name.endswith('.bias'):
                v = wave(name, t.shape, 0.5)
            elif t.dim() == 4:
                fan_in = t.shape[1] * t.shape[2] * t.shape[3]
                v = wave(name, t.shape, float(np.sqrt(3.0 / fan_in)))
            else:
                v = wave(name, t.shape, 0.1)
            t.copy_(v)


def image_batch(batch: int, size: int, name='image') -> torch.Tensor:
    """Synthetic (B,3,S,S) float32 image batch in [0,1)."""
    return wave(name, (batch, 3, size, size), 0.5, 0.5, freq=0.1234567).clamp_(0.0, 0.999)


def default_args(**kw):
    """Namespace with the model-side defaults of the reference CLI (run.py:29-58)."""
    import argparse
    a = argparse.Namespace(
        model_type='s', input_size=640, output_class=5, output_size=(80, 40, 20),
        output_layer=(3, 3, 3),
        anchor=(((0.02, 0.03), (0.03, 0.06), (0.06, 0.04)),
                ((0.06, 0.11), (0.11, 0.08), (0.11, 0.22)),
                ((0.22, 0.17), (0.30, 0.40), (0.72, 0.62))),
        label_smooth=0.01, device='cpu')
    for k, v in kw.items():
        setattr(a, k, v)
    if 'output_size' not in kw:
        a.output_size = (a.input_size // 8, a.input_size // 16, a.input_size // 32)
    return a
