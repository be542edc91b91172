"""The C-ABI library must load on a machine without a GPU and export every declared symbol."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'odamd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(odamd_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from objectdetection_amd import _lib
    names = _declared()
    assert len(names) >= 15
    assert _lib.MISSING == []
    for n in names:
        assert hasattr(_lib.lib, n), n
        assert n in _lib.SIGNATURES, f'{n} has no ctypes signature'
    assert sorted(_lib.SIGNATURES) == names


def test_error_strings_and_version():
    from objectdetection_amd import _lib
    assert _lib.lib.odamd_abi_version() == _lib.ABI_VERSION
    assert b'workspace' in _lib.lib.odamd_error_string(-3)
    assert _lib.lib.odamd_postprocess_workspace_bytes(2, 25200, 85, 0) > 0


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any HIP call, so this runs on a CPU-only box."""
    import ctypes as C
    from objectdetection_amd import _lib
    d = _lib.ConvDesc()
    assert _lib.lib.odamd_conv_forward(C.byref(d), None) == -1
    assert _lib.lib.odamd_postprocess(None, 1, 10, 6, 0.5, 0.5, None, 0, None, None, None, None, 10, None, 0, None) == -1
    assert _lib.lib.odamd_iou_mask(None, 64, 0.5, None, None) == -1


def test_state_dict_names_match_reference_layout():
    """474 entries for preset s (counted on the reference's own module), fixed key examples."""
    from objectdetection_amd.model.yolov7 import yolov7
    from oracle import patterns
    m = yolov7(patterns.default_args(input_size=64, output_class=80))
    sd = m.state_dict()
    assert len(sd) == 474
    for k in ('cbs0.conv.weight', 'cbs0.bn.running_mean', 'elan4.sequential2.0.conv.weight', 'mp5.cbs3.bn.bias',
              'sppcspc11.cbs11.conv.weight', 'l8_concat.bn.num_batches_tracked', 'elan_h25.cbs7.conv.weight',
              'head2.output.bias'):
        assert k in sd, k
    assert sum(p.numel() for p in m.parameters()) == 7010957


def test_pruned_constructor_matches_reference_fixture():
    """yolov7(args, config=widths): same width list, state_dict keys and shapes as the reference built
    from the same list (tests/golden/pruned.npz)."""
    import os
    import numpy as np
    import torch
    from oracle import patterns
    from objectdetection_amd.model import layer
    from objectdetection_amd.model.yolov7 import yolov7
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'pruned.npz'))
    args = patterns.default_args(input_size=64, output_class=3)
    std = yolov7(args)
    widths = [m.conv.out_channels for m in std.modules() if isinstance(m, layer.cbs)]
    assert widths == g['widths'].tolist()
    net = yolov7(args, config=g['cfg'].tolist())
    sd = net.state_dict()
    assert list(sd.keys()) == g['keys'].tolist()
    for k, shp in zip(g['keys'].tolist(), g['shapes'].tolist()):
        assert tuple(sd[k].shape) == tuple(v for v in shp[:sd[k].dim()]), k
