"""ctypes binding of libodamd.so (the C ABI of include/odamd.h).

The library is built in-tree by objectdetection_amd/csrc/Makefile.  Loading fails loudly when
it is absent: nothing in this package computes on the CPU instead.
"""
import ctypes as C
import os

# torch ships its own HIP runtime (torch/lib/libamdhip64.so).  It must be the first one loaded
# into the process so that this library binds to the SAME runtime instance (streams and device
# pointers are shared with torch); loading /opt/rocm's copy first leaves torch without a device.
import torch  # noqa: F401  (import order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libodamd.so')
ABI_VERSION = 3

if not os.path.exists(LIB_PATH):
    raise ImportError(f'{LIB_PATH} not found: build it with `make -C objectdetection_amd/csrc` '
                      '(or `python -c "import __graft_entry__ as g; g.build()"`). There is no fallback path.')
lib = C.CDLL(LIB_PATH)

c_int, c_float, c_void_p, c_size_t = C.c_int, C.c_float, C.c_void_p, C.c_size_t


class DecodeDesc(C.Structure):
    """odamd_decode_desc"""
    _fields_ = [('head', c_void_p * 3), ('h', c_int * 3), ('w', c_int * 3), ('stride', c_float * 3),
                ('anchor_px', ((c_float * 2) * 4) * 3), ('n_anchor', c_int), ('batch', c_int), ('kdim', c_int),
                ('deploy', c_int)]


class ConvDesc(C.Structure):
    """odamd_conv_desc"""
    _fields_ = [('x', c_void_p), ('w', c_void_p), ('scale', c_void_p), ('shift', c_void_p), ('y', c_void_p),
                ('batch', c_int), ('h', c_int), ('w_in', c_int), ('cin', c_int), ('cout', c_int),
                ('x_ld', c_int), ('y_ld', c_int), ('ksize', c_int), ('stride', c_int), ('act', c_int),
                ('dtype', c_int), ('in_pool2', c_int), ('out_nchw_f32', c_int), ('in_upzero2', c_int),
                ('accumulate', c_int), ('w_packed', c_int)]


class PackDesc(C.Structure):
    """One entry of the odamd_pack_weights table."""
    _fields_ = [('w', c_void_p), ('fwd', c_void_p), ('bwd', c_void_p), ('cout', c_int), ('cin', c_int), ('k', c_int),
                ('cin_phys', c_int), ('cout_phys', c_int), ('n_seg', c_int), ('seg_off', c_int * 8),
                ('seg_len', c_int * 8), ('fwd_f32', c_int), ('fwd_frag', c_int), ('bwd_frag', c_int), ('reserved', c_int)]


# name -> (restype, argtypes); every symbol include/odamd.h declares must appear here.
SIGNATURES = {
    'odamd_abi_version': (c_int, []),
    'odamd_error_string': (C.c_char_p, [c_int]),
    'odamd_decode': (c_int, [C.POINTER(DecodeDesc), c_void_p, c_void_p]),
    'odamd_deploy_sigmoid': (c_int, [c_void_p, c_size_t, c_int, c_void_p]),
    'odamd_decode_backward': (c_int, [C.POINTER(DecodeDesc), c_void_p, C.POINTER(c_void_p * 3), c_void_p]),
    'odamd_postprocess_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'odamd_postprocess': (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_int, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'odamd_postprocess_heads': (c_int, [C.POINTER(DecodeDesc), c_float, c_float, c_void_p, c_int, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'odamd_nms_rows': (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_int,
                               c_void_p, c_size_t, c_void_p]),
    'odamd_nms_rows_static': (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_int,
                                      c_void_p]),
    'odamd_metric_nms': (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_float, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'odamd_metric_nms_rows': (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                      c_void_p, c_size_t, c_void_p]),
    'odamd_metric_match': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p,
                                   c_void_p]),
    'odamd_iou_pairs': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    'odamd_iou_mask': (c_int, [c_void_p, c_int, c_float, c_void_p, c_void_p]),
    'odamd_conv_forward': (c_int, [C.POINTER(ConvDesc), c_void_p]),
    'odamd_conv_weight_layout': (c_int, [C.POINTER(ConvDesc), C.POINTER(c_int), C.POINTER(c_int)]),
    'odamd_conv_mp_supported': (c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc)]),
    'odamd_conv_mp_forward': (c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), c_void_p]),
    'odamd_conv_chain_pair_supported': (c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc)]),
    'odamd_conv_chain_pair_forward': (c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), c_void_p]),
    'odamd_conv_first_forward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                         c_int, c_int, c_int, c_int, c_void_p]),
    'odamd_conv_first2_forward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_int, c_int, c_int, c_int, c_void_p]),
    'odamd_maxpool2_forward': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'odamd_spp_pool3_forward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                        c_int, c_int, c_void_p]),
    'odamd_upsample2_forward': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'odamd_bn_workspace_bytes': (c_size_t, [c_int]),
    'odamd_bn_train_forward': (c_int, [c_void_p, c_int, c_void_p, c_int, c_size_t, c_int, c_void_p, c_void_p, c_float,
                                       c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                       c_void_p, c_size_t, c_void_p]),
    'odamd_bn_silu_backward': (c_int, [c_void_p, c_int, c_void_p, c_int, c_size_t, c_int, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'odamd_bias_grad': (c_int, [c_void_p, c_int, c_size_t, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'odamd_maxpool2_backward': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                        c_int, c_void_p]),
    'odamd_maxpool_s1_backward': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                          c_int, c_int, c_void_p]),
    'odamd_upsample2_backward': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'odamd_spp_backward': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'odamd_conv_wgrad_workspace_bytes': (c_size_t, [c_int] * 8),
    'odamd_conv_wgrad': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                 c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'odamd_loss_workspace_bytes': (c_size_t, []),
    'odamd_detection_loss': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                     c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    'odamd_detection_loss_dev': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                         c_int, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    'odamd_preprocess_u8': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    'odamd_preprocess_u8_batch': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    'odamd_assign_labels': (c_int, [C.POINTER(DecodeDesc), c_float, c_float, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                    c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'odamd_adam_ema_step': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, C.c_double, C.c_double,
                                    C.c_double, C.c_double, C.c_double, c_int, C.c_double, C.c_double, c_void_p]),
    'odamd_ema_update': (c_int, [c_void_p, c_void_p, c_size_t, C.c_double, c_void_p]),
    'odamd_pack_weights': (c_int, [c_void_p, c_int, c_int, c_void_p]),
    'odamd_pack_desc_bytes': (c_size_t, []),
    'odamd_nchw_to_nhwc': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'odamd_nhwc_to_nchw': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
}

MISSING = []
for _name, (_res, _args) in SIGNATURES.items():
    try:
        _fn = getattr(lib, _name)
    except AttributeError:
        MISSING.append(_name)
        continue
    _fn.restype = _res
    _fn.argtypes = _args

if 'odamd_abi_version' in MISSING or lib.odamd_abi_version() != ABI_VERSION:
    raise ImportError(f'{LIB_PATH}: ABI version mismatch, rebuild the library')


def check(code, what=''):
    """Raise RuntimeError for a non-zero return code of an odamd_* call."""
    if code != 0:
        msg = lib.odamd_error_string(code).decode()
        raise RuntimeError(f'libodamd {what} failed ({code}): {msg}')
