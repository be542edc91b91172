"""Maps a demangled kernel name (as rocprofv3 prints it) to the layer family used by the per-layer
profile of engine.Plan (conv{k}x{k}_s{stride}_{c32|c64|c128}, by output-channel class), so that the
rocprofv3 summaries and the HIP-event table of bench.py speak about the same groups of launches.
Several kernels can serve one family (generic implicit GEMM, persistent 1x1 stream kernel, LDS-DMA
kernels); the class of the 3x3 LDS-DMA kernel is not visible in its name (it serves 64..256 output
channels), so it forms a group of its own."""
import re

_CG = {1: 'c32', 2: 'c64', 4: 'c128', 8: 'c128'}

OTHER = ('conv_first2_kernel', 'conv_first_wide_kernel', 'conv_first_kernel', 'compact_live_kernel', 'iou_mask_win_kernel', 'iou_mask_kernel', 'emit_heads_kernel',
         'area_order_kernel', 'nms_scan_kernel', 'nms_resolve_kernel', 'filter_kernel', 'emit_rows_kernel',
         'bitonic_lds_kernel', 'bitonic_global_kernel', 'spp_pool3_kernel', 'spp_cascade_kernel', 'pack_cols_kernel',
         'upsample2_kernel')


def family(name, conv_only=False):
    m = re.search(r'conv1x1_stream_kernel<(\d+), (\d+),', name)
    if m:
        return 'conv1x1_s1_%s' % _CG[int(m.group(2))]
    m = re.search(r'conv1x1_wres_kernel<(\d+), (\d+),', name)      # <KCH, CGW, ...>: weights resident in registers, output channels in multiples of 128
    if m:
        return 'conv1x1_s1_c128'
    m = re.search(r'conv_igemm_kernel<(\d+), (\d+), (\d+), (\d+),', name)
    if m:
        _, ks, st, cg = (int(v) for v in m.groups())
        return 'conv%dx%d_s%d_%s' % (ks, ks, st, _CG[cg])
    m = re.search(r'conv_dma_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>', name)
    if m:
        ks, st = int(m.group(1)), int(m.group(8))
        return 'conv1x1_s1_c128' if ks == 1 else 'conv3x3_s%d_dma' % st
    if 'conv_mp_kernel' in name:
        return 'mp_1x1_pair'
    if 'conv_pair_kernel' in name:
        return 'conv3x3_pair'
    m = re.search(r'conv_s3_kernel<(\d+), (\d+), (\d+),', name)
    if m:
        _, cg, st = (int(v) for v in m.groups())
        return 'conv3x3_s%d_%s' % (st, _CG[cg])
    if conv_only:
        return None
    for key in OTHER:
        if key in name:
            return key
    return None
