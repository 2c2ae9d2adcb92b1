"""ctypes binding of liboadg_hip.so (C ABI: include/oadg_hip.h).

Nothing here computes: it validates tensors, hands device pointers and the current HIP stream to the
library and raises on any non-zero return code.  A missing library is a hard error at first use.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int64, c_long, c_size_t, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('OADG_HIP_LIB') or os.path.join(_HERE, 'csrc', 'liboadg_hip.so')   # (override: A/B probes)
_lib = None

vp, ci, cf, cl, cs, cd = c_void_p, c_int, c_float, c_long, c_size_t, c_double


class RpnLevel(Structure):
    """oadg_rpn_level (include/oadg_hip.h)"""
    _fields_ = [('deltas', c_void_p), ('sN', c_long), ('sC', c_long), ('sH', c_long), ('sW', c_long), ('anchors', c_void_p),
                ('scores', c_void_p), ('index', c_void_p), ('H', c_int), ('W', c_int), ('A', c_int), ('k', c_int),
                ('dtype', c_int), ('first', c_int)]


class RpnLossLevel(Structure):
    """oadg_rpn_loss_level (include/oadg_hip.h)"""
    _fields_ = [('y', c_void_p), ('gy', c_void_p), ('sN', c_long), ('sC', c_long), ('sH', c_long), ('sW', c_long),
                ('H', c_int), ('W', c_int), ('Cy', c_int), ('first', c_long), ('pix0', c_long)]


class RoiAssignImage(Structure):
    """oadg_roi_assign_image (include/oadg_hip.h)"""
    _fields_ = [('proposals', c_void_p), ('gt_bboxes', c_void_p), ('gt_labels', c_void_p), ('stride', c_int),
                ('num_gts', c_int)]


ROI_ASSIGN_MAX_IMAGES = 32


class RoiTargetEntry(Structure):
    """oadg_roi_target_entry (include/oadg_hip.h)"""
    _fields_ = [('bboxes', c_void_p), ('gt_bboxes', c_void_p), ('gt_inds', c_void_p), ('labels', c_void_p),
                ('pos_inds', c_void_p), ('neg_inds', c_void_p), ('npos', c_int), ('nneg', c_int), ('stride', c_int),
                ('batch', c_int)]


ROI_TARGET_MAX_ENTRIES = 32


class RoiSampleImage(Structure):
    """oadg_roi_sample_image (include/oadg_hip.h)"""
    _fields_ = [('gt_inds', c_void_p), ('n', c_int)]


ROI_SAMPLE_MAX_IMAGES = 8


class RegionOp(Structure):
    """oadg_region_op (include/oadg_hip.h)"""
    _fields_ = [('kind', c_int), ('param', c_int), ('image', c_void_p), ('minv', c_double * 6)]


OP_COPY, OP_LUT_AUTOCONTRAST, OP_LUT_EQUALIZE, OP_POSTERIZE, OP_SOLARIZE, OP_IMAGE, OP_BG_WARP, OP_WARP_NEG = range(8)
OP_ENH_BRIGHTNESS, OP_ENH_COLOR, OP_ENH_CONTRAST, OP_ENH_SHARPNESS = 8, 9, 10, 11

# name -> (restype, argtypes); must list every symbol include/oadg_hip.h declares
SIGNATURES = {
    'oadg_supcon_workspace_bytes': (cs, [ci, ci]),
    'oadg_supcon_fwd': (ci, [vp, vp, ci, ci, ci, ci, ci, cf, ci, cf, vp, cs, vp, vp]),
    'oadg_supcon_bwd': (ci, [vp, ci, ci, ci, ci, ci, cf, cf, vp, vp, cs, vp, vp]),
    'oadg_supcon_status': (ci, [vp]),
    'oadg_cls_loss_workspace_bytes': (cs, []),
    'oadg_ce_jsd_fwd': (ci, [vp, vp, vp, cl, ci, ci, cf, cf, cf, vp, cs, vp, vp]),
    'oadg_ce_jsd_bwd': (ci, [vp, vp, vp, cl, ci, ci, cf, cf, cf, vp, vp, vp]),
    'oadg_roi_align_fwd': (ci, [POINTER(vp), POINTER(ci), POINTER(ci), POINTER(cf), ci, ci, ci, ci, cf,
                                vp, ci, ci, ci, ci, ci, vp, vp, vp]),
    'oadg_roi_align_bwd': (ci, [POINTER(vp), POINTER(ci), POINTER(ci), POINTER(cf), ci, ci, ci, ci, cf,
                                vp, ci, ci, ci, ci, ci, vp, vp, vp]),
    'oadg_roi_order_keys': (ci, [vp, ci, ci, ci, cf, vp, vp]),
    'oadg_roi_order': (ci, [vp, ci, ci, ci, cf, vp, vp, vp]),
    'oadg_roi_align_bwd_tiles': (ci, [POINTER(vp), POINTER(ci), POINTER(ci), POINTER(cf), ci, ci, ci, cf, vp, ci, ci, ci,
                                      ci, ci, vp, vp, vp, vp, vp]),
    'oadg_rpn_loss_workspace_bytes': (cs, []),
    'oadg_rpn_loss_fwd': (ci, [vp, ci, ci, ci, cl, ci, vp, vp, vp, vp, cf, cf, cf, cf, vp, cs, vp, vp]),
    'oadg_rpn_loss_bwd': (ci, [vp, ci, ci, ci, cl, ci, vp, vp, vp, vp, cf, cf, cf, cf, vp, vp, vp]),
    'oadg_rpn_decode': (ci, [vp, ci, ci, vp, vp, cf, vp, ci, cf, vp, vp, vp, vp]),
    'oadg_rpn_order': (ci, [vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp]),
    'oadg_rpn_gather': (ci, [ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]),
    'oadg_rpn_topk_workspace_bytes': (cs, [POINTER(ci), ci, ci, ci]),
    'oadg_rpn_topk': (ci, [POINTER(vp), POINTER(cl), POINTER(ci), ci, ci, ci, ci, POINTER(vp), POINTER(vp), vp, cs, vp]),
    'oadg_nms_workspace_bytes': (cs, [ci, ci]),
    'oadg_nms_batched': (ci, [vp, vp, ci, ci, cf, ci, vp, cs, vp, vp, vp]),
    'oadg_resize_bilinear_u8': (ci, [vp, ci, ci, ci, vp, ci, ci, vp]),
    'oadg_flip_u8': (ci, [vp, ci, ci, ci, vp, ci, vp]),
    'oadg_conv2d_nhwc_bf16': (ci, [vp, vp, vp, vp, vp, vp] + [ci] * 11 + [vp]),
    'oadg_conv2d_auto_variant': (ci, [ci] * 10),
    'oadg_conv2d_nhwc_bf16_ex': (ci, [vp, vp, vp, vp, vp, vp] + [ci] * 12 + [vp, vp, vp, vp, vp]),
    'oadg_conv2d_pixel_tiles': (cl, [ci] * 11),
    'oadg_colsum_reduce': (ci, [vp, cl, ci, vp, vp]),
    'oadg_conv2d_nhwc_bf16_variant': (ci, [vp, vp, vp, vp, vp, vp] + [ci] * 12 + [vp]),
    'oadg_conv2d_wgrad_workspace_bytes': (cs, [ci] * 7),
    'oadg_conv2d_wgrad_variant': (ci, [ci] * 7),
    'oadg_conv2d_wgrad_nhwc_bf16': (ci, [vp, vp, vp, vp, vp, cs] + [ci] * 10 + [vp]),
    'oadg_conv2d_wgrad_parts_nhwc_bf16': (ci, [vp, vp, vp, vp, cs] + [ci] * 10 + [POINTER(ci), vp]),
    'oadg_prep_conv_weights_bwd_parts': (ci, [vp, ci, vp, vp, vp, vp, vp, cf, ci, ci, ci, ci, vp, vp, ci, vp]),
    'oadg_prep_conv_weights': (ci, [vp, vp, vp, vp, vp, cf, vp, ci, ci, ci, ci, vp, vp, vp, vp, ci, ci, vp]),
    'oadg_prep_conv_weights_multi': (ci, [vp, ci, ci, vp]),
    'oadg_prep_conv_weights_multi_blocks': (ci, [ci, ci, ci, ci]),
    'oadg_conv2d_nhwc_bf16_scatter': (ci, [vp, vp, vp, vp, vp, vp] + [ci] * 18 + [vp, vp, vp, vp]),
    'oadg_conv2d_dgrad_s2_nhwc_bf16': (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp]),
    'oadg_prep_conv_weights_bwd': (ci, [vp, vp, vp, vp, vp, vp, cf, ci, ci, ci, ci, vp, vp, ci, vp]),
    'oadg_relu_bias_bwd_workspace_bytes': (ctypes.c_size_t, [cl, ci]),
    'oadg_relu_bias_bwd': (ci, [vp, ci, vp, vp, vp, vp, ctypes.c_size_t, cl, ci, vp]),
    'oadg_stem_conv7x7s2_nhwc_bf16': (ci, [vp, vp, vp, ci, ci, ci, vp]),
    'oadg_bias_relu_maxpool_nhwc_bf16': (ci, [vp, vp, vp, ci, ci, ci, ci, vp]),
    'oadg_fpn_topdown_fwd': (ci, [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]),
    'oadg_fpn_topdown_bwd': (ci, [vp, vp, ci, ci, ci, ci, ci, ci, vp]),
    'oadg_max_iou_assign_workspace_bytes': (ctypes.c_size_t, [ci, ci]),
    'oadg_max_iou_assign': (ci, [vp, cl, vp, vp, vp, vp, ci, ci, ci, cf, cf, cf, cf, ci, vp, ctypes.c_size_t, vp, vp, vp,
                                 vp, vp]),
    'oadg_sample_select_workspace_bytes': (ctypes.c_size_t, [ci, cl]),
    'oadg_sample_select': (ci, [vp, ci, cl, ci, vp, vp, vp, ctypes.c_size_t, vp]),
    'oadg_anchor_targets': (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, c_int64, cf, vp, vp, vp, vp, vp, vp, vp]),
    'oadg_colsum_reduce_multi': (ci, [vp, ci, ci, vp]),
    'oadg_conv1x1_n16_fwd': (ci, [vp, vp, vp, vp, cl, ci, vp]),
    'oadg_conv1x1_n16_dgrad_rows': (cl, [cl]),
    'oadg_conv1x1_n16_dgrad': (ci, [vp, vp, vp, vp, vp, cl, ci, vp]),
    'oadg_bottleneck_frozen_256': (ci, [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]),
    'oadg_bottleneck_frozen_first_64': (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]),
    'oadg_conv1x1_n16_wgrad_splits': (cl, [cl]),
    'oadg_conv1x1_n16_wgrad_rows': (cl, [cl]),
    'oadg_conv1x1_n16_wgrad': (ci, [vp, vp, vp, vp, vp, cl, ci, vp]),
    'oadg_conv2d_wgrad_multi_plan': (ctypes.c_long, [vp, ci, ci, vp]),
    'oadg_conv2d_wgrad_multi': (ci, [vp, ci, ci, vp, vp, vp]),
    'oadg_prep_conv_weights_bwd_parts_multi': (ci, [vp, ci, ci, ci, vp]),
    'oadg_sgd_blocks': (ctypes.c_longlong, [ctypes.c_longlong]),
    'oadg_sgd_step_multi': (ci, [vp, ci, ctypes.c_longlong, cf, cf, cf, vp]),
    'oadg_fc_weight_permute': (ci, [vp, vp, ci, ci, ci, ci, vp]),
    'oadg_parse_losses': (ci, [vp, vp, ci, ci, ctypes.c_uint, vp, vp, vp]),
    'oadg_roi_reg_acc_fwd': (ci, [vp, ci, vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, cf, cf, cf, vp, vp]),
    'oadg_roi_reg_bwd': (ci, [vp, ci, vp, vp, vp, ci, ci, ci, ci, cf, cf, cf, vp, vp, vp]),
    'oadg_roi_assign_add_gt': (ci, [vp, ci, ci, ci, cf, cf, cf, cf, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_size_t,
                                    vp, vp]),
    'oadg_roi_targets': (ci, [vp, ci, ci, c_int64, cf, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    'oadg_roi_targets_dev': (ci, [vp, ci, ci, ci, vp, c_int64, cf, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    'oadg_roi_sample_max_rows': (ci, []),
    'oadg_roi_sample_device': (ci, [vp, ci, ci, ci, cf, vp, vp, vp, vp, vp, vp]),
    'oadg_conv2d_f32': (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp]),
    'oadg_conv2d_wgrad_f32_splits': (ci, [ci, ci, ci, ci, ci, ci, ci]),
    'oadg_conv2d_wgrad_f32': (ci, [vp, vp, vp, vp, cs, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp]),
    'oadg_host_randperm_prefix': (ci, [vp, POINTER(ci), POINTER(ctypes.c_uint64), c_int64, c_int64, vp]),
    'oadg_np_random_bboxes': (ci, [vp, ci, ci, ci, vp, ci, cd, cd, cd, cd, ci, cd, cd, vp]),
    'oadg_oamix_box_profiles': (ci, [vp, vp, ci, ci, ci, ci, vp, vp, vp]),
    'oadg_oamix_fg_union': (ci, [vp, vp, ci, ci, ci, vp, vp, vp]),
    'oadg_oamix_fg_union_rects': (ci, [vp, vp, vp, ci, ci, ci, vp, vp, vp]),
    'oadg_oamix_saliency_workspace_bytes': (cs, [ci]),
    'oadg_oamix_saliency': (ci, [vp, ci, ci, vp, ci, ci, vp, vp, cs, vp]),
    'oadg_oamix_saliency_batch': (ci, [vp, ctypes.c_longlong, vp, ci, ci, vp, ci, ci, vp, vp, ctypes.c_size_t, vp]),
    'oadg_oamix_hist': (ci, [vp, cl, vp, vp]),
    'oadg_oamix_luts': (ci, [vp, vp, vp]),
    'oadg_oamix_gray_sum': (ci, [vp, cl, vp, vp]),
    'oadg_oamix_bbox_step': (ci, [vp, ci, ci, POINTER(cd), ci, ci, ci, ci, vp, vp, vp, vp]),
    'oadg_oamix_bbox_levels': (ci, [POINTER(ci), POINTER(cd), ci, ci, ci, POINTER(ci)]),
    'oadg_oamix_bbox_plan_bytes': (cs, [ci]),
    'oadg_oamix_bbox_plan': (ci, [ci, cd, vp, vp, ci, vp, ci, ci, ci, vp, cs, vp, vp, vp]),
    'oadg_oamix_bbox_chain': (ci, [vp, ci, ci, vp, vp, POINTER(ci), ci, POINTER(ci), vp, vp, vp, vp]),
    'oadg_oamix_bbox_chain_multi': (ci, [vp, ci, vp]),
    'oadg_png_size': (ci, [c_char_p, POINTER(ci), POINTER(ci)]),
    'oadg_png_decode_bgr': (ci, [c_char_p, vp, ci, ci]),
    'oadg_glass_shuffle_u8': (ci, [vp, ci, ci, ci, ci, ci, vp]),
    'oadg_chamfer_l2_5x5': (ci, [vp, ci, ci, vp]),
    'oadg_oamix_compose': (ci, [vp, vp, ci, ci, POINTER(RegionOp), POINTER(ci), ci, vp, vp, vp, vp, cf, ci, vp]),
    'oadg_oamix_final': (ci, [vp, vp, ci, ci, vp, ci, vp, vp, cd, POINTER(cf), POINTER(cf), ci, vp, vp, ci, ci,
                              ci, vp]),
    'oadg_oamix_final_tiles_workspace_bytes': (ctypes.c_size_t, [ci, ci, ci]),
    'oadg_oamix_final_tiles': (ci, [vp, vp, ci, ci, vp, ci, vp, vp, vp, cd, POINTER(cf), POINTER(cf), ci, vp, vp, ci, ci,
                                    ci, vp, ctypes.c_size_t, vp]),
    'oadg_oamix_normalize': (ci, [vp, ci, ci, POINTER(cf), POINTER(cf), ci, vp, ci, ci, ci, vp]),
}


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raise loudly if the library was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'(or `make -C oa-dg_amd/csrc`). There is no CPU fallback for the OA-DG hot ops.')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        kind = 'argument error' if rc < 0 else 'hipError_t'
        raise RuntimeError(f'{what} failed: {kind} {rc}')


def ptr(t):
    """device pointer of a tensor (or of anything exposing data_ptr()); None -> NULL"""
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def stream_ptr():
    """raw hipStream_t of torch's current stream on the current device"""
    import torch
    return c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('OA-DG hot ops run on the MI355X only (got a CPU tensor); '
                               'there is no CPU fallback')
