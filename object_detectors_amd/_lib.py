"""ctypes binding of libmi355det.so — the only way the Python mirror reaches the GPU.

There is NO fallback: if the library is missing or a call fails, an exception is raised.
Signatures mirror include/mi355det.h one to one.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355DET_LIB", os.path.join(HERE, "libmi355det.so"))      # the override serves same-box A/B runs of two builds

MAX_SCALES, MAX_ANCHORS = 4, 8

vp, i32, i64, f32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t


class YoloGeom(C.Structure):
    _fields_ = [("num_scales", i32), ("na", i32), ("num_classes", i32), ("pad0", i32),
                ("img_size", f32), ("ignore_thr", f32), ("iou_type", i32), ("pad1", i32),
                ("grid", i32 * MAX_SCALES), ("off", i32 * (MAX_SCALES + 1)),
                ("anchor_w", (f32 * MAX_ANCHORS) * MAX_SCALES), ("anchor_h", (f32 * MAX_ANCHORS) * MAX_SCALES)]


class HeadView(C.Structure):
    _fields_ = [("ptr", vp), ("sb", i64), ("sc", i64), ("sp", i64)]


class YoloLossCfg(C.Structure):
    _fields_ = [("lambda_iou", f32), ("lambda_xy", f32), ("lambda_wh", f32), ("lambda_conf", f32),
                ("lambda_no_conf", f32), ("lambda_cls", f32), ("alpha", f32), ("gamma", f32),
                ("grad_scale", f32), ("grad_is_bf16", i32), ("class_weights", vp), ("class_loss", i32), ("reduction_mean", i32),
                ("eq_mask", vp)]


class ConvShape(C.Structure):
    _fields_ = [("n", i32), ("h", i32), ("w", i32), ("cin", i32), ("ho", i32), ("wo", i32), ("cout", i32),
                ("ksize", i32), ("stride", i32), ("pad", i32), ("in_ld", i32), ("out_ld", i32)]


class ConvEpilogue(C.Structure):
    _fields_ = [("scale", vp), ("shift", vp), ("residual", vp), ("residual_ld", i32), ("relu", i32), ("out_image_stride", i64),
                ("slope", C.c_float)]


class LevelGrads(C.Structure):
    """mi355det_level_grads: where mi355det_retina_loss_lv writes the bf16 class gradient (one NHWC buffer per pyramid level)."""
    _fields_ = [("n_levels", i32), ("anchors_per_pixel", i32), ("pixels", i64 * 8), ("grad", vp * 8), ("grad_ld", i32 * 8)]


class PackItem(C.Structure):
    _fields_ = [("w", vp), ("w_fwd", vp), ("w_dgrad", vp), ("shape", ConvShape), ("cout_pad", i32), ("w_is_ohwi", i32)]


P = C.POINTER
PROTOTYPES = {
    # name: (restype, argtypes)
    "mi355det_last_error": (C.c_char_p, []),
    "mi355det_version": (C.c_int, []),
    "mi355det_debug_set": (C.c_int, [C.c_int, C.c_int]),
    "mi355det_debug_ptr": (C.c_int, [C.c_int, vp]),
    "mi355det_bbox_iou": (C.c_int, [vp, vp, vp, i64, i64, C.c_int, C.c_int, C.c_int, vp]),
    "mi355det_yolo_assign": (C.c_int, [P(YoloGeom), vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]),
    "mi355det_yolo_loss_workspace": (sz, [i32, i64]),
    "mi355det_yolo_loss": (C.c_int, [P(YoloGeom), P(YoloLossCfg), P(HeadView), P(HeadView), vp, vp, vp, vp, vp, vp,
                                      i32, i32, vp, sz, vp, vp]),
    "mi355det_yolo_decode": (C.c_int, [P(YoloGeom), P(HeadView), vp, i32, C.c_int, vp, vp, vp, vp]),
    "mi355det_yolo_candidates_workspace": (sz, [i32, i64]),
    "mi355det_yolo_candidates": (C.c_int, [vp, vp, vp, i32, i64, i32, f32, vp, vp, i32, vp, sz, vp]),
    "mi355det_nms_workspace": (sz, [i32, i32]),
    "mi355det_nms_majority": (C.c_int, [vp, vp, i32, i32, f32, i32, vp, vp, vp, vp, sz, vp]),
    "mi355det_box_iou": (C.c_int, [vp, vp, vp, i64, i64, vp]),
    "mi355det_nms": (C.c_int, [vp, vp, vp, i32, f32, vp, vp, vp, sz, vp]),
    "mi355det_nms_batch": (C.c_int, [vp, vp, vp, i32, i32, C.c_float, vp, vp, vp, sz, vp]),
    "mi355det_topk_segments": (C.c_int, [vp, i32, i64, i32, vp, vp, vp, f32, vp, vp, vp, vp, sz, vp]),
    "mi355det_rpn_proposals_workspace": (sz, [i32, vp, i32, i32]),
    "mi355det_rpn_proposals": (C.c_int, [vp, vp, vp, vp, i32, vp, i32, i32, i32, f32, f32, f32, f32, vp, vp, vp, vp, sz, vp]),
    "mi355det_retina_detections_workspace": (sz, [i32, vp, i32, i32, i32]),
    "mi355det_retina_detections": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, vp, f32, i32, f32, i32, f32, vp, vp, vp, vp, vp, sz, vp]),
    "mi355det_roi_detections_workspace": (sz, [i32, i32, i32, i32]),
    "mi355det_roi_detections": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, f32, i32, f32, f32, f32, f32, f32, f32, f32, i32, vp, vp, vp, vp, vp, vp, sz, vp]),
    "mi355det_rpn_loss": (C.c_int, [vp, vp, vp, vp, i64, vp, i32, vp, i32, vp, vp, vp, vp]),
    "mi355det_roi_match": (C.c_int, [vp, vp, i32, i32, vp, vp, vp, f32, f32, i32, vp, vp, vp, vp]),
    "mi355det_roi_sample": (C.c_int, [vp, vp, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, f32, f32, f32, f32, vp, vp, vp, vp, vp]),
    "mi355det_match_anchors": (C.c_int, [vp, vp, i32, i64, f32, f32, C.c_int, vp, vp, vp]),
    "mi355det_box_encode": (C.c_int, [vp, vp, vp, i64, f32, f32, f32, f32, vp]),
    "mi355det_box_decode": (C.c_int, [vp, vp, vp, i64, i32, f32, f32, f32, f32, f32, vp]),
    "mi355det_anchor_grid": (C.c_int, [vp, i32, i32, i32, i32, i32, vp, vp]),
    "mi355det_sigmoid_focal_loss": (C.c_int, [vp, vp, vp, vp, i64, i32, f32, f32, f32, vp, vp, vp]),
    "mi355det_sigmoid_focal_loss_elem": (C.c_int, [vp, vp, i64, f32, f32, vp, vp, vp]),
    "mi355det_retina_cls_loss": (C.c_int, [vp, vp, vp, vp, i64, i32, f32, f32, f32, vp, vp, vp]),
    "mi355det_roi_align_nhwc": (C.c_int, [vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, C.c_int, i32, i32, vp, vp, vp, vp]),
    "mi355det_roi_align": (C.c_int, [vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, C.c_int, i32, i32, vp, vp, vp, vp]),
    "mi355det_topk": (C.c_int, [vp, i32, i64, i64, i32, f32, vp, vp, vp, vp]),
    "mi355det_conv_fwd": (C.c_int, [P(ConvShape), vp, vp, vp, vp, C.c_int, vp, i32, vp]),
    "mi355det_conv_dgrad": (C.c_int, [P(ConvShape), vp, vp, vp, vp, i32, vp]),
    "mi355det_conv_dgrad_mask": (C.c_int, [P(ConvShape), vp, vp, vp, vp, i32, vp, i32, vp]),
    "mi355det_conv_dgrad_workspace": (sz, [P(ConvShape)]),
    "mi355det_conv_dgrad_ws": (C.c_int, [P(ConvShape), vp, vp, vp, vp, i32, vp, sz, vp]),
    "mi355det_conv_wgrad_workspace": (sz, [P(ConvShape)]),
    "mi355det_conv_wgrad_autotune": (C.c_int, [P(ConvShape), vp, vp, vp, vp, sz, vp]),
    "mi355det_conv_wgrad": (C.c_int, [P(ConvShape), vp, vp, vp, vp, vp, sz, vp]),
    "mi355det_dgrad_pack_elems": (sz, [P(ConvShape)]),
    "mi355det_pack_weights": (C.c_int, [P(ConvShape), vp, C.c_int, vp, i32, vp, vp]),
    "mi355det_pack_table_bytes": (sz, [P(PackItem), i32, P(i32), P(i32)]),
    "mi355det_pack_table_build": (C.c_int, [P(PackItem), i32, vp, sz]),
    "mi355det_pack_weights_batched": (C.c_int, [vp, i32, i32, vp]),
    "mi355det_unpack_wgrad": (C.c_int, [P(ConvShape), vp, vp, vp]),
    "mi355det_conv_autotune_mode": (C.c_int, [C.c_int]),
    "mi355det_tune_export": (sz, [vp, sz]),
    "mi355det_tune_import": (C.c_int, [vp, sz, C.c_int]),
    "mi355det_tune_lock": (C.c_int, [C.c_int]),
    "mi355det_tune_clear": (C.c_int, []),
    "mi355det_conv_stats_rows": (C.c_int, [P(ConvShape), i32]),
    "mi355det_stem_im2col": (C.c_int, [vp, vp, i32, i32, i32, vp]),
    "mi355det_stem_rows": (C.c_int, [i32, i32, i32]),
    "mi355det_stem_l1_rows": (C.c_int, [i32, i32, i32]),
    "mi355det_stem_l1_fwd": (C.c_int, [vp, vp, vp, f32, vp, vp, i32, vp, i32, vp, i32, i32, i32, vp]),
    "mi355det_stem_l1_fwd_eval": (C.c_int, [vp, vp, vp, f32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "mi355det_stem_fwd_stats": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
    "mi355det_stem_fwd_apply": (C.c_int, [vp, vp, vp, f32, vp, i32, i32, i32, i32, vp]),
    "mi355det_stem_bwd_reduce": (C.c_int, [vp, vp, vp, f32, vp, i32, vp, i32, i32, i32, vp]),
    "mi355det_stem_bwd_apply_wgrad": (C.c_int, [vp, vp, vp, vp, f32, vp, i32, vp, vp, vp, vp, i32, i32, i32, vp]),
    "mi355det_stem_bwd_fused": (C.c_int, [vp, vp, vp, f32, vp, i32, vp, vp, vp, i32, i32, i32, vp]),
    "mi355det_stem_bwd_finish": (C.c_int, [vp, vp, vp, vp, i64, vp, vp, vp, vp]),
    "mi355det_bn_finalize": (C.c_int, [vp, i32, i32, i32, i64, vp, vp, f32, f32, vp, vp, vp, vp]),
    "mi355det_bn_eval_scale_shift": (C.c_int, [i32, vp, vp, vp, vp, f32, vp, vp]),
    "mi355det_bn_fold_partials_f64": (C.c_int, [vp, i32, i32, i32, vp, vp]),
    "mi355det_bn_finalize_f64": (C.c_int, [vp, i32, i32, i64, vp, vp, f32, f32, vp, vp, vp, vp]),
    "mi355det_conv_dgrad_bn_rows": (C.c_int, [P(ConvShape)]),
    "mi355det_conv_dgrad_bn": (C.c_int, [P(ConvShape), vp, vp, vp, vp, i32, vp, i32, vp, C.c_float, vp, vp]),
    "mi355det_bn_bwd_sum_partials": (C.c_int, [vp, i32, i32, i32, vp, vp]),
    "mi355det_conv_fwd_ex": (C.c_int, [P(ConvShape), vp, vp, P(ConvEpilogue), vp, C.c_int, i32, vp]),
    "mi355det_retina_loss": (C.c_int, [vp] * 8 + [i32, i64, i32, C.c_float, C.c_float, C.c_float, vp, vp, vp, vp, vp]),
    "mi355det_retina_loss_lv": (C.c_int, [vp] * 8 + [i32, i64, i32, C.c_float, C.c_float, C.c_float, vp, vp, P(LevelGrads), vp, vp]),
    "mi355det_im2col_nchw": (C.c_int, [vp, vp, vp, vp] + [i32] * 8 + [vp]),
    "mi355det_maxpool3x3s2": (C.c_int, [vp, i32, i32, i32, i32, i32, vp, i32, vp]),
    "mi355det_resnet_stem_fwd": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, vp]),
    "mi355det_maxpool3x3s2_bwd": (C.c_int, [vp, i32, vp, i32, i32, i32, i32, i32, vp, i32, vp]),
    "mi355det_relu_affine_bwd": (C.c_int, [vp, i32, vp, i32, vp, i32, vp, i32, i64, C.c_int, vp, i32, vp, i32, vp]),
    "mi355det_upsample_nearest_add": (C.c_int, [vp, i32, i32, i32, i32, i32, vp, i32, i32, i32, vp, i32, vp]),
    "mi355det_upsample_nearest_bwd": (C.c_int, [vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, i32, vp]),
    "mi355det_cast_rows_bf16": (C.c_int, [vp, i64, i64, i32, i64, i32, C.c_float, vp, i32, vp]),
    "mi355det_sgd_step": (C.c_int, [vp, vp, vp, i64] + [C.c_float] * 5 + [C.c_int] * 3 + [vp]),
    "mi355det_adam_step": (C.c_int, [vp, vp, vp, vp, i64] + [C.c_float] * 6 + [i32, C.c_int, vp]),
    "mi355det_sgd_step_guarded": (C.c_int, [vp, vp, vp, i64] + [C.c_float] * 5 + [C.c_int] * 3 + [vp, vp]),
    "mi355det_adam_step_guarded": (C.c_int, [vp, vp, vp, vp, i64] + [C.c_float] * 6 + [i32, C.c_int, vp, vp]),
    "mi355det_grad_nonfinite": (C.c_int, [vp, i64, vp, vp]),
    "mi355det_add_bf16": (C.c_int, [vp, i32, vp, i32, i32, i64, vp, i32, vp]),
    "mi355det_bn_act_fwd": (C.c_int, [vp, i32, vp, i32, i64, f32, vp, i32, vp, i32, vp]),
    "mi355det_bn_act_bwd_reduce": (C.c_int, [vp, i32, vp, i32, vp, i32, vp, i32, i64, f32, vp, vp]),
    "mi355det_bn_act_bwd_reduce_workspace": (sz, [i32, i64]),
    "mi355det_bn_act_bwd_reduce_det": (C.c_int, [vp, i32, vp, i32, vp, i32, vp, i32, i64, f32, vp, vp, sz, vp]),
    "mi355det_bn_act_bwd_apply": (C.c_int, [vp, i32, vp, i32, vp, i32, vp, vp, vp, i32, i64, f32, vp, i32, vp, vp, vp]),
    "mi355det_upsample2x_fwd": (C.c_int, [vp, i32, i32, i32, i32, i32, vp, i32, vp]),
    "mi355det_upsample2x_bwd": (C.c_int, [vp, i32, i32, i32, i32, i32, vp, i32, vp]),
    "mi355det_nchw_f32_to_nhwc": (C.c_int, [vp, i32, i32, i32, i32, vp, C.c_int, i32, vp]),
    "mi355det_nhwc_to_nchw_f32": (C.c_int, [vp, C.c_int, i32, i32, i32, i32, i32, vp, vp]),
    "mi355det_topk_workspace": (sz, [i32]),
    "mi355det_topk_ws": (C.c_int, [vp, i32, i64, i64, i32, C.c_float, vp, vp, vp, vp, sz, vp]),
    "mi355det_resize_bilinear": (C.c_int, [vp, i32, i32, i32, i32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "mi355det_resize_boxes": (C.c_int, [vp, vp, i64, i32, i32, i32, i32, vp]),
    "mi355det_coco_rows": (C.c_int, [vp, i32, vp, vp, i32, i64, f32, f32, f32, i32, i32, vp, vp, vp, vp]),
    "mi355det_fastrcnn_loss_workspace": (sz, [i32]),
    "mi355det_fastrcnn_loss": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, sz, vp]),
}

def _f16_twins():
    """Entry points that exist a second time with fp16 storage (csrc/f16_names.h): same prototype, suffix _f16."""
    import re
    txt = open(os.path.join(HERE, "csrc", "f16_names.h")).read().split("// ---- cross-file symbols")[0]
    return re.findall(r"#define (mi355det_[a-z0-9_]+) \1_f16", txt)


F16_TWINS = _f16_twins()
for _n in F16_TWINS:
    PROTOTYPES[_n + "_f16"] = PROTOTYPES[_n]

_lib = None


class Mi355detError(RuntimeError):
    pass


def lib():
    """Load (once) and return the bound library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Mi355detError(
                f"{LIB_PATH} not found: build it with `python -m object_detectors_amd.build` "
                "(there is no CPU/PyTorch fallback for the hot path)")
        import torch  # noqa: F401  (load torch's bundled HIP runtime first so both sides share ONE libamdhip64)
        L = C.CDLL(LIB_PATH)
        missing = []
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.restype, fn.argtypes = res, args
        if missing:
            raise Mi355detError(f"{LIB_PATH} lacks symbols {missing}: rebuild with `python -m object_detectors_amd.build --force`")
        _lib = L
    return _lib


class _StorageLib:
    """The library as seen by an engine that stores fp16: every entry point that has an fp16 twin resolves to the twin, everything else
    (fp32 statistics, optimizers, the criterion, box kernels) to the one function there is."""

    def __init__(self, L):
        self._L = L
        self._twins = set(F16_TWINS)

    def __getattr__(self, name):
        fn = getattr(self._L, name + "_f16" if name in self._twins else name)
        setattr(self, name, fn)
        return fn


_storage_libs = {}


def storage_lib(storage="bf16"):
    """lib() for the given storage format of activations / gradients / packed weights: "bf16" (default) or "fp16" (the reference's apex-O2
    format, yolo/procedures/initialize.py:44-45)."""
    if storage in ("bf16", None):
        return lib()
    if storage != "fp16":
        raise ValueError("storage must be 'bf16' or 'fp16'")
    if "fp16" not in _storage_libs:
        _storage_libs["fp16"] = _StorageLib(lib())
    return _storage_libs["fp16"]


def check(status, what=""):
    if status != 0:
        msg = lib().mi355det_last_error().decode()
        if status == -1:
            raise ValueError(f"mi355det {what}: {msg}")
        raise Mi355detError(f"mi355det {what} failed ({status}): {msg}")


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())
