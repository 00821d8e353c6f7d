"""ctypes loader of the C / OpenMP form of the CPU oracle (oracle/c/box_oracle.c) - TEST INFRASTRUCTURE ONLY.
Built by oracle/c/build.sh (`__graft_entry__.build()` runs it); `available()` tells tests / bench.py whether the library is there."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libbox_oracle.so")
_LIB = None
F32, I64 = np.float32, np.int64


def build():
    subprocess.run(["sh", os.path.join(_HERE, "c", "build.sh")], check=True, stdout=subprocess.DEVNULL)


def available():
    return os.path.exists(_SO)


def _lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(_SO)
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int64)
        L.oracle_box_iou.argtypes = [fp, C.c_int64, fp, C.c_int64, fp]
        L.oracle_matcher.argtypes = [fp, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_int, ip]
        L.oracle_nms.argtypes = [fp, fp, C.c_int64, C.c_float, ip]
        L.oracle_nms.restype = C.c_int64
        L.oracle_batched_nms.argtypes = [fp, fp, ip, C.c_int64, C.c_float, ip]
        L.oracle_batched_nms.restype = C.c_int64
        L.oracle_encode_boxes.argtypes = [fp, fp, C.c_int64, fp, fp]
        L.oracle_decode_boxes.argtypes = [fp, fp, C.c_int64, fp, C.c_float, fp]
        L.oracle_sigmoid_focal_loss_sum.argtypes = [fp, fp, C.c_int64, C.c_float, C.c_float, fp]
        L.oracle_sigmoid_focal_loss_sum.restype = C.c_double
        _LIB = L
    return _LIB


def _f(a):
    a = np.ascontiguousarray(a, F32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    a = np.ascontiguousarray(a, I64)
    return a, a.ctypes.data_as(C.POINTER(C.c_int64))


def box_iou(a, b):
    a, pa = _f(np.asarray(a, F32).reshape(-1, 4))
    b, pb = _f(np.asarray(b, F32).reshape(-1, 4))
    out = np.empty((a.shape[0], b.shape[0]), F32)
    _lib().oracle_box_iou(pa, a.shape[0], pb, b.shape[0], out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def matcher(q, high, low, allow_low_quality):
    q, pq = _f(q)
    if q.size == 0:
        raise ValueError("No ground-truth boxes available for one of the images during training" if q.shape[0] == 0 else
                         "No proposal boxes available for one of the images during training")
    out = np.empty(q.shape[1], I64)
    _lib().oracle_matcher(pq, q.shape[0], q.shape[1], float(high), float(low), int(bool(allow_low_quality)), out.ctypes.data_as(C.POINTER(C.c_int64)))
    return out


def nms(boxes, scores, thr):
    b, pb = _f(np.asarray(boxes, F32).reshape(-1, 4))
    s, ps = _f(scores)
    keep = np.empty(max(b.shape[0], 1), I64)
    k = _lib().oracle_nms(pb, ps, b.shape[0], float(thr), keep.ctypes.data_as(C.POINTER(C.c_int64)))
    return keep[:k].copy()


def batched_nms(boxes, scores, idxs, thr):
    b, pb = _f(np.asarray(boxes, F32).reshape(-1, 4))
    s, ps = _f(scores)
    ix, pi = _i(idxs)
    keep = np.empty(max(b.shape[0], 1), I64)
    k = _lib().oracle_batched_nms(pb, ps, pi, b.shape[0], float(thr), keep.ctypes.data_as(C.POINTER(C.c_int64)))
    return keep[:k].copy()


def encode_boxes(ref, prop, weights):
    r, pr = _f(ref)
    p, pp = _f(prop)
    w, pw = _f(weights)
    out = np.empty_like(r)
    _lib().oracle_encode_boxes(pr, pp, r.shape[0], pw, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def decode_boxes(codes, boxes, weights, clip=float(np.log(1000.0 / 16))):
    c, pc = _f(codes)
    b, pb = _f(boxes)
    w, pw = _f(weights)
    out = np.empty_like(c)
    _lib().oracle_decode_boxes(pc, pb, c.shape[0], pw, float(clip), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def sigmoid_focal_loss_sum(x, t, alpha=0.25, gamma=2.0, want_grad=True):
    x, px = _f(np.asarray(x, F32).reshape(-1))
    t, pt = _f(np.asarray(t, F32).reshape(-1))
    grad = np.empty_like(x) if want_grad else None
    total = _lib().oracle_sigmoid_focal_loss_sum(px, pt, x.shape[0], float(alpha), float(gamma),
                                                 grad.ctypes.data_as(C.POINTER(C.c_float)) if want_grad else None)
    return total, grad
