"""ctypes binding of the C oracle (oracle_match.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; the product never does.

numpy in, numpy out; every function cites the reference lines it follows in
oracle_match.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_match.so")

OK, ERR_LABEL_RANGE, ERR_DEGENERATE_BOX, ERR_INVALID_COST, ERR_INFEASIBLE, ERR_NOMEM = range(6)

#: exception type the reference raises for each oracle status
#: (IndexError: matcher.py:73,81; AssertionError: box_ops.py:59-60;
#: ValueError: scipy's linear_sum_assignment called at matcher.py:94)
ERROR_TYPES = {
    ERR_LABEL_RANGE: IndexError,
    ERR_DEGENERATE_BOX: AssertionError,
    ERR_INVALID_COST: ValueError,
    ERR_INFEASIBLE: ValueError,
    ERR_NOMEM: MemoryError,
}


def build(force=False):
    """Compile oracle_match.c with gcc (no-op when the .so is up to date)."""
    src = os.path.join(_HERE, "oracle_match.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int64)
        L.orc_cost_block_f32.restype = ctypes.c_int
        L.orc_cost_block_f32.argtypes = [fp, fp, ip, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                         ctypes.c_float, fp]
        L.orc_lsap_f32.restype = ctypes.c_int
        L.orc_lsap_f32.argtypes = [fp, ctypes.c_int, ctypes.c_int, ip, ip]
        L.orc_match_f32.restype = ctypes.c_int
        L.orc_match_f32.argtypes = [fp, fp, ip, fp, ip, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                    fp, ip, ip]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


def _raise(rc):
    if rc != OK:
        raise ERROR_TYPES[rc]("oracle status %d" % rc)


def cost_block(logits, boxes, labels, tboxes, cls_loss="ce", cost_class=1.0, cost_bbox=1.0, cost_giou=1.0,
               focal_alpha=0.25, focal_gamma=2.0):
    """[Q, T] fp32 cost block of one image (matcher.py:59-90 on one diagonal block)."""
    logits, boxes, labels, tboxes = _f32(logits), _f32(boxes), _i64(labels), _f32(tboxes)
    Q, C = logits.shape
    T = labels.shape[0]
    out = np.empty((Q, T), dtype=np.float32)
    rc = lib().orc_cost_block_f32(_fp(logits), _fp(boxes), _ip(labels), _fp(tboxes), Q, T, C,
                                  0 if cls_loss == "ce" else 1, cost_class, cost_bbox, cost_giou,
                                  focal_alpha, focal_gamma, _fp(out))
    _raise(rc)
    return out


def lsap(cost):
    """(row_ind, col_ind) of scipy.optimize.linear_sum_assignment for an fp32 [nr, nc] matrix."""
    cost = _f32(cost)
    nr, nc = cost.shape
    n = min(nr, nc)
    a = np.empty(n, dtype=np.int64)
    b = np.empty(n, dtype=np.int64)
    rc = lib().orc_lsap_f32(_fp(cost), nr, nc, _ip(a), _ip(b))
    _raise(rc)
    return a, b


def match(logits, boxes, labels_list, tboxes_list, cls_loss="ce", cost_class=1.0, cost_bbox=1.0, cost_giou=1.0,
          focal_alpha=0.25, focal_gamma=2.0, return_cost=False):
    """One matcher call (matcher.py:59-95): list of (src, tgt) int64 arrays per image.

    logits [B, Q, C], boxes [B, Q, 4]; labels_list/tboxes_list hold one array per image.
    """
    logits, boxes = _f32(logits), _f32(boxes)
    B, Q, C = logits.shape
    sizes = [int(len(l)) for l in labels_list]
    off = np.zeros(B + 1, dtype=np.int64)
    off[1:] = np.cumsum(sizes)
    M = int(off[-1])
    labels = _i64(np.concatenate([np.asarray(l, dtype=np.int64).reshape(-1) for l in labels_list])) if B else _i64([])
    tboxes = _f32(np.concatenate([np.asarray(t, dtype=np.float32).reshape(-1, 4) for t in tboxes_list])) \
        if B else _f32(np.zeros((0, 4)))
    npairs = [min(Q, t) for t in sizes]
    src = np.empty(max(sum(npairs), 1), dtype=np.int64)
    tgt = np.empty(max(sum(npairs), 1), dtype=np.int64)
    cost = np.empty(max(Q * M, 1), dtype=np.float32) if return_cost else None
    rc = lib().orc_match_f32(_fp(logits), _fp(boxes), _ip(labels), _fp(tboxes), _ip(off), B, Q, C,
                             0 if cls_loss == "ce" else 1, cost_class, cost_bbox, cost_giou,
                             focal_alpha, focal_gamma,
                             _fp(cost) if return_cost else ctypes.cast(None, ctypes.POINTER(ctypes.c_float)),
                             _ip(src), _ip(tgt))
    _raise(rc)
    out, pos = [], 0
    for n in npairs:
        out.append((src[pos:pos + n].copy(), tgt[pos:pos + n].copy()))
        pos += n
    if return_cost:
        blocks, cpos = [], 0
        for t in sizes:
            blocks.append(cost[cpos:cpos + Q * t].reshape(Q, t).copy())
            cpos += Q * t
        return out, blocks
    return out
