"""ctypes front-end of the CPU oracle (oracle/bbme_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never by the product package.  Parity status of
each piece is stated in oracle/bbme_oracle.h (hot path: PARITY UNPINNED; .flo
codec and EPE: pinned by the reference's vendored flowIO.cpp and GT files).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libbbme_oracle.so")
FLO_REF = os.path.join(_HERE, "_ref", "flo_ref")


class _Level(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("block_size", C.c_int),
                ("search_size", C.c_int), ("lambda_", C.c_float),
                ("image1", C.POINTER(C.c_uint8)), ("image2", C.POINTER(C.c_uint8)),
                ("flow", C.POINTER(C.c_float)), ("cache", C.POINTER(C.c_int32))]


class _MF(C.Structure):
    _fields_ = [("num_levels", C.c_int), ("lv", C.POINTER(_Level)),
                ("lambda_multiplier", C.c_int),
                ("padded_height", C.c_int), ("padded_width", C.c_int),
                ("padding_x", C.c_int), ("padding_y", C.c_int),
                ("orig_height", C.c_int), ("orig_width", C.c_int), ("use_cache", C.c_int),
                ("raster_search", C.c_int), ("jacobi_regularizer", C.c_int)]


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "bbme_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        P = C.POINTER
        L.orc_plan_padding.argtypes = [C.c_int, C.c_int, P(C.c_int), C.c_int] + [P(C.c_int)] * 4
        L.orc_pad_zero.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_pad_zero.restype = None
        L.orc_pyr_down.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_pyr_down.restype = None
        L.orc_resize_linear_x4.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_resize_linear_x4.restype = None
        L.orc_mf_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    P(C.c_int), P(C.c_int), C.c_int, C.c_int, P(P(_MF))]
        L.orc_mf_create_from_planes.argtypes = [P(C.c_void_p), P(C.c_void_p), P(C.c_int), P(C.c_int),
                                                P(C.c_int), P(C.c_int), C.c_int, C.c_int, P(P(_MF))]
        L.orc_mf_destroy.argtypes = [P(_MF)]
        L.orc_mf_destroy.restype = None
        for name in ("orc_copy_mvs", "orc_calc_level_bm", "orc_regularize_mvs", "orc_divide_blocks",
                     "orc_copy_to_all_pixels", "orc_level_schedule"):
            f = getattr(L, name)
            f.argtypes = [P(_MF), C.c_int]
            f.restype = None
        L.orc_calc_motion_block_matching.argtypes = [P(_MF)]
        L.orc_calc_motion_block_matching.restype = P(C.c_float)
        L.orc_find_min_block_spiral.argtypes = [P(_MF)] + [C.c_int] * 5 + [P(C.c_int)] * 2
        L.orc_find_min_block_spiral.restype = None
        L.orc_spiral_walk.argtypes = [C.c_int, P(C.c_int), P(C.c_int), C.c_int]
        L.orc_regularize_fixpoint.argtypes = [P(_MF), C.c_int, P(C.c_int), C.c_int]
        L.orc_regularize_fixpoint.restype = C.c_int
        L.orc_flo_read.argtypes = [C.c_char_p, P(C.c_int), P(C.c_int), P(P(C.c_float))]
        L.orc_flo_write.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_calculate_mse.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_calculate_mse.restype = C.c_double
        L.orc_subsample_div4.argtypes = [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_int]
        L.orc_subsample_div4.restype = None
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_free.restype = None
        _lib = L
    return _lib


def _iarr(v):
    return (C.c_int * len(v))(*[int(x) for x in v])


def plan_padding(width, height, block_size):
    pw, ph, px, py = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = lib().orc_plan_padding(width, height, _iarr(block_size), len(block_size),
                                C.byref(pw), C.byref(ph), C.byref(px), C.byref(py))
    return rc, pw.value, ph.value, px.value, py.value


def pad_zero(img, pad_x, pad_y):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty((h + 2 * pad_y, w + 2 * pad_x), np.uint8)
    lib().orc_pad_zero(img.ctypes.data, w, h, w, pad_x, pad_y, out.ctypes.data)
    return out


def pyr_down(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty((h // 2, w // 2), np.uint8)
    lib().orc_pyr_down(img.ctypes.data, w, h, out.ctypes.data)
    return out


def resize_linear_x4(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty((h * 4, w * 4), np.uint8)
    lib().orc_resize_linear_x4(img.ctypes.data, w, h, out.ctypes.data)
    return out


def spiral_walk(shift):
    n = lib().orc_spiral_walk(shift, None, None, 0)
    dx, dy = (C.c_int * n)(), (C.c_int * n)()
    lib().orc_spiral_walk(shift, dx, dy, n)
    return np.array(dx[:]), np.array(dy[:])


class OracleMF:
    """The reference's MF class, stage by stage (motion_framework.h:9-54)."""

    def __init__(self, image1=None, image2=None, search_size=None, block_size=None,
                 planes1=None, planes2=None, use_cache=True):
        L = lib()
        self._p = C.POINTER(_MF)()
        n = len(block_size)
        if planes1 is not None:
            self._keep = [np.ascontiguousarray(p, np.uint8) for p in list(planes1) + list(planes2)]
            a1 = (C.c_void_p * n)(*[p.ctypes.data for p in self._keep[:n]])
            a2 = (C.c_void_p * n)(*[p.ctypes.data for p in self._keep[n:]])
            ws = _iarr([p.shape[1] for p in self._keep[:n]])
            hs = _iarr([p.shape[0] for p in self._keep[:n]])
            rc = L.orc_mf_create_from_planes(a1, a2, ws, hs, _iarr(search_size), _iarr(block_size),
                                             n, int(use_cache), C.byref(self._p))
        else:
            i1 = np.ascontiguousarray(image1, np.uint8)
            i2 = np.ascontiguousarray(image2, np.uint8)
            assert i1.shape == i2.shape
            h, w = i1.shape
            rc = L.orc_mf_create(i1.ctypes.data, i2.ctypes.data, w, h, w, _iarr(search_size),
                                 _iarr(block_size), n, int(use_cache), C.byref(self._p))
        if rc != 0:
            raise ValueError("orc_mf_create failed: %d" % rc)
        self.num_levels = n
        m = self._p.contents
        self.padded_height, self.padded_width = m.padded_height, m.padded_width
        self.padding_x, self.padding_y = m.padding_x, m.padding_y

    def set_jacobi_regularizer(self, flag):
        """NOT the reference: sweeps read the field as the previous sweep left it (checker of the product's fast mode)."""
        self._p.contents.jacobi_regularizer = int(bool(flag))

    def set_raster_search(self, flag):
        """calcLevelBM calls find_min_block (:235) instead of find_min_block_spiral (:236)."""
        self._p.contents.raster_search = int(bool(flag))

    def close(self):
        if self._p:
            lib().orc_mf_destroy(self._p)
            self._p = C.POINTER(_MF)()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _lv(self, level):
        return self._p.contents.lv[level]

    def level_shape(self, level):
        lv = self._lv(level)
        return lv.height, lv.width

    def block_size(self, level):
        return self._lv(level).block_size

    def set_block_size(self, level, bs):
        self._p.contents.lv[level].block_size = bs

    def lambda_(self, level):
        return self._lv(level).lambda_

    def set_lambda(self, level, v):
        self._p.contents.lv[level].lambda_ = v

    def set_lambda_multiplier(self, m):
        self._p.contents.lambda_multiplier = m

    def image(self, level, which):
        lv = self._lv(level)
        ptr = lv.image1 if which == 1 else lv.image2
        return np.ctypeslib.as_array(ptr, shape=(lv.height, lv.width))

    def flow(self, level):
        """Dense CV_32FC2 level_flow as a (H, W, 2) float32 view (not a copy)."""
        lv = self._lv(level)
        return np.ctypeslib.as_array(lv.flow, shape=(lv.height, lv.width, 2))

    def block_mvs(self, level, bs):
        """MVs at the origins of bs x bs blocks, as int32 (rows, cols, 2)."""
        f = self.flow(level)[::bs, ::bs, :]
        i = f.astype(np.int32)
        assert np.array_equal(i.astype(np.float32), f), "non-integer MV in oracle"
        return i

    def copy_mvs(self, level):
        lib().orc_copy_mvs(self._p, level)

    def calc_level_bm(self, level):
        lib().orc_calc_level_bm(self._p, level)

    def regularize_mvs(self, level, lambda_multiplier):
        self.set_lambda_multiplier(lambda_multiplier)
        lib().orc_regularize_mvs(self._p, level)

    def regularize_fixpoint(self, level, lambda_multiplier, max_stats=64):
        """CPU model of the GPU schedule (Jacobi pass + dirty fix-up passes)."""
        self.set_lambda_multiplier(lambda_multiplier)
        stats = (C.c_int * max_stats)()
        n = lib().orc_regularize_fixpoint(self._p, level, stats, max_stats)
        return n, list(stats[:min(n, max_stats)])

    def divide_blocks(self, level):
        lib().orc_divide_blocks(self._p, level)

    def copy_to_all_pixels(self, level):
        lib().orc_copy_to_all_pixels(self._p, level)

    def level_schedule(self, level):
        lib().orc_level_schedule(self._p, level)

    def find_min_block_spiral(self, level, y1, x1, y2, x2):
        px, py = C.c_int(), C.c_int()
        lib().orc_find_min_block_spiral(self._p, level, y1, x1, y2, x2, C.byref(px), C.byref(py))
        return px.value, py.value

    def calc_motion_block_matching(self):
        lib().orc_calc_motion_block_matching(self._p)
        return self.flow(0).copy()


def flo_read(path):
    w, h = C.c_int(), C.c_int()
    data = C.POINTER(C.c_float)()
    rc = lib().orc_flo_read(os.fsencode(path), C.byref(w), C.byref(h), C.byref(data))
    if rc != 0:
        raise IOError("orc_flo_read(%s) failed: %d" % (path, rc))
    out = np.ctypeslib.as_array(data, shape=(h.value, w.value, 2)).copy()
    lib().orc_free(data)
    return out


def flo_write(path, flow):
    flow = np.ascontiguousarray(flow, np.float32)
    h, w, _ = flow.shape
    rc = lib().orc_flo_write(os.fsencode(path), w, h, flow.ctypes.data)
    if rc != 0:
        raise IOError("orc_flo_write(%s) failed: %d" % (path, rc))


def calculate_mse(gtruth, flow):
    g = np.ascontiguousarray(gtruth, np.float32)
    f = np.ascontiguousarray(flow, np.float32)
    assert g.shape == f.shape
    return lib().orc_calculate_mse(g.ctypes.data, f.ctypes.data, g.shape[1], g.shape[0])


def motion_to_color(flow, maxmotion=-1.0, vendored=False):
    """Flow::MotionToColor -> ((H, W, 3) uint8 B,G,R, (max radius, min u, max u, min v, max v)).
    vendored=True: expression types of the vendored Middlebury colorcode.cpp (pinning only)."""
    f = np.ascontiguousarray(flow, np.float32)
    out = np.empty((f.shape[0], f.shape[1], 3), np.uint8)
    rng = (C.c_float * 5)()
    fn = lib().orc_motion_to_color_flavour
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.POINTER(C.c_float), C.c_int]
    fn(f.ctypes.data, f.shape[1], f.shape[0], float(maxmotion), out.ctypes.data, rng, int(bool(vendored)))
    return out, tuple(rng)


def subsample_div4(flow_padded, pad_x, pad_y, out_width, out_height):
    f = np.ascontiguousarray(flow_padded, np.float32)
    out = np.zeros((out_height, out_width, 2), np.float32)
    lib().orc_subsample_div4(f.ctypes.data, f.shape[1], f.shape[0], pad_x, pad_y,
                             out.ctypes.data, out_width, out_height)
    return out
