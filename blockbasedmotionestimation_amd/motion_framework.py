"""Python mirror of the reference's MF class (motion_framework.h:9-54) over the C-ABI.

    mf = MF(image1, image2, search_size, block_size, num_levels)   # MF::MF, motion_framework.cpp:4-111
    flow = mf.calcMotionBlockMatching()                             # :113-219 -> (H_pad, W_pad, 2) float32
    mf.padded_height, mf.padded_width, mf.padding_x, mf.padding_y   # public fields :16-19

Argument meaning and order follow the reference: arrays are indexed [0] = finest level,
search_size is the window side length.  Errors the reference reports with assert / exit(1)
raise BbmeError here.  All arithmetic runs in the HIP kernels of libbbme.so.
"""
import ctypes as C

import numpy as np

from . import _capi


class MF:
    def __init__(self, image1, image2, search_size, block_size, num_levels=None, device=0,
                 frames_on_device=False):
        if num_levels is None:
            num_levels = len(block_size)
        if num_levels <= 0:
            raise _capi.BbmeError(_capi.ERR_INVALID, "num_levels must be > 0")       # assert :7
        search_size = list(search_size)[:num_levels]
        block_size = list(block_size)[:num_levels]
        self._ctx = C.c_void_p()
        self._lib = _capi.lib()
        self.device = device
        self._torch_frames = None
        if frames_on_device:
            import torch
            if tuple(image1.shape) != tuple(image2.shape):
                raise _capi.BbmeError(_capi.ERR_INVALID, "image1.size() != image2.size()")
            h, w = image1.shape
        else:
            image1 = np.ascontiguousarray(image1, dtype=np.uint8)
            image2 = np.ascontiguousarray(image2, dtype=np.uint8)
            if image1.ndim != 2 or image1.shape != image2.shape:                        # assert :8
                raise _capi.BbmeError(_capi.ERR_INVALID, "image1.size() != image2.size()")
            h, w = image1.shape
        self.orig_height, self.orig_width = h, w
        self.params = _capi.make_params(search_size, block_size)
        _capi.check(self._lib.bbme_create(C.byref(self.params), w, h, device, C.byref(self._ctx)))
        pw, ph, px, py = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _capi.check(self._lib.bbme_get_geometry(self._ctx, C.byref(pw), C.byref(ph), C.byref(px), C.byref(py)))
        self.padded_width, self.padded_height = pw.value, ph.value
        self.padding_x, self.padding_y = px.value, py.value
        self.num_levels = num_levels
        if frames_on_device:
            self.set_frames_device(image1, image2)
        else:
            _capi.check(self._lib.bbme_set_frames_host(self._ctx, image1.ctypes.data, image2.ctypes.data, w))

    # -- lifetime -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.bbme_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- inputs ---------------------------------------------------------------------------
    def set_frames(self, image1, image2):
        """A new pair of the same size into this context (host arrays): what a second MF::MF would do,
        without re-allocating the level state or re-capturing the launch graph."""
        image1 = np.ascontiguousarray(image1, dtype=np.uint8)
        image2 = np.ascontiguousarray(image2, dtype=np.uint8)
        if image1.shape != (self.orig_height, self.orig_width) or image2.shape != image1.shape:
            raise _capi.BbmeError(_capi.ERR_INVALID, "frames must keep the size the context was created for")
        _capi.check(self._lib.bbme_set_frames_host(self._ctx, image1.ctypes.data, image2.ctypes.data, self.orig_width))

    def _check_device_frames(self, image1, image2):
        """The padding kernel reads orig_height x pitch bytes behind each pointer: a tensor of any other shape must be
        refused here (the C-ABI sees only a pointer and a pitch)."""
        import torch
        for t in (image1, image2):
            if not (t.is_cuda and t.dtype == torch.uint8 and t.dim() == 2 and tuple(t.shape) == (self.orig_height, self.orig_width)):
                raise _capi.BbmeError(_capi.ERR_INVALID, "device frames must be 2-D uint8 CUDA tensors of %d x %d (the size the "
                                      "context was created for)" % (self.orig_height, self.orig_width))
        if image1.stride(1) != 1 or image2.stride(1) != 1 or image1.stride(0) != image2.stride(0):
            raise _capi.BbmeError(_capi.ERR_INVALID, "device frames must have unit column stride and a common row pitch")

    def set_frames_device(self, image1, image2):
        """Frames already in HBM (torch uint8 CUDA tensors, H x W): padding + pyramid on the GPU."""
        self._check_device_frames(image1, image2)
        self._torch_frames = (image1, image2)
        # the tensors may still be being written by work on torch's current stream: order the context's stream behind it
        import torch
        _capi.check(self._lib.bbme_wait_for_stream(self._ctx, C.c_void_p(torch.cuda.current_stream(image1.device).cuda_stream)))
        _capi.check(self._lib.bbme_set_frames_device(self._ctx, image1.data_ptr(), image2.data_ptr(),
                                                     image1.stride(0)))

    def set_search_mode(self, raster):
        """False: find_min_block_spiral (the reference's live search); True: the raster find_min_block (:246-294)."""
        _capi.check(self._lib.bbme_set_search_mode(self._ctx, 1 if raster else 0))

    def set_regularizer_mode(self, jacobi):
        """False: the reference's in-place raster sweep, bit for bit.  True: opt-in Jacobi sweeps (not the reference's field)."""
        _capi.check(self._lib.bbme_set_regularizer_mode(self._ctx, 1 if jacobi else 0))

    def set_speculation(self, enabled):
        """Speculative search of the next finer level beside a level's late sweeps (bbme_set_speculation); same result."""
        _capi.check(self._lib.bbme_set_speculation(self._ctx, int(bool(enabled))))

    def set_relaxation(self, enabled):
        """Scheduling only (same field): the relaxation launches in front of the solver on large grids of small blocks.
        Turn them off, like the speculation, when several pairs are in flight on the GPU."""
        _capi.check(self._lib.bbme_set_relaxation(self._ctx, 1 if enabled else 0))

    def set_stream(self, hip_stream_handle):
        _capi.check(self._lib.bbme_set_stream(self._ctx, C.c_void_p(hip_stream_handle)))

    def level_geometry(self, level):
        w, h, b, s = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _capi.check(self._lib.bbme_level_geometry(self._ctx, level, C.byref(w), C.byref(h), C.byref(b), C.byref(s)))
        return w.value, h.value, b.value, s.value

    def set_level_planes(self, level, image1, image2):
        w, h, _, _ = self.level_geometry(level)
        image1 = np.ascontiguousarray(image1, np.uint8)
        image2 = np.ascontiguousarray(image2, np.uint8)
        assert image1.shape == (h, w) and image2.shape == (h, w)
        _capi.check(self._lib.bbme_set_level_planes_host(self._ctx, level, image1.ctypes.data, image2.ctypes.data))

    def get_level_planes(self, level):
        w, h, _, _ = self.level_geometry(level)
        a = np.empty((h, w), np.uint8)
        b = np.empty((h, w), np.uint8)
        _capi.check(self._lib.bbme_get_level_planes_host(self._ctx, level, a.ctypes.data, b.ctypes.data))
        return a, b

    # -- the hot path ---------------------------------------------------------------------
    def estimate_async(self):
        """Enqueue MF::calcMotionBlockMatching on the context's stream; no host wait."""
        _capi.check(self._lib.bbme_estimate(self._ctx))

    def synchronize(self):
        _capi.check(self._lib.bbme_synchronize(self._ctx))

    def get_flow(self, out=None):
        """The dense padded field; `out` may be a preallocated C-contiguous float32 array of that shape, e.g. a view
        of pinned memory (the 66.8 MB of a 4K field download about three times faster into pinned memory)."""
        shape = (self.padded_height, self.padded_width, 2)
        if out is None:
            out = np.empty(shape, np.float32)
        elif out.shape != shape or out.dtype != np.float32 or not out.flags.c_contiguous:
            raise _capi.BbmeError(_capi.ERR_INVALID, "get_flow: out must be a C-contiguous float32 array of shape %s" % (shape,))
        _capi.check(self._lib.bbme_get_flow_host(self._ctx, out.ctypes.data))
        return out

    def get_cells(self, out=None):
        shape = (self.padded_height // 2, self.padded_width // 2, 2)
        if out is None:
            out = np.empty(shape, np.int16)
        elif out.shape != shape or out.dtype != np.int16 or not out.flags.c_contiguous:
            raise _capi.BbmeError(_capi.ERR_INVALID, "get_cells: out must be a C-contiguous int16 array of shape %s" % (shape,))
        _capi.check(self._lib.bbme_get_cells_host(self._ctx, out.ctypes.data))
        return out

    def flow_device_ptr(self):
        p = C.c_void_p()
        _capi.check(self._lib.bbme_flow_device(self._ctx, C.byref(p)))
        return p.value

    def cells_device_ptr(self):
        p = C.c_void_p()
        _capi.check(self._lib.bbme_cells_device(self._ctx, C.byref(p)))
        return p.value

    def expand_cells_device(self, cells_ptr, flow_ptr, hip_stream_handle=None):
        """copy_to_all_pixels for a cell grid anywhere in HBM -> dense padded field (device pointers), on the
        context's stream or on the given HIP stream."""
        _capi.check(self._lib.bbme_expand_cells_device_on(self._ctx, C.c_void_p(cells_ptr), C.c_void_p(flow_ptr),
                                                          C.c_void_p(hip_stream_handle or 0)))

    def calculate_mse_device(self, gtruth, scale=4):
        """Flow::CalculateMSE against a ground-truth field already in HBM (torch float32 CUDA tensor (h, w, 2)),
        fused with the driver's every-`scale`-th-pixel / `scale` subsampling; nothing is downloaded but the sums."""
        assert gtruth.is_cuda and gtruth.is_contiguous() and gtruth.dim() == 3 and gtruth.shape[2] == 2
        assert gtruth.dtype.itemsize == 4 and gtruth.dtype.is_floating_point
        out = C.c_double()
        _capi.check(self._lib.bbme_calculate_mse_device(self._ctx, C.c_void_p(gtruth.data_ptr()), gtruth.shape[1],
                                                        gtruth.shape[0], int(scale), C.byref(out)))
        return out.value

    def calcMotionBlockMatching(self):
        """cv::Mat MF::calcMotionBlockMatching() -- dense padded (H, W, 2) float32 (u, v) field."""
        self.estimate_async()
        return self.get_flow()

    # -- the reference's private methods, one stage at a time (parity tests) --------------
    def stage_search(self, level):
        """copyMVs() + calcLevelBM() of one level."""
        _capi.check(self._lib.bbme_stage_search(self._ctx, level))

    def stage_regularize(self, level, block, lambda_multiplier):
        """One regularize_MVs() sweep (divide_blocks() first when the grid is at 2*block)."""
        _capi.check(self._lib.bbme_stage_regularize(self._ctx, level, block, lambda_multiplier))

    def stage_get_mvs(self, level, block):
        w, h, _, _ = self.level_geometry(level)
        out = np.empty((h // block, w // block, 2), np.int16)
        _capi.check(self._lib.bbme_stage_get_mvs(self._ctx, level, block, out.ctypes.data))
        return out

    def stage_set_mvs(self, level, block, mvs):
        w, h, _, _ = self.level_geometry(level)
        mvs = np.ascontiguousarray(mvs, np.int16)
        assert mvs.shape == (h // block, w // block, 2)
        _capi.check(self._lib.bbme_stage_set_mvs(self._ctx, level, block, mvs.ctypes.data))

    def stage_expand(self):
        _capi.check(self._lib.bbme_stage_expand(self._ctx))

    def last_sweep_passes(self):
        v = (C.c_int * 2)()
        _capi.check(self._lib.bbme_last_sweep_passes(self._ctx, v))
        return v[0], v[1]

    def sweep_stats(self):
        v = (C.c_uint * 16)()
        _capi.check(self._lib.bbme_sweep_stats(self._ctx, v))
        return list(v)

    def set_profiling(self, enabled):
        _capi.check(self._lib.bbme_set_profiling(self._ctx, int(enabled)))

    def timings(self):
        v = [C.c_float() for _ in range(5)]
        _capi.check(self._lib.bbme_get_timings(self._ctx, *[C.byref(x) for x in v]))
        return dict(zip(("total_ms", "search_ms", "regularize_ms", "expand_ms", "search_level0_ms"),
                        [x.value for x in v]))


class MFBatch(MF):
    """Several independent frame pairs of one size behind ONE launch sequence (bbme_create_batch): the pairs of a sequence
    that share a GPU.  Every kernel of the estimate works on all pairs at once; each pair's field is bit for bit what an MF
    of its own returns.  `pairs` = [(image1, image2), ...] host arrays, or torch uint8 CUDA tensors with
    frames_on_device=True.  Methods inherited from MF without a pair index address pair 0."""

    def __init__(self, pairs, search_size, block_size, num_levels=None, device=0, frames_on_device=False):
        if num_levels is None:
            num_levels = len(block_size)
        if num_levels <= 0 or not pairs:
            raise _capi.BbmeError(_capi.ERR_INVALID, "num_levels must be > 0 and pairs non-empty")
        self._ctx = C.c_void_p()
        self._lib = _capi.lib()
        self.device = device
        self.batch = len(pairs)
        self._torch_frames = [None] * self.batch
        h, w = pairs[0][0].shape
        self.orig_height, self.orig_width = h, w
        self.params = _capi.make_params(list(search_size)[:num_levels], list(block_size)[:num_levels])
        _capi.check(self._lib.bbme_create_batch(C.byref(self.params), w, h, device, self.batch, C.byref(self._ctx)))
        pw, ph, px, py = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _capi.check(self._lib.bbme_get_geometry(self._ctx, C.byref(pw), C.byref(ph), C.byref(px), C.byref(py)))
        self.padded_width, self.padded_height = pw.value, ph.value
        self.padding_x, self.padding_y = px.value, py.value
        self.num_levels = num_levels
        for p, (image1, image2) in enumerate(pairs):
            if frames_on_device:
                self.set_pair_device(p, image1, image2)
            else:
                self.set_pair(p, image1, image2)

    def set_pair(self, pair, image1, image2):
        image1 = np.ascontiguousarray(image1, dtype=np.uint8)
        image2 = np.ascontiguousarray(image2, dtype=np.uint8)
        if image1.shape != (self.orig_height, self.orig_width) or image2.shape != image1.shape:
            raise _capi.BbmeError(_capi.ERR_INVALID, "frames must keep the size the context was created for")
        _capi.check(self._lib.bbme_set_frames_host_pair(self._ctx, pair, image1.ctypes.data, image2.ctypes.data, self.orig_width))

    def set_frames(self, image1, image2):
        """Pair 0 (the inherited entry point without a pair index)."""
        self.set_pair(0, image1, image2)

    def set_frames_device(self, image1, image2):
        """Pair 0 (the inherited entry point without a pair index)."""
        self.set_pair_device(0, image1, image2)

    def set_pair_device(self, pair, image1, image2):
        if not 0 <= pair < self.batch:
            raise _capi.BbmeError(_capi.ERR_INVALID, "pair %d of a batch of %d" % (pair, self.batch))
        self._check_device_frames(image1, image2)
        self._torch_frames[pair] = (image1, image2)
        import torch
        _capi.check(self._lib.bbme_wait_for_stream(self._ctx, C.c_void_p(torch.cuda.current_stream(image1.device).cuda_stream)))
        _capi.check(self._lib.bbme_set_frames_device_pair(self._ctx, pair, image1.data_ptr(), image2.data_ptr(), image1.stride(0)))

    def get_pair_flow(self, pair, out=None):
        shape = (self.padded_height, self.padded_width, 2)
        if out is None:
            out = np.empty(shape, np.float32)
        elif out.shape != shape or out.dtype != np.float32 or not out.flags.c_contiguous:
            raise _capi.BbmeError(_capi.ERR_INVALID, "get_pair_flow: out must be a C-contiguous float32 array of shape %s" % (shape,))
        _capi.check(self._lib.bbme_get_flow_host_pair(self._ctx, pair, out.ctypes.data))
        return out

    def get_pair_cells(self, pair, out=None):
        shape = (self.padded_height // 2, self.padded_width // 2, 2)
        if out is None:
            out = np.empty(shape, np.int16)
        elif out.shape != shape or out.dtype != np.int16 or not out.flags.c_contiguous:
            raise _capi.BbmeError(_capi.ERR_INVALID, "get_pair_cells: out must be a C-contiguous int16 array of shape %s" % (shape,))
        _capi.check(self._lib.bbme_get_cells_host_pair(self._ctx, pair, out.ctypes.data))
        return out

    def calcMotionBlockMatching(self):
        """Every pair's dense padded field, in order."""
        self.estimate_async()
        return [self.get_pair_flow(p) for p in range(self.batch)]


def plan_padding(width, height, search_size, block_size):
    """padded_width, padded_height, padding_x, padding_y of MF::MF (motion_framework.cpp:14-54)."""
    p = _capi.make_params(search_size, block_size)
    v = [C.c_int() for _ in range(4)]
    _capi.check(_capi.lib().bbme_plan_padding(width, height, C.byref(p), *[C.byref(x) for x in v]))
    return tuple(x.value for x in v)


def pad_zero(img, pad_x, pad_y):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.empty((h + 2 * pad_y, w + 2 * pad_x), np.uint8)
    _capi.check(_capi.lib().bbme_pad_zero_host(img.ctypes.data, w, h, w, pad_x, pad_y, out.ctypes.data))
    return out


def pyr_down(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.empty((h // 2, w // 2), np.uint8)
    _capi.check(_capi.lib().bbme_pyr_down_host(img.ctypes.data, w, h, out.ctypes.data))
    return out


def resize_x4(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.empty((h * 4, w * 4), np.uint8)
    _capi.check(_capi.lib().bbme_resize_x4_host(img.ctypes.data, w, h, out.ctypes.data))
    return out
