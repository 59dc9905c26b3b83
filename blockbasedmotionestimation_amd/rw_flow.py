"""Python mirror of the reference's Flow class (rw_flow.h:9-38) over the C-ABI.

ReadFlowFile / WriteFlowFile / MotionToColor / CalculateMSE keep the reference's names and argument
order.  ShowImage (rw_flow.cpp:334-340) opens a GUI window and writes flowimg.png in the reference;
here it only writes the image, as binary PPM (there is neither a GUI nor a PNG codec in this image).
Errors the reference reports with a message and exit(1) raise BbmeError instead.
"""
import ctypes as C
import os

import numpy as np

from . import _capi


class Flow:
    def ReadFlowFile(self, filename):
        """Flow::ReadFlowFile (rw_flow.cpp:50-136) -> (H, W, 2) float32."""
        w, h = C.c_int(), C.c_int()
        data = C.POINTER(C.c_float)()
        fn = None if filename is None else os.fsencode(filename)
        _capi.check(_capi.lib().bbme_flo_read(fn, C.byref(w), C.byref(h), C.byref(data)))
        try:
            return np.ctypeslib.as_array(data, shape=(h.value, w.value, 2)).copy()
        finally:
            _capi.lib().bbme_free(data)

    def WriteFlowFile(self, img, filename):
        """Flow::WriteFlowFile (rw_flow.cpp:139-200)."""
        img = np.ascontiguousarray(img, np.float32)
        if img.ndim != 3 or img.shape[2] != 2:
            raise _capi.BbmeError(_capi.ERR_INVALID, "WriteFlowFile: image must have 2 bands")
        fn = None if filename is None else os.fsencode(filename)
        _capi.check(_capi.lib().bbme_flo_write(fn, img.shape[1], img.shape[0], img.ctypes.data))

    def CalculateMSE(self, gtruth, flow):
        """Flow::CalculateMSE (rw_flow.cpp:309-332): average end-point error over known GT pixels."""
        g = np.ascontiguousarray(gtruth, np.float32)
        f = np.ascontiguousarray(flow, np.float32)
        if g.shape != f.shape:
            raise _capi.BbmeError(_capi.ERR_INVALID, "CalculateMSE: shapes differ")
        out = C.c_double()
        _capi.check(_capi.lib().bbme_calculate_mse(g.ctypes.data, f.ctypes.data, g.shape[1], g.shape[0], C.byref(out)))
        return out.value


    def MotionToColor(self, input_img, maxmotion=-1.0, verbose=True):
        """Flow::MotionToColor (rw_flow.cpp:202-249): Middlebury colour coding -> (H, W, 3) uint8, B,G,R
        as in the reference's CV_8UC3 output.  Prints the reference's "max motion" line unless verbose=False."""
        f = np.ascontiguousarray(input_img, np.float32)
        if f.ndim != 3 or f.shape[2] != 2:
            raise _capi.BbmeError(_capi.ERR_INVALID, "MotionToColor: image must have 2 bands")
        out = np.empty((f.shape[0], f.shape[1], 3), np.uint8)
        rng = (C.c_float * 5)()
        _capi.check(_capi.lib().bbme_motion_to_color(f.ctypes.data, f.shape[1], f.shape[0], float(maxmotion),
                                                     out.ctypes.data, rng))
        self.last_range = tuple(rng)
        if verbose:
            print("max motion: %.4f  motion range: u = %.3f .. %.3f;  v = %.3f .. %.3f" % self.last_range)
        return out

    def ShowImage(self, flow_img, filename="flowimg.ppm"):
        """Flow::ShowImage (rw_flow.cpp:334-340) without the window: writes the B,G,R image as binary PPM."""
        img = np.ascontiguousarray(flow_img, np.uint8)
        if img.ndim != 3 or img.shape[2] != 3:
            raise _capi.BbmeError(_capi.ERR_INVALID, "ShowImage: image must have 3 bands")
        _capi.check(_capi.lib().bbme_ppm_write_bgr(os.fsencode(filename), img.shape[1], img.shape[0], img.ctypes.data))


def subsample_div4(flow_padded, pad_x, pad_y, out_width, out_height):
    """main_class.cpp:58-70: strip padding, every 4th pixel, divide by 4."""
    f = np.ascontiguousarray(flow_padded, np.float32)
    out = np.zeros((out_height, out_width, 2), np.float32)
    _capi.check(_capi.lib().bbme_subsample_div4(f.ctypes.data, f.shape[1], f.shape[0], pad_x, pad_y,
                                                out.ctypes.data, out_width, out_height))
    return out


class FlowWriter:
    """Flow::WriteFlowFile on a worker thread (bbme_flo_writer_*): submit() returns at once, the file is written while
    the next pair is being estimated.  The array handed to submit() must stay alive and untouched until a wait covers it.
    workers > 1: a pool, one file per worker at a time (files then finish in any order); submit*() return the job's ticket
    and wait(ticket) returns once every job up to it is on disk."""

    def __init__(self, workers=1):
        self._w = C.c_void_p()
        _capi.check(_capi.lib().bbme_flo_writer_create_pool(int(workers), C.byref(self._w)))
        self._keep = []

    def _ticket(self):
        t = C.c_ulonglong()
        _capi.check(_capi.lib().bbme_flo_writer_ticket(self._w, C.byref(t)))
        return t.value

    def submit(self, flow_padded, filename, pad_x=0, pad_y=0, width=None, height=None):
        """The (height, width) window at (pad_y, pad_x) of a C-contiguous float32 (H, W, 2) field (default: all of it)."""
        f = flow_padded
        if f.dtype != np.float32 or f.ndim != 3 or f.shape[2] != 2 or not f.flags.c_contiguous:
            raise _capi.BbmeError(_capi.ERR_INVALID, "FlowWriter.submit: C-contiguous float32 (H, W, 2) field expected")
        width = f.shape[1] - pad_x if width is None else width
        height = f.shape[0] - pad_y if height is None else height
        if pad_x < 0 or pad_y < 0 or pad_x + width > f.shape[1] or pad_y + height > f.shape[0]:
            raise _capi.BbmeError(_capi.ERR_INVALID, "FlowWriter.submit: window outside the field")
        self._keep.append(f)
        ptr = f.ctypes.data + 8 * (pad_y * f.shape[1] + pad_x)
        _capi.check(_capi.lib().bbme_flo_writer_submit(self._w, os.fsencode(filename), width, height, C.c_void_p(ptr), f.shape[1]))
        return self._ticket()

    def submit_cells(self, cells, filename, pad_x=0, pad_y=0, width=None, height=None):
        """The same file from the compact result of MF.get_cells (int16 (rows, cols, 2), one (dx, dy) per 2x2 pixels of the
        padded field): expansion, padding strip and write happen on the worker (bbme_flo_writer_submit_cells)."""
        c = cells
        if c.dtype != np.int16 or c.ndim != 3 or c.shape[2] != 2 or not c.flags.c_contiguous:
            raise _capi.BbmeError(_capi.ERR_INVALID, "FlowWriter.submit_cells: C-contiguous int16 (rows, cols, 2) grid expected")
        width = 2 * c.shape[1] - pad_x if width is None else width
        height = 2 * c.shape[0] - pad_y if height is None else height
        self._keep.append(c)
        _capi.check(_capi.lib().bbme_flo_writer_submit_cells(self._w, os.fsencode(filename), width, height,
                                                             C.c_void_p(c.ctypes.data), c.shape[0], c.shape[1], pad_x, pad_y))
        return self._ticket()

    def wait(self, ticket=None):
        """Every job submitted so far, or (ticket) every job up to that one; the arrays of later jobs stay referenced."""
        if ticket is not None:
            _capi.check(_capi.lib().bbme_flo_writer_wait_ticket(self._w, int(ticket)))
            return
        try:
            _capi.check(_capi.lib().bbme_flo_writer_wait(self._w))
        finally:
            self._keep.clear()

    def close(self):
        if self._w:
            _capi.lib().bbme_flo_writer_destroy(self._w)
            self._w = C.c_void_p()
            self._keep.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
