"""Minimal device-buffer helper over the C ABI (gpx_malloc / gpx_memcpy_*).

Used by the benchmark and the tests to keep inputs resident in HBM and to call
the device-level entry points (gpx_d_*) directly.  Not a tensor library: a
DeviceBuffer is a pointer, a byte count and the numpy shape/dtype to read it
back with.
"""
import ctypes

import numpy as np

from . import _lib


class DeviceBuffer(object):
    def __init__(self, shape, dtype=np.float64):
        self.shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = ctypes.c_void_p()
        _lib.check(_lib.load().gpx_malloc(ctypes.byref(p), max(self.nbytes, 16)))
        self.ptr = p

    @classmethod
    def from_host(cls, a, dtype=None):
        a = np.ascontiguousarray(a, dtype=dtype if dtype is not None else a.dtype)
        buf = cls(a.shape, a.dtype)
        _lib.check(_lib.load().gpx_memcpy_h2d(buf.ptr, a.ctypes.data_as(ctypes.c_void_p),
                                              buf.nbytes, None))
        return buf

    def to_host(self):
        out = np.empty(self.shape, dtype=self.dtype)
        _lib.check(_lib.load().gpx_memcpy_d2h(out.ctypes.data_as(ctypes.c_void_p), self.ptr,
                                              self.nbytes, None))
        return out

    def zero(self):
        _lib.check(_lib.load().gpx_memset(self.ptr, 0, self.nbytes, None))
        _lib.check(_lib.load().gpx_device_sync())
        return self

    def free(self):
        if self.ptr:
            _lib.load().gpx_free(self.ptr)
            self.ptr = ctypes.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sync():
    _lib.check(_lib.load().gpx_device_sync())


class Event(object):
    """HIP event on a gpx stream (None = the null stream)."""

    def __init__(self):
        self.ev = ctypes.c_void_p()
        _lib.check(_lib.load().gpx_event_create(ctypes.byref(self.ev)))

    def record(self, stream=None):
        _lib.check(_lib.load().gpx_event_record(self.ev, stream))

    def sync(self):
        _lib.check(_lib.load().gpx_event_sync(self.ev))

    def elapsed_ms(self, stop):
        ms = ctypes.c_float(0.0)
        _lib.check(_lib.load().gpx_event_elapsed_ms(self.ev, stop.ev, ctypes.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if self.ev:
                _lib.load().gpx_event_destroy(self.ev)
        except Exception:
            pass
