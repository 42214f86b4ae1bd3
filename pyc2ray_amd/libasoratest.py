"""Stand-in for the reference's ``pyc2ray.lib.libasoratest`` module (src/asora/python_test_module.cu):
``cinterp(pos0, dens)`` -> (4,4,4,2) array of (column density, path) around source ``pos0``."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import libasora as _la

_ip = C.POINTER(C.c_int32)
_dp = C.POINTER(C.c_double)


def cinterp(pos0, dens, fortran_constants=True):
    """Reference signature ``libasoratest.cinterp(pos0, dens)`` (tests/test_asora_internal.py:16)."""
    dens = np.ascontiguousarray(dens, dtype=np.float64)
    if dens.ndim != 3 or len({*dens.shape}) != 1:
        raise ValueError("dens must be a cubic 3-D array")
    p = (C.c_int32 * 3)(*[int(v) for v in pos0])
    out = np.empty((4, 4, 4, 2))
    rc = _la._lib.asora_test_cinterp(p, dens.ctypes.data_as(_dp), dens.shape[0], out.ctypes.data_as(_dp),
                                     1 if fortran_constants else 0)
    if rc:
        raise MemoryError(_la._lib.asora_last_error().decode())  # the reference raises MemoryError here
    return out


def cinterp_ex(pos0, dens, sigma=1.0, lo=0, cnt=4, fortran_constants=True):
    """Generalised harness: offsets lo..lo+cnt-1 per axis, any sigma."""
    dens = np.ascontiguousarray(dens, dtype=np.float64)
    p = (C.c_int32 * 3)(*[int(v) for v in pos0])
    out = np.empty((cnt, cnt, cnt, 2))
    rc = _la._lib.asora_test_cinterp_ex(p, dens.ctypes.data_as(_dp), dens.shape[0], float(sigma), int(lo), int(cnt),
                                        out.ctypes.data_as(_dp), 1 if fortran_constants else 0)
    if rc:
        raise MemoryError(_la._lib.asora_last_error().decode())
    return out
