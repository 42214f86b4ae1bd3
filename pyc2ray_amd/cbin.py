"""C2Ray binary grids: the wire format either side of the hot path.

The reference writes its outputs with ``tools21cm.save_cbin`` (``pyc2ray/c2ray_base.py:409-421``: ``xfrac``
as 64-bit, ``IonRates`` as 32-bit floats, Fortran order) and restarts from them with ``tools21cm.read_cbin``
(``c2ray_fstar.py:317-320``).  tools21cm is a third-party dependency that is absent from the reference
snapshot and from this image; the format below restates its published ``file_io`` routines (the Fortran
C2Ray's unformatted-stream files): a header of three little-endian int32 mesh sizes followed by the flat array.
No fixture of this format ships with the reference, so the byte layout is **parity unpinned** beyond that
description; ``tests/test_boundary.py::test_cbin_round_trip`` checks layout and round trip.
"""
from __future__ import annotations

import numpy as np


def save_cbin(filename, data, bits=32, order="C"):
    """Write ``data`` (3-D) as a C2Ray binary file: int32[3] mesh, then the array as float32 / float64 in
    ``order`` ('F' for files the Fortran code and the reference's ``write_output`` exchange)."""
    if bits not in (32, 64):
        raise ValueError("bits must be 32 or 64")
    data = np.asarray(data)
    if data.ndim != 3:
        raise ValueError("save_cbin expects a 3-D grid")
    dtype = "<f4" if bits == 32 else "<f8"
    with open(filename, "wb") as f:
        np.asarray(data.shape, dtype="<i4").tofile(f)
        np.ravel(data, order=order).astype(dtype).tofile(f)


def read_cbin(filename, bits=32, order="C", dimensions=3):
    """Read a file written by ``save_cbin`` (or by C2Ray); returns the grid in its mesh shape."""
    if bits not in (32, 64):
        raise ValueError("bits must be 32 or 64")
    dtype = "<f4" if bits == 32 else "<f8"
    with open(filename, "rb") as f:
        mesh = np.fromfile(f, dtype="<i4", count=dimensions)
        if mesh.size != dimensions or np.any(mesh <= 0):
            raise ValueError(f"{filename}: bad mesh header {mesh}")
        n = int(np.prod(mesh.astype(np.int64)))
        flat = np.fromfile(f, dtype=dtype, count=n)
    if flat.size != n:
        raise ValueError(f"{filename}: expected {n} values, found {flat.size}")
    return flat.reshape(tuple(int(m) for m in mesh), order=order)


def write_output(results_basename, z, xh, phi_ion, ext=".dat"):
    """The reference's ``C2Ray.write_output`` (``c2ray_base.py:397-427``): xfrac (f64) and IonRates (f32)."""
    suffix = f"_z{z:.3f}" + ext
    if suffix.endswith(".dat"):
        save_cbin(results_basename + "xfrac" + suffix, xh, bits=64, order="F")
        save_cbin(results_basename + "IonRates" + suffix, phi_ion, bits=32, order="F")
    elif suffix.endswith(".npy"):
        np.save(results_basename + "xfrac" + suffix, xh)
        np.save(results_basename + "IonRates" + suffix, phi_ion)
    else:
        raise ValueError("ext must be '.dat' or '.npy'")
