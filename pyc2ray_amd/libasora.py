"""Drop-in stand-in for the reference's ``pyc2ray.lib.libasora`` extension module.

Same six functions, same positional signatures (reference: ``src/asora/python_module.cu:139-150``,
argument formats ``:35-38, :69, :90, :105, :122``), implemented as a thin ctypes layer over the
C-ABI of ``libasora_hip.so`` (``include/asora_hip.h``).  Differences, all deliberate
(SURVEY.md 8(b)):

* a failing HIP call raises ``RuntimeError`` (the reference prints to stderr and carries on);
* array arguments are validated (dtype, contiguity, size) instead of being reinterpreted blindly;
* there is no CPU fallback: importing this module without the built library raises ``ImportError``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def _find_library():
    """libasora_hip.so next to this file (the drop-in layout: both files copied into ``pyc2ray/lib/``, where the
    reference's meson build installs its extension, src/asora/meson.build:39-64) or in ``lib/`` below it (this
    repository's layout).  ASORA_HIP_LIB names an alternative build of the same library (A/B experiments); it is
    never a fallback."""
    if os.environ.get("ASORA_HIP_LIB"):
        return os.environ["ASORA_HIP_LIB"]
    cands = [os.path.join(_HERE, "libasora_hip.so"), os.path.join(_HERE, "lib", "libasora_hip.so")]
    for p in cands:
        if os.path.exists(p):
            return p
    return cands[-1]


LIB_PATH = _find_library()

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)

# every symbol include/asora_hip.h declares: (restype, argtypes)
ABI = {
    "asora_device_init": (C.c_int, [C.c_int] * 4),
    "asora_device_close": (C.c_int, []),
    "asora_device_index": (C.c_int, [C.POINTER(C.c_int)]),
    "asora_density_to_device": (C.c_int, [_dp, C.c_int]),
    "asora_photo_table_to_device": (C.c_int, [_dp, _dp, C.c_int]),
    "asora_source_data_to_device": (C.c_int, [_ip, _dp, C.c_int]),
    "asora_do_all_sources": (C.c_int, [C.c_double, _dp, C.c_double, C.c_double, _dp, _dp, _dp, C.c_int, C.c_int,
                                       C.c_double, C.c_double, C.c_int]),
    "asora_heat_table_to_device": (C.c_int, [_dp, _dp, C.c_int]),
    "asora_do_all_sources_heat": (C.c_int, [C.c_double, _dp, C.c_double, C.c_double, _dp, _dp, _dp, _dp, C.c_int,
                                            C.c_int, C.c_double, C.c_double, C.c_int]),
    "asora_do_all_sources_dev": (C.c_int, [C.c_double, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                           C.c_void_p]),
    "asora_density_from_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "asora_global_pass": (C.c_int, [C.c_double] + [_dp] * 7 + [C.c_double] * 5 + [C.c_int, C.POINTER(C.c_int)]),
    "asora_global_pass_dev": (C.c_int, [C.c_double] + [C.c_void_p] * 7 + [C.c_double] * 5 +
                              [C.c_int, C.POINTER(C.c_longlong), _dp, _dp, C.c_void_p]),
    "asora_global_pass_dev_noalloc": (C.c_int, [C.c_double] + [C.c_void_p] * 7 + [C.c_double] * 5 +
                                      [C.c_int, C.POINTER(C.c_longlong), _dp, _dp, C.c_void_p]),
    "asora_test_cinterp": (C.c_int, [_ip, _dp, C.c_int, _dp, C.c_int]),
    "asora_test_cinterp_ex": (C.c_int, [_ip, _dp, C.c_int, C.c_double, C.c_int, C.c_int, _dp, C.c_int]),
    "asora_visit_count": (C.c_int, [C.c_double, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "asora_geometry_census": (C.c_int, [C.c_double, C.c_double, C.c_int] + [C.POINTER(C.c_longlong)] * 3 +
                              [C.POINTER(C.c_int)] * 2),
    "asora_last_sweep_ms": (C.c_int, [C.POINTER(C.c_float)]),
    "asora_last_launch_info": (C.c_int, [C.POINTER(C.c_int)] * 4),
    "asora_set_option": (C.c_int, [C.c_char_p, C.c_longlong]),
    "asora_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_longlong)]),
    "asora_source_order": (C.c_int, [C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)]),
    "asora_last_sweep_is_table_driven": (C.c_int, []),
    "asora_last_launch_notice": (C.c_char_p, []),
    "asora_test_division": (C.c_int, [_dp, _dp, C.c_longlong] + [C.POINTER(C.c_longlong)] * 3),
    "asora_last_error": (C.c_char_p, []),
}


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so; if this shim
    loaded the system one first, a later ``import torch`` (used for RCCL / device tensors) would bring a
    second runtime into the process and see no GPU.  So when torch is installed but not yet imported,
    its bundled runtime is loaded first and libasora_hip.so binds to it (same soname)."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the ASORA path.")
    _preload_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in ABI.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = _load()


def _check(rc):
    if rc != 0:
        raise RuntimeError(f"libasora_hip error {rc}: {_lib.asora_last_error().decode()}")


def _f64(a, name, size=None):
    if not isinstance(a, np.ndarray) or a.dtype != np.float64:
        raise TypeError(f"{name} must be Array of type double")  # wording of python_module.cu:43
    if not a.flags.c_contiguous:
        raise ValueError(f"{name} must be C-contiguous")
    if size is not None and a.size < size:
        raise ValueError(f"{name} has {a.size} elements, need {size}")
    return a.ctypes.data_as(_dp)


def device_init(N, num_src_par, mpi_rank, num_gpus):
    """Select device ``mpi_rank % num_gpus`` and allocate grid memory (memory.cu:42-86)."""
    _check(_lib.asora_device_init(int(N), int(num_src_par), int(mpi_rank), int(num_gpus)))


def device_close():
    """Free device memory (memory.cu:138-150)."""
    _check(_lib.asora_device_close())


def device_index():
    """The HIP device ``device_init`` selected (``mpi_rank % num_gpus``)."""
    d = C.c_int(-1)
    _check(_lib.asora_device_index(C.byref(d)))
    return d.value


def density_to_device(ndens, N):
    """Copy the density grid (flat float64, N^3) to the device (memory.cu:91-98)."""
    N = int(N)
    _check(_lib.asora_density_to_device(_f64(ndens, "ndens", N ** 3), N))


def photo_table_to_device(thin_table, thick_table, num_tau):
    """Copy ``num_tau`` entries of the thin/thick photo-ionisation tables (memory.cu:100-115)."""
    n = int(num_tau)
    _check(_lib.asora_photo_table_to_device(_f64(thin_table, "thin_table", n), _f64(thick_table, "thick_table", n), n))


def heat_table_to_device(heat_thin_table, heat_thick_table, num_tau):
    """Extension: heating tables for ``do_all_sources_heat`` (photorates.f90:118,124)."""
    n = int(num_tau)
    _check(_lib.asora_heat_table_to_device(_f64(heat_thin_table, "heat_thin_table", n),
                                           _f64(heat_thick_table, "heat_thick_table", n), n))


def source_data_to_device(pos, flux, num_src):
    """Copy source positions (int32, x0,y0,z0,x1,... 0-based, periodic: a coordinate outside [0, N) addresses
    cell ``coordinate % N`` like in the reference) and fluxes (memory.cu:117-133)."""
    n = int(num_src)
    if not isinstance(pos, np.ndarray) or pos.dtype != np.int32 or not pos.flags.c_contiguous or pos.size < 3 * n:
        raise TypeError("pos must be a C-contiguous int32 array with 3*num_src entries")
    _check(_lib.asora_source_data_to_device(pos.ctypes.data_as(_ip), _f64(flux, "flux", n), n))


def do_all_sources(R, coldensh_out, sig, dr, ndens, xh_av, phi_ion, num_src, m1, minlogtau, dlogtau, num_tau):
    """Raytrace all sources; ``phi_ion`` and ``coldensh_out`` are overwritten in place; returns None.

    Signature of python_module.cu:21-59.  ``ndens`` is ignored like in the reference (the grid
    uploaded by ``density_to_device`` is used)."""
    m1 = int(m1)
    n3 = m1 ** 3
    _check(_lib.asora_do_all_sources(float(R), _f64(coldensh_out, "coldensh_out", n3), float(sig), float(dr), None,
                                     _f64(xh_av, "xh_av", n3), _f64(phi_ion, "phi_ion", n3), int(num_src), m1,
                                     float(minlogtau), float(dlogtau), int(num_tau)))
    return None


def do_all_sources_heat(R, coldensh_out, sig, dr, ndens, xh_av, phi_ion, phi_heat, num_src, m1, minlogtau, dlogtau,
                        num_tau):
    """Extension: as ``do_all_sources`` plus the heating-rate grid ``phi_heat``."""
    m1 = int(m1)
    n3 = m1 ** 3
    _check(_lib.asora_do_all_sources_heat(float(R), _f64(coldensh_out, "coldensh_out", n3), float(sig), float(dr),
                                          None, _f64(xh_av, "xh_av", n3), _f64(phi_ion, "phi_ion", n3),
                                          _f64(phi_heat, "phi_heat", n3), int(num_src), m1, float(minlogtau),
                                          float(dlogtau), int(num_tau)))
    return None


# ---- device-pointer entry points (integers = device addresses, e.g. torch.Tensor.data_ptr()) ---------

def do_all_sources_dev(R, coldensh_out_ptr, sig, dr, xh_av_ptr, phi_ion_ptr, phi_heat_ptr, num_src, m1, minlogtau,
                       dlogtau, num_tau, stream=0):
    _check(_lib.asora_do_all_sources_dev(float(R), C.c_void_p(coldensh_out_ptr or None), float(sig), float(dr),
                                         C.c_void_p(xh_av_ptr), C.c_void_p(phi_ion_ptr),
                                         C.c_void_p(phi_heat_ptr or None), int(num_src), int(m1), float(minlogtau),
                                         float(dlogtau), int(num_tau), C.c_void_p(stream or None)))


def density_from_device(ndens_ptr, N, stream=0):
    _check(_lib.asora_density_from_device(C.c_void_p(ndens_ptr), int(N), C.c_void_p(stream or None)))


def global_pass(dt, ndens, temp, xh, xh_av, xh_intermed, phi_ion, clump, bh00, albpow, colh0, temph0, abu_c):
    """Chemistry pass on host grids: signature and semantics of the reference's f2py routine
    ``libc2ray.chemistry.global_pass`` (chemistry.f90:13-49, called at evolve.py:326-340): ``xh_av`` and
    ``xh_intermed`` are updated in place, the number of non-converged cells is returned.  All seven
    grids must be float64 (N, N, N) arrays of one common memory layout (all C- or all F-contiguous)."""
    grids = (ndens, temp, xh, xh_av, xh_intermed, phi_ion, clump)
    N = grids[0].shape[0]
    for g in grids:
        if not isinstance(g, np.ndarray) or g.dtype != np.float64 or g.shape != (N, N, N):
            raise TypeError("global_pass: grids must be float64 arrays of shape (N, N, N)")
    if not (all(g.flags.c_contiguous for g in grids) or all(g.flags.f_contiguous for g in grids)):
        raise ValueError("global_pass: all grids must share one contiguous layout")
    conv = C.c_int(0)
    ptr = [g.ctypes.data_as(_dp) for g in grids]
    _check(_lib.asora_global_pass(float(dt), *ptr, float(bh00), float(albpow), float(colh0), float(temph0),
                                  float(abu_c), int(N), C.byref(conv)))
    return conv.value


def global_pass_dev(dt, ndens_ptr, temp_ptr, xh_ptr, xh_av_ptr, xh_intermed_ptr, phi_ion_ptr, clump_ptr, bh00,
                    albpow, colh0, temph0, abu_c, m1, stream=0, want_stats=True, noalloc=False):
    """Chemistry pass on device grids (integers = device addresses).  Returns
    ``(conv_flag, sum(xh_intermed), sum(1 - xh_intermed))`` after waiting for the stream, or ``None``
    (asynchronous) when ``want_stats`` is false.  ``noalloc=True`` goes through ``asora_global_pass_dev_noalloc``,
    the entry point that allocates nothing (include/asora_hip.h)."""
    cf, s1, s0 = C.c_longlong(0), C.c_double(0.0), C.c_double(0.0)
    out = (C.byref(cf), C.byref(s1), C.byref(s0)) if want_stats else (None, None, None)
    fn = _lib.asora_global_pass_dev_noalloc if noalloc else _lib.asora_global_pass_dev
    _check(fn(float(dt), *[C.c_void_p(p) for p in (ndens_ptr, temp_ptr, xh_ptr, xh_av_ptr,
                                                                           xh_intermed_ptr, phi_ion_ptr, clump_ptr)],
                                      float(bh00), float(albpow), float(colh0), float(temph0), float(abu_c), int(m1),
                                      *out, C.c_void_p(stream or None)))
    return (cf.value, s1.value, s0.value) if want_stats else None


def visit_count(R, m1):
    """(cells visited per source, of which inside the R sphere) -- the benchmark's unit of work."""
    v, s = C.c_longlong(), C.c_longlong()
    _check(_lib.asora_visit_count(float(R), int(m1), C.byref(v), C.byref(s)))
    return v.value, s.value


def geometry_census(R, dr, m1):
    """Host-side census of the sweep's cell tables: dict(owned, owned_rate, computed, staging_entries, variants)."""
    a, b, c = C.c_longlong(), C.c_longlong(), C.c_longlong()
    e, v = C.c_int(), C.c_int()
    _check(_lib.asora_geometry_census(float(R), float(dr), int(m1), C.byref(a), C.byref(b), C.byref(c), C.byref(e),
                                      C.byref(v)))
    return dict(owned=a.value, owned_rate=b.value, computed=c.value, staging_entries=e.value, variants=v.value)


def source_order(pos, N, order=2):
    """Permutation in which the library sweeps the source list `pos` (flat int32 x0,y0,z0,x1,...)."""
    pos = np.ascontiguousarray(pos, dtype=np.int32).ravel()
    n = pos.size // 3
    perm = np.empty(max(n, 1), dtype=np.int32)
    _check(_lib.asora_source_order(pos.ctypes.data_as(C.POINTER(C.c_int32)), n, int(N), int(order),
                                   perm.ctypes.data_as(C.POINTER(C.c_int32))))
    return perm[:n]


def set_option(name, value):
    _check(_lib.asora_set_option(name.encode(), int(value)))


def get_option(name):
    v = C.c_longlong()
    _check(_lib.asora_get_option(name.encode(), C.byref(v)))
    return v.value


def test_division(num, den):
    """(mismatches of div_exact, rcp_exact, div_const) against IEEE division, evaluated on the device."""
    num = np.ascontiguousarray(num, dtype=np.float64)
    den = np.ascontiguousarray(den, dtype=np.float64)
    a, b, c = C.c_longlong(), C.c_longlong(), C.c_longlong()
    _check(_lib.asora_test_division(_f64(num, "num"), _f64(den, "den", num.size), num.size, C.byref(a), C.byref(b),
                                    C.byref(c)))
    return a.value, b.value, c.value


def last_sweep_ms():
    ms = C.c_float()
    _check(_lib.asora_last_sweep_ms(C.byref(ms)))
    return ms.value


def last_launch_info():
    g, b, l, s = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    _check(_lib.asora_last_launch_info(C.byref(g), C.byref(b), C.byref(l), C.byref(s)))
    return dict(grid=g.value, block=b.value, lds_bytes=l.value, staging={0: "global", 1: "lds", 2: "lds-compact"}[s.value & 3], quad=bool((s.value >> 2) & 3), sectors_per_workgroup=1 << ((s.value >> 2) & 3),
                sources_per_workgroup=1 << ((s.value >> 4) & 3),
                single_layout=bool(s.value & 64),
                table_driven=bool(_lib.asora_last_sweep_is_table_driven()),
                notice=_lib.asora_last_launch_notice().decode())
